cd /root/repo
python scripts/tmp/attn_bisect.py 0 /tmp/ref.npy
for v in split1 split2; do HUDIFF_LIB=/root/repo/scripts/tmp/lib_$v.so python scripts/tmp/attn_bisect.py 1 /tmp/m$v.npy; done
HUDIFF_LIB=/root/repo/scripts/tmp/gpurun_tmp_lib_23.so python scripts/tmp/attn_bisect.py 1 /tmp/m23.npy
python - <<'PY'
import numpy as np
r = np.load('/tmp/ref.npy')
for mk in ('split1', 'split2', '23'):
    d = np.abs(np.load(f'/tmp/m{mk}.npy') - r)
    print(mk, d.max(), (d.max(axis=(1,2)) > 1e-4).sum(), 'rows bad of', d.shape[0])
PY
