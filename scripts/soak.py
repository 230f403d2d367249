import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
import hudiff_amd, torch
from hudiff_amd import synthetic as S
cfg = dict(S.AB_CONFIG); sd = S.random_state_dict('ab', cfg, seed=0)
m = hudiff_amd.AntiTFNet(**cfg, device=0); m.load_state_dict(sd)
free0 = torch.cuda.mem_get_info()[0]
ref = None
for it, B in enumerate([256, 64, 300, 33, 256, 129, 256]):
    b = S.synthetic_batch('ab', B, seed=3)
    T = np.minimum(b['T'], 4)
    t = time.time()
    out = m.sample(b['tokens'], b['region'], b['chain'], b['order'], T, seed=11, row0=0)
    dt = time.time() - t
    if B == 256:
        if ref is None: ref = out
        else: assert np.array_equal(ref, out), "same call, different result"
    fwd = m(b['tokens'][:3], b['region'][:3], np.concatenate([b['chain'][:3], b['chain'][B:B+3]]), dropout='off')
    assert np.isfinite(fwd).all()
    print(f"iter {it} B={B}: {dt:.2f} s, free mem delta {(free0 - torch.cuda.mem_get_info()[0]) / 2**20:.0f} MiB", flush=True)
m.close()
print("closed; free mem delta", (free0 - torch.cuda.mem_get_info()[0]) / 2**20, "MiB")
