"""Diagnostic for the round-2 finding "two co-resident workgroups that fill a CU's 163 840 B of LDS exactly corrupt each other":
a nanobody-shaped model with max_len = 160 makes attn_x3_k<10> ask for exactly 81 920 B per block.  With HUDIFF_LDS_NO_PAD=1 the
co-residency rule is bypassed (two blocks per CU, exact fill); without it the request is padded (one block per CU).
    HUDIFF_LDS_NO_PAD=1 python scripts/lds_fill_probe.py [forwards]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hudiff_amd
from hudiff_amd import synthetic as S
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
L = 160
cfg = dict(S.NB_CONFIG, max_len=L)
sd = S.random_state_dict("nb", cfg, seed=6)
os.environ["HUDIFF_X3"], os.environ["HUDIFF_ATTN_X3"] = "0", "0"
m32 = hudiff_amd.NanoAntiTFNet(**cfg); m32.load_state_dict(sd)
os.environ["HUDIFF_X3"] = "1"; os.environ.pop("HUDIFF_ATTN_X3")
mx3 = hudiff_amd.NanoAntiTFNet(**cfg); mx3.load_state_dict(sd)
B = 256
rng = np.random.default_rng(1)
tokens = rng.integers(0, 23, size=(B, L)).astype(np.int32); region = rng.integers(0, 7, size=(B, L)).astype(np.int32)
kw = dict(dropout="faithful", seed=5, row0=0, step=3)
ref = m32(tokens, region, None, **kw)
first, bad, worst = None, 0, 0.0
for i in range(n):
    x = mx3(tokens, region, None, **kw)
    first = x if first is None else first
    d = float(np.abs(x - ref).max())
    worst = max(worst, d)
    if not np.array_equal(x, first) or d > 1e-4:
        bad += 1
print(f"HUDIFF_LDS_NO_PAD={os.environ.get('HUDIFF_LDS_NO_PAD', '0')}: {n} forwards, {bad} differ from the first / from fp32 by > 1e-4; worst |dlogit| vs fp32 {worst:.2e}")
