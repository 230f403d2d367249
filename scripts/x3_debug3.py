import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
kind = sys.argv[1]
import hudiff_amd
from hudiff_amd import synthetic as S
cfg = dict(S.AB_CONFIG if kind == "ab" else S.NB_CONFIG)
sd = S.random_state_dict(kind, cfg, seed=21)
B = 32 if kind == "ab" else 56
batch = S.synthetic_batch(kind, B, seed=9)
cls = hudiff_amd.AntiTFNet if kind == "ab" else hudiff_amd.NanoAntiTFNet
os.environ["HUDIFF_X3"] = "0"
m0 = cls(**cfg); m0.load_state_dict(sd)
os.environ["HUDIFF_X3"] = "1"
m1 = cls(**cfg); m1.load_state_dict(sd)
def run(m):
    m.debug_stop_after(2)
    m(batch["tokens"], batch["region"], batch["chain"], dropout="off", seed=5, row0=100, step=3)
    return m.debug_read("Y", B)
a0, a1 = run(m0), run(m0)
print("fp32 vs fp32 bitwise equal:", bool(np.array_equal(a0, a1)))
xs = [run(m1) for _ in range(6)]
for i, x in enumerate(xs):
    d = np.abs(x - a0)
    bad = np.argwhere(d.max(axis=2) > 1e-3)
    cols = np.argwhere(d > 1e-3)[:, 2] if len(bad) else np.array([0])
    print("x3 run", i, "max err vs fp32", float(d.max()), "bad (row,slot):", bad.tolist()[:8], "bad cols min/max/count", int(cols.min()), int(cols.max()), len(set(cols.tolist())))
