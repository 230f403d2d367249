"""Determinism stress of the split-precision path: N forwards of the same batch must be bit-identical and close to fp32."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
kind = sys.argv[1]; n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
import hudiff_amd
from hudiff_amd import evalsets as E
from hudiff_amd import synthetic as S
cfg = dict(S.AB_CONFIG if kind == "ab" else S.NB_CONFIG)
sd = S.random_state_dict(kind, cfg, seed=0)
cls = hudiff_amd.AntiTFNet if kind == "ab" else hudiff_amd.NanoAntiTFNet
B = 64 if kind == "ab" else 128
batch = E.eval_batch("huab348" if kind == "ab" else "vhh", B, row0=0)
os.environ["HUDIFF_X3"] = "0"
m0 = cls(**cfg); m0.load_state_dict(sd)
os.environ["HUDIFF_X3"] = "1"
m1 = cls(**cfg); m1.load_state_dict(sd)
for drop in ("off", "faithful"):
    ref = m0(batch["tokens"], batch["region"], batch["chain"], dropout=drop, seed=9, row0=0, step=1)
    first, bad = None, 0
    for i in range(n):
        x = m1(batch["tokens"], batch["region"], batch["chain"], dropout=drop, seed=9, row0=0, step=1)
        if first is None:
            first = x
        d = np.abs(x - ref)
        if not np.array_equal(x, first) or d.max() > 1e-4 or not np.isfinite(x).all():
            bad += 1
            rows = np.argwhere(d.max(axis=2) > 1e-4)
            print(kind, drop, "run", i, "differs: max err", float(d.max()), "bad (row,slot) count", len(rows), rows[:6].tolist(), flush=True)
    print(kind, drop, "runs", n, "bad", bad, "max err vs fp32 (run 0)", float(np.abs(first - ref).max()), flush=True)

# ---- complete samples (graph replay, two lanes): repeated runs must give identical tokens --------------------------------
Bs = 256
big = E.eval_batch("huab348" if kind == "ab" else "vhh", Bs, row0=0)
T = np.minimum(big["T"], 24)
for name, m in (("fp32", m0), ("x3", m1)):
    outs = [m.sample(big["tokens"], big["region"], big["chain"], big["order"], T, seed=5, row0=0) for _ in range(4)]
    same = all(np.array_equal(outs[0], o) for o in outs[1:])
    print(kind, name, "24-step samples of 256 rows x4 identical:", same, flush=True)
print(kind, "x3 tokens == fp32 tokens:", bool(np.array_equal(m0.sample(big["tokens"], big["region"], big["chain"], big["order"], T, seed=5, row0=0),
                                                                 m1.sample(big["tokens"], big["region"], big["chain"], big["order"], T, seed=5, row0=0))))
