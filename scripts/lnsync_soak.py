"""Soak of the split-precision route with ln_sync under different lane counts (HUDIFF_LANES is read once per process): complete
256-row samples, repeated; prints a digest of the tokens so that runs can be compared across processes.
    HUDIFF_X3=1 HUDIFF_LANES=3 python scripts/lnsync_soak.py ab 3"""
import hashlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hudiff_amd
from hudiff_amd import synthetic as S, evalsets as E
kind = sys.argv[1]; reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
cfg = dict(S.AB_CONFIG if kind == "ab" else S.NB_CONFIG)
m = (hudiff_amd.AntiTFNet if kind == "ab" else hudiff_amd.NanoAntiTFNet)(**cfg)
m.load_state_dict(S.random_state_dict(kind, cfg, seed=0))
b = E.eval_batch("huab348" if kind == "ab" else "vhh", 256, row0=0)
digs = []
t0 = time.time()
for r in range(reps):
    tok = m.sample(b["tokens"], b["region"], b["chain"], b["order"], b["T"], seed=5, row0=0)
    digs.append(hashlib.sha256(tok.tobytes()).hexdigest()[:16])
print(kind, "lanes", os.environ.get("HUDIFF_LANES", "2"), "x3", os.environ.get("HUDIFF_X3", "0"), "lnsync", os.environ.get("HUDIFF_X3_LNSYNC", "2"),
      "digests", sorted(set(digs)), f"{(time.time() - t0) / reps:.2f} s per sample", m.precision_info())
