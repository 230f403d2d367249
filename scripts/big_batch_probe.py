"""Launches beyond the 32-bit-offset kernels: B rows in ONE call (operands of >= 2 GiB per lane take the generic 64-bit-address
GEMMs and the fp32 attention kernel) against the same rows sampled in shards of 256 -- tokens must be identical.
    python scripts/big_batch_probe.py ab 3000 3"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hudiff_amd
from hudiff_amd import synthetic as S
kind, B, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
cfg = dict(S.AB_CONFIG if kind == "ab" else S.NB_CONFIG)
m = (hudiff_amd.AntiTFNet if kind == "ab" else hudiff_amd.NanoAntiTFNet)(**cfg)
m.load_state_dict(S.random_state_dict(kind, cfg, seed=0))
b = S.synthetic_batch(kind, B, seed=2023)
T = np.minimum(b["T"], steps)
t0 = time.time()
big = m.sample(b["tokens"], b["region"], b["chain"], b["order"], T, seed=5, row0=0)
t1 = time.time()
parts = []
for r0 in range(0, B, 256):
    r1 = min(B, r0 + 256)
    ch = None if b["chain"] is None else np.concatenate([b["chain"][r0:r1], b["chain"][B + r0:B + r1]])
    parts.append(m.sample(b["tokens"][r0:r1], b["region"][r0:r1], ch, b["order"][r0:r1], T[r0:r1], seed=5, row0=r0))
small = np.concatenate(parts)
print(kind, "B", B, "steps", steps, "one call", f"{t1 - t0:.1f} s", "rows that differ:", int((big != small).any(axis=1).sum()), m.precision_info())
sys.exit(0 if np.array_equal(big, small) else 1)
