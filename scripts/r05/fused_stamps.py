#!/usr/bin/env python
"""Phase time stamps of the fused Q|K|V + attention kernel (HUDIFF_QA_ABL=32): 100 MHz ticks since kernel entry per wave of the head-0 workgroups,
read back from the unused V third of the QKV buffer.  python scripts/r05/fused_stamps.py [B] [ab|nb]   (GPU box)"""
import os, sys
import numpy as np
os.environ["HUDIFF_QA_ABL"] = "32"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import hudiff_amd
from hudiff_amd import evalsets as E, synthetic as S
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
kind = sys.argv[2] if len(sys.argv) > 2 else "ab"
cfg = dict(S.AB_CONFIG if kind == "ab" else S.NB_CONFIG); sd = S.random_state_dict(kind, cfg, seed=0)
cls = hudiff_amd.AntiTFNet if kind == "ab" else hudiff_amd.NanoAntiTFNet
m = cls(**cfg, precision="split", options={"fused_attn_min_grid": 0}); m.load_state_dict(sd)
batch = E.eval_batch("huab348" if kind == "ab" else "vhh", B, row0=0)
for stage, what in ((100, "first attention of block 0 (no LayerNorm fold)"), (3, "second attention of block 0 (LayerNorm folded)")):
    m.debug_stop_after(stage)
    m(batch["tokens"], batch["region"], batch["chain"], dropout="off")
    q = m.debug_read("QKV", B)[:, 0, 1024:1024 + 96].reshape(B, 12, 8)[:, :, :6] / 100.0      # microseconds
    names = ["K loop done", "step 1 done", "past barrier 1", "step 2 done", "past barrier 2", "attention done"]
    print(f"== {what}: mean over {B} head-0 workgroups, microseconds since kernel entry")
    for part, waves in (("Q waves (0, 1, 6, 7)", [0, 1, 6, 7]), ("K waves (2, 3, 8, 9)", [2, 3, 8, 9]), ("V waves (4, 5, 10, 11)", [4, 5, 10, 11])):
        v = q[:, waves, :].mean(axis=(0, 1))
        print(f"  {part:24s} " + "  ".join(f"{n} {x:7.2f}" for n, x in zip(names, v)))
m.close()
