#!/usr/bin/env python
"""Shader-clock stamps of the attention core inside the fused kernel (libraries built with -DHD_QA_STAMPS=1): per wave, its first two
query tiles -- tile start, Q fragment ready, S^T done, softmax done, O^T done, rows stored -- of the head-0 workgroups.
HUDIFF_LIB=... python scripts/r05/core_stamps.py [B]   (GPU box)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import hudiff_amd
from hudiff_amd import evalsets as E, synthetic as S
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
cfg = dict(S.AB_CONFIG); sd = S.random_state_dict("ab", cfg, seed=0)
m = hudiff_amd.AntiTFNet(**cfg, precision="split", options={"fused_attn_min_grid": 0}); m.load_state_dict(sd)
batch = E.eval_batch("huab348", B, row0=0)
m.debug_stop_after(100)
m(batch["tokens"], batch["region"], batch["chain"], dropout="off")
q = m.debug_read("QKV", B)[:, 2, 1024:1024 + 192].reshape(B, 12, 16)[:, :, :12].reshape(B, 12, 2, 6)
names = ["start", "Q ready", "S done", "softmax", "PV done", "stored"]
print(f"lib {os.environ.get('HUDIFF_LIB', 'in-tree')}: median over {B} head-0 workgroups, shader clocks since the wave entered the core")
for w in range(12):
    for t in range(2):
        v = np.median(q[:, w, t, :], axis=0)
        if t == 1 and w >= 7:
            continue                                     # waves 7 .. 11 have one tile (19 tiles on 12 waves)
        d = np.diff(v)
        print(f"  wave {w:2d} tile {t}: " + "  ".join(f"{n} {x:6.0f}" for n, x in zip(names, v)) + "   | phases " + " ".join(f"{x:5.0f}" for x in d))
m.close()
