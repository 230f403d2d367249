#!/usr/bin/env python
"""Per-tile shader-clock stamps of the fused kernel's projection loop (libraries built with -DHD_QA_STAMPS=1): k tiles 10 and 11 of the
head-0 workgroups -- tile start, last MFMA issued, operand DMA landed (vmcnt(0)), next tile's start (past the barrier).
HUDIFF_LIB=... python scripts/r05/tile_stamps.py [B]   (GPU box)"""
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
q = m.debug_read("QKV", B)[:, 1, 1024:1024 + 96].reshape(B, 12, 8)      # shader clocks since the start of tile 10
names = ["t10 dma issued", "mfma issued", "dma landed", "t11 start", "mfma issued", "dma landed", "t12 start"]
print(f"lib {os.environ.get('HUDIFF_LIB', 'in-tree')}: median over {B} head-0 workgroups, shader clocks since the start of k tile 10")
for part, waves in (("Q waves (0, 1, 6, 7)", [0, 1, 6, 7]), ("K waves (2, 3, 8, 9)", [2, 3, 8, 9]), ("V waves (4, 5, 10, 11)", [4, 5, 10, 11])):
    v = np.median(q[:, waves, 1:8], axis=(0, 1))
    print(f"  {part:24s} " + "  ".join(f"{n} {x:6.0f}" for n, x in zip(names, v)))
print("  per wave (median):")
for w in range(12):
    print(f"    wave {w:2d} ({'QKV'[((w % 6) >> 1)]}) " + "  ".join(f"{x:6.0f}" for x in np.median(q[:, w, 1:8], axis=0)))
m.close()
