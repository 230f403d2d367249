#!/usr/bin/env python
"""Debug aid (round 5): where does the fused Q|K|V + attention kernel differ from the two-launch form?  python scripts/r05/fused_debug.py [ab|nb] [B]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import hudiff_amd
from hudiff_amd import evalsets as E, synthetic as S

def x16(a):            # [.., C] float32 holding X16 rows -> float32 values
    h = a.view(np.float16).reshape(*a.shape[:-1], a.shape[-1] // 16, 2, 16).astype(np.float32)
    return (h[..., 0, :] + h[..., 1, :]).reshape(*a.shape[:-1], a.shape[-1])

kind = sys.argv[1] if len(sys.argv) > 1 else "ab"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
cfg = dict(S.AB_CONFIG if kind == "ab" else S.NB_CONFIG)
sd = S.random_state_dict(kind, cfg, seed=0)
cls = hudiff_amd.AntiTFNet if kind == "ab" else hudiff_amd.NanoAntiTFNet
ms = {}
for name, f in (("two", 0), ("one", 1)):
    m = cls(**cfg, precision="split", options={"fused_attn": f, "fused_attn_min_grid": 0}); m.load_state_dict(sd); ms[name] = m
batch = E.eval_batch("huab348" if kind == "ab" else "vhh", B, row0=5)
L = batch["tokens"].shape[1]
for stage in (100, 3):
    out = {}
    for name, m in ms.items():
        m.debug_stop_after(stage)
        m(batch["tokens"], batch["region"], batch["chain"], dropout="off")
        out[name] = {k: m.debug_read(k, B) for k in ("Y", "AT", "O", "QKV")}
        m.debug_stop_after(0)
    print(f"== stage {stage}")
    for k in ("Y", "AT", "O", "QKV"):
        a, b = out["two"][k], out["one"][k]
        if k == "O":
            a, b = x16(a), x16(b)
        if k == "QKV":
            a, b = a[..., :cfg["att_model"]], b[..., :cfg["att_model"]]
        d = np.abs(a - b)
        print(f"{k}: max |two - one| {d.max():.3e}  max |two| {np.abs(a).max():.3e}  finite {np.isfinite(b).all()}")
        if k in ("O", "QKV") and d.max() > 0:
            per_head = d.reshape(B, L, 8, 64).max(axis=(0, 1, 3))
            per_rowtile = np.array([d[:, i:i + 16].max() for i in range(0, L, 16)])
            per_seq = d.max(axis=(1, 2))
            per_d = d.reshape(B, L, 8, 64).max(axis=(0, 1, 2))
            print("  per head   ", np.array2string(per_head, precision=2))
            print("  per q tile ", np.array2string(per_rowtile, precision=2, max_line_width=200))
            print("  per seq    ", np.array2string(per_seq, precision=2, max_line_width=200))
            print("  per d      ", np.array2string(per_d, precision=1, max_line_width=250))
for m in ms.values():
    m.close()

# ---- pattern hunt on Q after the first attention of block 0 ----
ms = {}
for name, f in (("two", 0), ("one", 1)):
    m = cls(**cfg, precision="split", options={"fused_attn": f, "fused_attn_min_grid": 0}); m.load_state_dict(sd); ms[name] = m
q = {}
for name, m in ms.items():
    m.debug_stop_after(100)
    m(batch["tokens"], batch["region"], batch["chain"], dropout="off")
    q[name] = m.debug_read("QKV", B)[..., :cfg["att_model"]]
    m.close()
np.set_printoptions(precision=4, linewidth=220, suppress=True)
print("two[0, 0:4, 0:8]\n", q["two"][0, 0:4, 0:8]); print("one[0, 0:4, 0:8]\n", q["one"][0, 0:4, 0:8])
print("two[0, 0:4, 64:72]\n", q["two"][0, 0:4, 64:72]); print("one[0, 0:4, 64:72]\n", q["one"][0, 0:4, 64:72])
flat2 = q["two"].reshape(-1, 8, 64)
for (b_, r_, h_) in ((0, 0, 0), (0, 5, 0), (0, 40, 3), (1, 200, 7)):
    v = q["one"][b_, r_, h_ * 64:(h_ + 1) * 64]
    d = np.abs(flat2 - v[None, None, :]).max(-1)          # [rows, heads]
    i = np.unravel_index(d.argmin(), d.shape)
    print(f"one[b {b_} row {r_} head {h_}] is closest to two[flat row {i[0]} = (b {i[0] // L}, row {i[0] % L}), head {i[1]}]: max diff {d.min():.3e}")
    # same row/head: per-d difference pattern
    dd = q["one"][b_, r_, h_ * 64:(h_ + 1) * 64] - q["two"][b_, r_, h_ * 64:(h_ + 1) * 64]
    print("   diff per d:", np.array2string(dd, precision=2, max_line_width=250))
