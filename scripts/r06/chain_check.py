#!/usr/bin/env python
"""Row-owner chain kernel (hd_chain.hip.h) against the three-launch ByteNet blocks it replaces: logits of one forward with option
bn_chain = 0 and = mask on the same rows and weights (dropout off / faithful), and tokens of a short sample.   chain_check.py [ab|nb] [B] [mask]"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import hudiff_amd
from hudiff_amd import synthetic as S, evalsets as E

kind = sys.argv[1] if len(sys.argv) > 1 else "ab"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
mask = int(sys.argv[3]) if len(sys.argv) > 3 else 3
cfg = dict(S.AB_CONFIG if kind == "ab" else S.NB_CONFIG)
sd = S.random_state_dict(kind, cfg, seed=0)
cls = hudiff_amd.AntiTFNet if kind == "ab" else hudiff_amd.NanoAntiTFNet
b = E.eval_batch("huab348" if kind == "ab" else "vhh", B, row0=0)
res = {}
for name, opts in (("old", {"bn_chain": 0}), ("new", {"bn_chain": mask, "bn_chain_min_tiles": 1})):
    m = cls(**cfg, precision="split", options=opts); m.load_state_dict(sd)
    out = {}
    for dr in ("off", "faithful"):
        out[dr] = m(b["tokens"], b["region"], b["chain"], dropout=dr, seed=5, row0=0, step=3)
    m.debug_stop_after(1); m(b["tokens"], b["region"], b["chain"], dropout="off"); out["feat"] = m.debug_read("FEAT", B)
    m.debug_stop_after(2); m(b["tokens"], b["region"], b["chain"], dropout="off"); out["conv"] = m.debug_read("Y", B); out["yx"] = m.debug_read("YX", B)
    m.debug_stop_after(0)
    T4 = np.minimum(b["T"], 6)
    t0 = time.time()
    out["tok"] = m.sample(b["tokens"], b["region"], b["chain"], b["order"], T4, seed=11, row0=0)
    out["info"] = m.precision_info()
    m.close()
    res[name] = out
for k in ("feat", "conv", "yx", "off", "faithful"):
    a, c = res["old"][k], res["new"][k]
    print(kind, B, k, "max|old|", float(np.abs(a).max()), "max diff", float(np.abs(a - c).max()), "finite", bool(np.isfinite(c).all()))
print("tokens equal:", bool(np.array_equal(res["old"]["tok"], res["new"]["tok"])), res["new"]["info"])
