#!/usr/bin/env python
"""Fused pruned tail + draw (hd_tail_fused.hip.h) against the separate launches: tokens of full samples and seconds per sample, in two
processes (HUDIFF_TAIL is read once; TAIL_A / TAIL_B pick the two forms, default 0 and 2).  python scripts/tail_fused_check.py [ab|nb] [B] [route]"""
import os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(kind, B, route, out):
    import hudiff_amd
    from hudiff_amd import synthetic as S, evalsets as E
    cfg = dict(S.AB_CONFIG if kind == "ab" else S.NB_CONFIG)
    m = (hudiff_amd.AntiTFNet if kind == "ab" else hudiff_amd.NanoAntiTFNet)(**cfg, precision=route)
    m.load_state_dict(S.random_state_dict(kind, cfg, seed=0))
    b = E.eval_batch("huab348" if kind == "ab" else "vhh", B, row0=3)
    res = {}
    for name, kw in (("philox", {}), ("noise", dict(q_noise=np.random.default_rng(1).exponential(size=(int(b["T"].max()), B, 22)).astype(np.float32)))):
        m.sample(b["tokens"], b["region"], b["chain"], b["order"], np.minimum(b["T"], 3), seed=4, row0=0, **kw)     # warm (graph capture)
        t0 = time.perf_counter()
        res["tokens_" + name] = m.sample(b["tokens"], b["region"], b["chain"], b["order"], b["T"], seed=4, row0=0, **kw)
        res["t_" + name] = time.perf_counter() - t0
    res["info"] = str(m.precision_info())
    np.savez(out, **res)


if __name__ == "__main__":
    if len(sys.argv) > 4:
        child(sys.argv[1], int(sys.argv[2]), sys.argv[3], sys.argv[4])
        sys.exit(0)
    kind = sys.argv[1] if len(sys.argv) > 1 else "ab"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    route = sys.argv[3] if len(sys.argv) > 3 else "split"
    outs = []
    for f in (os.environ.get("TAIL_A", "0"), os.environ.get("TAIL_B", "2")):
        out = f"/tmp/tail_fused_{kind}_{f}.npz"
        subprocess.run([sys.executable, os.path.abspath(__file__), kind, str(B), route, out], env=dict(os.environ, HUDIFF_TAIL=f), check=True)
        outs.append(np.load(out))
    a, b = outs
    for k in ("tokens_philox", "tokens_noise"):
        same = (a[k] == b[k])
        print(kind, B, route, k, "rows with identical tokens", int(same.all(1).sum()), "of", B, "| differing slots", int((~same).sum()), "of", same.size)
    print(kind, B, route, "seconds per sample, form A / form B:", round(float(a["t_philox"]), 4), round(float(b["t_philox"]), 4),
          "| seq/s", round(B / float(a["t_philox"]), 2), round(B / float(b["t_philox"]), 2), "| info", b["info"])
