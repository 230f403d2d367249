#!/usr/bin/env python
"""Error of the split-precision GEMM path (HUDIFF_X3=1) next to the fp32 path's own, at production width.

    python scripts/x3_eval.py [ab|nb]          (GPU box)

Same weights, same rows (HuAb348 / VHH fixture), B = 32 per forward so that every GEMM takes the big-launch kernels:
  * max |dlogit| of fp32-HIP and of x3-HIP against a float64 evaluation of the oracle (4 rows; rows are independent),
    dropout off and dropout on (shared Philox masks);
  * a 256-row complete sample with identical noise on both paths: rows whose final tokens are identical.
Prints one JSON line.
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "ab"
    import hudiff_oracle as ho
    import hudiff_amd
    from hudiff_amd import evalsets as E
    from hudiff_amd import synthetic as S
    cfg = dict(S.AB_CONFIG if kind == "ab" else S.NB_CONFIG)
    sd = S.random_state_dict(kind, cfg, seed=0)
    cls = hudiff_amd.AntiTFNet if kind == "ab" else hudiff_amd.NanoAntiTFNet
    os.environ["HUDIFF_X3"] = "0"
    m32 = cls(**cfg); m32.load_state_dict(sd)
    os.environ["HUDIFF_X3"] = "1"
    mx3 = cls(**cfg); mx3.load_state_dict(sd)
    B = 32 if kind == "ab" else 64
    batch = E.eval_batch("huab348" if kind == "ab" else "vhh", B, row0=0)
    tokens = batch["tokens"].copy()
    for b in range(0, B, 2):                       # every other row half-way through its sample
        loc = batch["order"][b, :batch["T"][b] // 2]
        tokens[b, loc] = batch["truth"][b, loc]
    out = {"kind": kind, "rows_per_forward": B}
    n64 = 4
    ch64 = None if batch["chain"] is None else np.concatenate([batch["chain"][:n64], batch["chain"][B:B + n64]])
    for name, drop in (("dropout_off", "off"), ("dropout_on", "faithful")):
        kw = dict(dropout=drop, seed=99, row0=3, step=17)
        a = m32(tokens, batch["region"], batch["chain"], **kw)
        b = mx3(tokens, batch["region"], batch["chain"], **kw)
        c = dict(cfg) if drop == "faithful" else dict(cfg, dropout=0.0)
        dr = ho.Dropout("philox", seed=99, rows=np.arange(n64) + 3, step=17) if drop == "faithful" else None
        o64 = ho.OracleNet(kind, c, sd, dtype=np.float64)(tokens[:n64], batch["region"][:n64], ch64, dropout=dr)
        o32 = ho.OracleNet(kind, c, sd)(tokens[:n64], batch["region"][:n64], ch64, dropout=dr)
        out[name] = {"max_abs_logit": float(np.abs(o64).max()),
                     "fp32_hip_vs_f64": float(np.abs(a[:n64] - o64).max()), "x3_hip_vs_f64": float(np.abs(b[:n64] - o64).max()),
                     "fp32_cpu_vs_f64": float(np.abs(o32 - o64).max()), "x3_vs_fp32_hip_all_rows": float(np.abs(a - b).max()),
                     "rms_x3_vs_f64": float(np.sqrt(np.mean((b[:n64] - o64) ** 2))),
                     "rms_fp32_vs_f64": float(np.sqrt(np.mean((a[:n64] - o64) ** 2)))}
    # complete samples with identical noise
    Bs = 256
    big = E.eval_batch("huab348" if kind == "ab" else "vhh", Bs, row0=0)
    args = (big["tokens"], big["region"], big["chain"], big["order"], big["T"])
    t0 = time.perf_counter(); s32 = m32.sample(*args, seed=5, row0=0); t1 = time.perf_counter()
    sx3 = mx3.sample(*args, seed=5, row0=0); t2 = time.perf_counter()
    same_rows = int((s32 == sx3).all(1).sum())
    out["sample_256_rows"] = {"rows_with_identical_tokens": same_rows, "rows": Bs,
                              "tokens_equal_fraction": float((s32 == sx3).mean()),
                              "seconds_fp32": round(t1 - t0, 3), "seconds_x3": round(t2 - t1, 3)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
