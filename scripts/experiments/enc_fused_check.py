#!/usr/bin/env python
"""Fused token-encoder stack (hd_enc_fused.hip.h) against the per-GEMM launches: the stack's output (FEAT columns 0..255 after
hd_debug_stop_after(1)) and complete forwards, in two processes (HUDIFF_ENC_FUSED is read once).  python scripts/enc_fused_check.py [ab|nb] [B]"""
import os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(kind, B, out):
    import hudiff_amd
    from hudiff_amd import synthetic as S, evalsets as E
    cfg = dict(S.AB_CONFIG if kind == "ab" else S.NB_CONFIG)
    m = (hudiff_amd.AntiTFNet if kind == "ab" else hudiff_amd.NanoAntiTFNet)(**cfg, precision="split")
    m.load_state_dict(S.random_state_dict(kind, cfg, seed=0))
    b = E.eval_batch("huab348" if kind == "ab" else "vhh", B, row0=3)
    res = {}
    for drop in ("off", "faithful"):
        kw = dict(dropout=drop, seed=9, row0=5, step=2)
        m.debug_stop_after(1)
        m(b["tokens"], b["region"], b["chain"], **kw)
        res["feat_" + drop] = m.debug_read("FEAT", B)[:, :, :256]
        m.debug_stop_after(0)
        t0 = time.perf_counter()
        res["logits_" + drop] = m(b["tokens"], b["region"], b["chain"], **kw)
        res["t_" + drop] = time.perf_counter() - t0
    T = np.minimum(b["T"], 4)
    res["tokens"] = m.sample(b["tokens"], b["region"], b["chain"], b["order"], T, seed=4, row0=0)
    res["info"] = str(m.precision_info())
    np.savez(out, **res)


if __name__ == "__main__":
    if len(sys.argv) > 3:
        child(sys.argv[1], int(sys.argv[2]), sys.argv[3])
        sys.exit(0)
    kind = sys.argv[1] if len(sys.argv) > 1 else "ab"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    outs = []
    for f in ("0", "1"):
        out = f"/tmp/enc_fused_{kind}_{f}.npz"
        subprocess.run([sys.executable, os.path.abspath(__file__), kind, str(B), out], env=dict(os.environ, HUDIFF_ENC_FUSED=f), check=True)
        outs.append(np.load(out))
    a, b = outs
    for k in ("feat_off", "feat_faithful", "logits_off", "logits_faithful"):
        d = np.abs(a[k] - b[k])
        print(kind, B, k, "max |diff|", float(d.max()), "max |value|", float(np.abs(a[k]).max()), "finite", bool(np.isfinite(b[k]).all()),
              "worst at", np.unravel_index(int(d.argmax()), d.shape))
    d = np.abs(a["feat_off"] - b["feat_off"]).max(-1)          # [B, L]
    for bb in range(min(B, 3)):
        bad = np.nonzero(d[bb] > 1e-4)[0]
        print(kind, "seq", bb, "slots with |diff| > 1e-4:", bad.tolist()[:60], "max", float(d[bb].max()))
    print(kind, B, "tokens equal rows", int((a["tokens"] == b["tokens"]).all(1).sum()), "of", B, "| info", b["info"])
    print(kind, B, "forward seconds unfused / fused:", float(a["t_faithful"]), float(b["t_faithful"]))
