#!/usr/bin/env python
"""Repeated complete samples at small batches (the three-stage small tiles with loader waves, the sliced tail, the self-advancing draw): every repetition of a
batch size must give the same tokens, and the same tokens as the separate-launch / two-stage settings of a child process.
    python scripts/small_soak.py [ab|nb] [reps]"""
import hashlib, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(kind, reps):
    import hudiff_amd
    from hudiff_amd import synthetic as S, evalsets as E
    cfg = dict(S.AB_CONFIG if kind == "ab" else S.NB_CONFIG)
    m = (hudiff_amd.AntiTFNet if kind == "ab" else hudiff_amd.NanoAntiTFNet)(**cfg)
    m.load_state_dict(S.random_state_dict(kind, cfg, seed=0))
    for B in (1, 3, 8, 16, 40):
        b = E.eval_batch("huab348" if kind == "ab" else "vhh", B, row0=11)
        digs = set()
        for r in range(reps):
            tok = m.sample(b["tokens"], b["region"], b["chain"], b["order"], b["T"], seed=5, row0=0)
            digs.add(hashlib.sha256(tok.tobytes()).hexdigest()[:16])
        print(kind, "B", B, "reps", reps, "digests", sorted(digs), m.precision_info()["lnsync_fallbacks"], m.precision_info()["range_fallbacks"], flush=True)


if __name__ == "__main__":
    kind = sys.argv[1] if len(sys.argv) > 1 else "ab"
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    if os.environ.get("SMALL_SOAK_CHILD"):
        run(kind, reps)
        sys.exit(0)
    for name, env in (("default", {}), ("two stages, no loader waves, separate tail launches, one attention workgroup", dict(HUDIFF_X3_TINY_NS="2", HUDIFF_X3_SMALL_NS="2", HUDIFF_X3_LOADERS="0", HUDIFF_TAIL="0", HUDIFF_ATTN_QSPLIT_MAX="0"))):
        print("settings:", name, flush=True)
        subprocess.run([sys.executable, os.path.abspath(__file__), kind, str(reps if not env else 2)], env=dict(os.environ, SMALL_SOAK_CHILD="1", **env), check=True)
