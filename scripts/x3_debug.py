"""Which GEMM family breaks under HUDIFF_X3?  One process per mask value (the mask is read once per process)."""
import os, sys, subprocess, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 2:
    kind, mask = sys.argv[1], sys.argv[2]
    import hudiff_amd
    from hudiff_amd import synthetic as S
    cfg = dict(S.AB_CONFIG if kind == "ab" else S.NB_CONFIG)
    sd = S.random_state_dict(kind, cfg, seed=21)
    B = 32 if kind == "ab" else 56
    batch = S.synthetic_batch(kind, B, seed=9)
    cls = hudiff_amd.AntiTFNet if kind == "ab" else hudiff_amd.NanoAntiTFNet
    os.environ["HUDIFF_X3"] = "0"
    m0 = cls(**cfg); m0.load_state_dict(sd)
    os.environ["HUDIFF_X3"] = "1"
    m1 = cls(**cfg); m1.load_state_dict(sd)
    for drop in ("off", "faithful"):
        a = m0(batch["tokens"], batch["region"], batch["chain"], dropout=drop, seed=5, row0=100, step=3)
        b = m1(batch["tokens"], batch["region"], batch["chain"], dropout=drop, seed=5, row0=100, step=3)
        d = np.abs(a - b)
        print(kind, "mask", mask, drop, "max", float(d.max()), "nan", int(np.isnan(b).sum()), "worst row", int(d.max(axis=(1, 2)).argmax()),
              "rows>1e-4", int((d.max(axis=(1, 2)) > 1e-4).sum()), flush=True)
else:
    for kind in ("nb", "ab"):
        for mask in (1, 2, 4, 8, 16, 63):
            env = dict(os.environ, HUDIFF_X3_MASK=str(mask))
            subprocess.run([sys.executable, __file__, kind, str(mask)], env=env)
