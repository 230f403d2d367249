"""Errors of both HIP paths on the reference-generated adversarial-statistics vectors (tests/golden/adv_*.npz), as numbers
rather than pass / fail:  python scripts/adv_report.py [out.json]   (GPU box)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import hudiff_amd  # noqa: E402
from test_adversarial_golden import KINDS, VARIANTS, load_adv  # noqa: E402
from test_gpu_adversarial import _big_batch, _model  # noqa: E402

rows = []
for kind in KINDS:
    for variant in VARIANTS:
        z, cfg, sd = load_adv(kind, variant)
        st = dict(zip([str(n) for n in z["stat_names"]], z["stats"].tolist()))
        for path in ("f32_all", "f32", "x3"):
            m = _model(hudiff_amd, kind, cfg, sd, x3=(path == "x3"), attn_x3=(path != "f32_all"))
            B = 32 if kind == "ab" else 56
            fill, tokens, region, chain = _big_batch(kind, z, B)
            lg = m(tokens, region, chain, dropout="off")
            info = m.precision_info()
            m.close()
            rows.append({"kind": kind, "variant": variant, "path": path, "max_abs_dlogit_vs_reference_f32": float(np.abs(lg[:2] - z["logits"]).max()),
                         "max_abs_dlogit_vs_reference_f64": float(np.abs(lg[:2] - z["logits_f64"]).max()),
                         "reference_f32_vs_f64": float(z["reference_f32_vs_f64"]), "range_fallbacks": info["range_fallbacks"],
                         "max_mean_over_std_norm1_in": st["norm1_in"][0], "max_abs_x": max(v[1] for v in st.values())})
            print(rows[-1], flush=True)
if len(sys.argv) > 1:
    json.dump({"bound": 1e-4, "rows": rows}, open(sys.argv[1], "w"), indent=1)
