#!/usr/bin/env python
"""Which evaluation sequences should an ANARCI run target first?  (ANARCI parity of the built-in slotter
is unpinned offline.)

    python scripts/numbering_ambiguity.py          (build container: reads /root/reference/data/...)

For every sequence of the reference's evaluation CSVs the slotter's decisions that are NOT forced by the framework
anchors are listed -- the places where ANARCI's HMM could legitimately choose differently:
  * framework columns deleted other than the germline gaps IMGT 10 / 73 (81, 82 for light chains);
  * residues inserted inside a framework stretch;
  * CDR1 / CDR2 longer than the IMGT positions (27-38 / 56-65) or CDR3 longer than 13 + 24 insertion slots;
  * the best chain class beating the runner-up by < 5 % of its score (H / K / L call);
  * N- or C-terminal truncation (first numbered position > 1, last < 127 / 128);
  * residues that fall outside the reference's slot tables (dropped by get_input_element, sample.py:107-131).
Output: tests/golden/numbering_review.json (names / indices only, no sequences) and a summary on stdout.
"""
import json
import os
import sys

import numpy as np
import pandas as pd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hudiff_amd import numbering as N  # noqa: E402
from hudiff_amd import tables as T  # noqa: E402

REF = "/root/reference/data"


def review(seq, want_cls):
    q = N._encode(seq.strip().upper())
    res = {c: N._align(q, c) for c in "HKL"}
    ranked = sorted(res, key=lambda c: -res[c][0])
    cls = ranked[0]
    score, cols, loops, inserts = res[cls]
    flags = []
    if (want_cls == "H") != (cls == "H"):
        flags.append(f"class {cls} (expected {'H' if want_cls == 'H' else 'K/L'})")
    if res[ranked[1]][0] > 0.95 * score and not (cls in "KL" and ranked[1] in "KL" and False):
        flags.append(f"class margin {cls}>{ranked[1]} {score - res[ranked[1]][0]} of {score}")
    germline_gaps = {10, 73} | ({81, 82} if cls != "H" else set())
    last = 128 if cls == "H" else 127
    fw = [c for c in list(range(1, 27)) + list(range(39, 56)) + list(range(66, 105)) + list(range(118, last + 1))]
    present = sorted(cols)
    if present:
        lo, hi = present[0], present[-1]
        missing = [c for c in fw if lo <= c <= hi and c not in cols and c not in germline_gaps]
        if missing:
            flags.append(f"framework deletions at {missing}")
        if lo > 1:
            flags.append(f"starts at IMGT {lo}")
        if hi < last:
            flags.append(f"ends at IMGT {hi}")
    if inserts:
        flags.append(f"framework insertions after {sorted(set(c for c, _ in inserts))}")
    for first, width, name in ((27, 12, "CDR1"), (56, 10, "CDR2"), (105, 13, "CDR3")):
        if first in loops:
            n = loops[first][1] - loops[first][0]
            cap = width + (2 * (12 if cls == "H" else 6) if name == "CDR3" else 0)
            if n > (width if name != "CDR3" else cap):
                flags.append(f"{name} length {n} > {width if name != 'CDR3' else cap} slots")
    d, _ = N.number_sequence_builtin(seq)
    table = T.HEAVY_POSITIONS_dict if cls == "H" else T.LIGHT_POSITIONS_dict
    lost = [k for k, v in d.items() if v != "-" and k not in table]
    if lost:
        flags.append(f"outside the slot tables: {lost}")
    return cls, flags


def main():
    sets = []
    hu = pd.read_csv(f"{REF}/antibody_eval_data/HuAb348_data/humanization_pair_data_filter.csv")
    for i, r in enumerate(hu.itertuples()):
        sets.append(("HuAb348", f"{r.type}:{r.name}", "H", r.h_seq)); sets.append(("HuAb348", f"{r.type}:{r.name}", "L", r.l_seq))
    hm = pd.read_csv(f"{REF}/antibody_eval_data/Humab25_data/parental_mouse.csv")
    for r in hm.itertuples():
        sets.append(("Humab25", str(r.name), "H", r.h_seq)); sets.append(("Humab25", str(r.name), "L", r.l_seq))
    pu = pd.read_csv(f"{REF}/antibody_eval_data/putative_data/humanization_pair152.csv")
    hcol = "h_seq" if "h_seq" in pu.columns else [c for c in pu.columns if "h" in c.lower() and "seq" in c.lower()][0]
    lcol = "l_seq" if "l_seq" in pu.columns else [c for c in pu.columns if "l" in c.lower() and "seq" in c.lower()][0]
    for i, r in pu.iterrows():
        sets.append(("putative152", str(i), "H", r[hcol])); sets.append(("putative152", str(i), "L", r[lcol]))
    for name, col in (("abnativ_select_vhh", "vhhseq"), ("nanobert_exp", None)):
        df = pd.read_csv(f"{REF}/nanobody_eval_data/{name}.csv")
        c = col or [x for x in df.columns if "seq" in x.lower()][0]
        for i, s in enumerate(df[c]):
            sets.append((name, str(i), "H", s))
    out, n = [], 0
    counts = {}
    for ds, name, chain, seq in sets:
        if not isinstance(seq, str):
            continue
        n += 1
        try:
            cls, flags = review(seq, chain)
        except Exception as e:          # NumberingError: not a variable domain
            cls, flags = "?", [f"not numbered: {e}"]
        if flags:
            out.append({"set": ds, "name": name, "chain": chain, "class": cls, "length": len(seq), "flags": flags})
            for f in flags:
                key = f.split(" ")[0] + " " + f.split(" ")[1]
                counts[key] = counts.get(key, 0) + 1
    path = os.path.join(ROOT, "tests", "golden", "numbering_review.json")
    json.dump({"sequences_reviewed": n, "flagged": out}, open(path, "w"), indent=0)
    print(f"{n} sequences reviewed, {len(out)} flagged -> {path}")
    for k, v in sorted(counts.items(), key=lambda kv: -kv[1]):
        print(f"  {v:5d}  {k}")


if __name__ == "__main__":
    main()
