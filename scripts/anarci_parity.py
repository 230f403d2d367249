#!/usr/bin/env python
"""ANARCI / abnumber parity of the built-in IMGT slotter, one command away (VERDICT r3 "Next" #8).

    python scripts/anarci_parity.py [--data-root /path/to/HuDiff/data] [--out anarci_parity.json] [--limit N]

The reference numbers every input with ANARCI (``anarci.number(seq, scheme='imgt')``) and types it with ``abnumber.Chain``
(antibody_scripts/sample.py:78-90, nanobody_scripts/nanosample.py:75-88), and the nanobody sampler keeps a sample only if
``abnumber.Chain(seq, scheme='imgt')`` parses (nanosample.py:338-353).  Neither package can be installed in the build
container (no network, HMMER binary), so ``hudiff_amd/numbering.py`` is pinned by everything that can be produced offline
(tests/test_numbering.py, tests/test_numbering_pairs.py) and its agreement with ANARCI itself stays unpinned.  This script
closes that on ANY machine that has ``anarci`` + ``abnumber`` importable:

  1. numbering   every evaluation sequence -- the five CSVs of the reference's data/ directory when --data-root is given (2 368
                 chains), else the sequences held as package data (hudiff_amd/data/real_rows.npz: HuAb348, Humab25, the 300
                 VHHs) -- is numbered by both; per chain the chain type and the {IMGT position: residue} dictionaries are
                 compared, and the diff is reduced to what the model sees: the slot rows of hudiff_amd.inputs.slot_residues
                 (positions outside the reference's slot tables are dropped by both, sample.py:107-131).
                 The 55 sequences of tests/golden/numbering_review.json and the 4 pairs of tests/golden/pair_cdr_review.json --
                 where the framework anchors do not force the slotter's decision -- are reported FIRST.
  2. validity    the panel of tests/test_numbering.py::test_validity_predicate_panel (the 300 VHHs, framework re-samples of
                 caplacizumab, flanked domains, the rejected classes) through ``abnumber.Chain`` and through
                 ``numbering.is_variable_domain``: every disagreement is listed (the stand-in is known to be stricter on a lost
                 disulfide cysteine).

Output: one JSON file (default gpurun_out/anarci_parity.json) with counts, the review cases and every differing chain (names,
positions and residues; sequences only for differing chains), and a summary on stdout.  Exit code 0 = ran; the verdict is in the
file.  Without anarci / abnumber it exits with code 2 and says so -- it never fakes a comparison.
"""
from __future__ import annotations

import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def need_anarci():
    try:
        from anarci import number  # noqa: F401
        from abnumber import Chain  # noqa: F401
    except ImportError as e:
        sys.stderr.write(f"anarci_parity: {e}; this script needs `anarci` (with HMMER) and `abnumber` -- e.g. "
                         "`conda install -c bioconda anarci abnumber` -- and cannot run in the offline build container.\n")
        raise SystemExit(2)


def anarci_numbering(seq):
    """What the reference's get_pad_seq computes (sample.py:78-90) -> ({position: residue}, chain type) or (None, reason)."""
    from anarci import number
    from abnumber import Chain
    try:
        results = number(seq, scheme="imgt")
        if not results or not results[0]:
            return None, "anarci.number returned nothing"
        d = {str(key[0]) + key[1].strip(): value for key, value in results[0]}
        return d, Chain(seq, scheme="imgt").chain_type
    except Exception as e:                      # abnumber.ChainParseError, AssertionError inside anarci, ...
        return None, f"{type(e).__name__}: {e}"


def builtin_numbering(seq):
    from hudiff_amd import numbering as N
    try:
        return N.number_sequence_builtin(seq)
    except N.NumberingError as e:
        return None, f"NumberingError: {e}"


def slot_row(d, chain_type):
    """The model-visible form of a numbering: residues in the reference's slot tables (152 heavy / 139 light slots)."""
    from hudiff_amd import inputs as I
    return list(I.slot_residues(d, "H" if chain_type == "H" else "L", quiet=True))


def collect_sequences(data_root):
    """[(set, name, expected chain 'H' | 'L', sequence)] -- the reference's five evaluation CSVs, or the package data."""
    out = []
    if data_root:
        import pandas as pd
        hu = pd.read_csv(os.path.join(data_root, "antibody_eval_data", "HuAb348_data", "humanization_pair_data_filter.csv"))
        for r in hu.itertuples():
            out += [("HuAb348", f"{r.type}:{r.name}", "H", r.h_seq), ("HuAb348", f"{r.type}:{r.name}", "L", r.l_seq)]
        hm = pd.read_csv(os.path.join(data_root, "antibody_eval_data", "Humab25_data", "parental_mouse.csv"))
        for r in hm.itertuples():
            out += [("Humab25", str(r.name), "H", r.h_seq), ("Humab25", str(r.name), "L", r.l_seq)]
        pu = pd.read_csv(os.path.join(data_root, "antibody_eval_data", "putative_data", "humanization_pair152.csv"))
        hcol = "h_seq" if "h_seq" in pu.columns else [c for c in pu.columns if "h" in c.lower() and "seq" in c.lower()][0]
        lcol = "l_seq" if "l_seq" in pu.columns else [c for c in pu.columns if "l" in c.lower() and "seq" in c.lower()][0]
        for i, r in pu.iterrows():
            out += [("putative152", str(i), "H", r[hcol]), ("putative152", str(i), "L", r[lcol])]
        for name, col in (("abnativ_select_vhh", "vhhseq"), ("nanobert_exp", None)):
            df = pd.read_csv(os.path.join(data_root, "nanobody_eval_data", f"{name}.csv"))
            c = col or [x for x in df.columns if "seq" in x.lower()][0]
            out += [(name, str(i), "H", s) for i, s in enumerate(df[c])]
    else:
        from hudiff_amd import evalsets as E
        for ds, label in (("huab348", "HuAb348 (mouse rows, package data)"), ("humab25", "Humab25")):
            for i, (h, l) in enumerate(E.sequences(ds)):
                out += [(label, str(i), "H", h), (label, str(i), "L", l)]
        out += [("abnativ_select_vhh", str(i), "H", s) for i, s in enumerate(E.sequences("vhh"))]
    return [(a, b, c, s.strip().upper()) for a, b, c, s in out if isinstance(s, str) and s.strip()]


def compare_chain(seq):
    a, ta = anarci_numbering(seq)
    b, tb = builtin_numbering(seq)
    if a is None or b is None:
        return {"anarci": "ok" if a is not None else ta, "builtin": "ok" if b is not None else tb,
                "same": a is None and b is None}
    a_res = {k: v for k, v in a.items() if v != "-"}
    b_res = {k: v for k, v in b.items() if v != "-"}
    moved = sorted((k for k in set(a_res) | set(b_res) if a_res.get(k) != b_res.get(k)),
                   key=lambda k: (int("".join(c for c in k if c.isdigit()) or 0), k))
    same_type = ta == tb
    rows_equal = same_type and slot_row(a, ta) == slot_row(b, tb)
    return {"anarci": "ok", "builtin": "ok", "chain_type": [ta, tb], "same": same_type and not moved, "slot_rows_equal": bool(rows_equal),
            "positions_that_differ": [{"pos": k, "anarci": a_res.get(k, "-"), "builtin": b_res.get(k, "-")} for k in moved]}


def validity_panel():
    """(label, sequence, what the stand-in says) for the panel of tests/test_numbering.py::test_validity_predicate_panel."""
    from hudiff_amd import evalsets as E
    from hudiff_amd import numbering as N
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_numbering import KNOWN
    rng = np.random.default_rng(3)
    aa = "ACDEFGHIKLMNPQRSTVWY"
    panel = [(f"vhh[{i}]", s) for i, s in enumerate(E.sequences("vhh"))]
    cap = KNOWN["caplacizumab_VHH"][0]
    d, _ = N.number_sequence_builtin(cap)
    pos_of = [int("".join(c for c in k if c.isdigit())) for k in
              sorted(d, key=lambda k: (int("".join(c for c in k if c.isdigit())), k)) if d[k] != "-"]
    anchors = {23, 41, 104, 118, 119, 121}
    for n in range(40):
        s = list(cap)
        for i in rng.choice(len(s), size=12, replace=False):
            if pos_of[i] not in anchors and not (27 <= pos_of[i] <= 38 or 56 <= pos_of[i] <= 65 or 105 <= pos_of[i] <= 117):
                s[i] = aa[rng.integers(20)]
        panel.append((f"caplacizumab framework re-sample {n}", "".join(s)))
    panel.append(("pelB leader + caplacizumab + His tag", "MKYLLPTAAAGLLLLAAQPAMA" + cap + "HHHHHH"))
    half = len(cap) // 2
    panel += [("N-terminal half", cap[:half]), ("C-terminal half", cap[half:]), ("cut inside CDR3", cap[:-30]),
              ("second half out of frame", cap[:half] + "".join(aa[rng.integers(20)] for _ in range(len(cap) - half))),
              ("reversed", cap[::-1]), ("poly-alanine", "A" * 120), ("random", "".join(aa[rng.integers(20)] for _ in range(120))),
              ("short peptide", "EVQLVESGGG")]
    for p in (23, 104, 41, 118, 119):
        i = pos_of.index(p)
        panel.append((f"caplacizumab with IMGT {p} -> {'A' if p in (23, 104) else 'R'}", cap[:i] + ("A" if p in (23, 104) else "R") + cap[i + 1:]))
    return [(label, s, bool(N.is_variable_domain(s))) for label, s in panel]


def abnumber_parses(seq):
    from abnumber import Chain
    try:
        Chain(seq, scheme="imgt")
        return True
    except Exception:
        return False


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--data-root", default=None, help="the reference's data/ directory (all five evaluation CSVs); default: package data")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "anarci_parity.json"))
    ap.add_argument("--limit", type=int, default=0, help="number only the first N chains (smoke run)")
    ap.add_argument("--list-only", action="store_true", help="print what would be compared and exit (needs neither package)")
    args = ap.parse_args()
    seqs = collect_sequences(args.data_root)
    review = json.load(open(os.path.join(ROOT, "tests", "golden", "numbering_review.json")))["flagged"]
    pairs = json.load(open(os.path.join(ROOT, "tests", "golden", "pair_cdr_review.json")))["pairs"]
    review_keys = {(r["set"], r["name"], r["chain"]) for r in review}
    pair_names = {n for p in pairs for n in p["pair"].split("|")}
    first = [s for s in seqs if (s[0], s[1], s[2]) in review_keys or s[1].split(":")[-1] in pair_names]
    rest = [s for s in seqs if s not in first]
    if args.limit:
        first, rest = first[:args.limit], rest[:max(0, args.limit - len(first))]
    if args.list_only:
        print(f"{len(seqs)} chains ({len(first)} review cases first), validity panel of {len(validity_panel())} sequences")
        return 0
    need_anarci()
    out = {"chains": len(first) + len(rest), "review_cases": [], "differences": [], "validity": {}}
    n_same = n_rows_equal = 0
    for is_review, group in ((True, first), (False, rest)):
        for ds, name, chain, seq in group:
            c = compare_chain(seq)
            n_same += bool(c["same"])
            n_rows_equal += bool(c.get("slot_rows_equal", c["same"]))
            rec = {"set": ds, "name": name, "chain": chain, **c}
            if is_review:
                out["review_cases"].append(rec)
            if not c["same"]:
                out["differences"].append(dict(rec, sequence=seq))
    out["identical_numbering"] = n_same
    out["identical_slot_rows"] = n_rows_equal
    vp = validity_panel()
    dis = [{"case": label, "abnumber_parses": abnumber_parses(s), "builtin_is_variable_domain": ok, "sequence": s}
           for label, s, ok in vp]
    dis = [d for d in dis if d["abnumber_parses"] != d["builtin_is_variable_domain"]]
    out["validity"] = {"panel": len(vp), "disagreements": dis}
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    json.dump(out, open(args.out, "w"), indent=1)
    print(f"{out['chains']} chains: identical numbering {n_same}, identical model-visible slot rows {n_rows_equal}; "
          f"{len(out['review_cases'])} review cases, {sum(1 for r in out['review_cases'] if not r['same'])} of them differ; "
          f"validity panel {len(vp)}: {len(dis)} disagreements -> {args.out}")
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
