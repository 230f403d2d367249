"""TEST INFRASTRUCTURE ONLY -- input-preparation fixtures produced by the REFERENCE's own functions.

    python oracle/make_golden_inputs.py            (build container: needs /root/reference)

SURVEY.md §8 row a13 (``batch_input_element`` antibody_scripts/sample.py:142-179, nanobody_scripts/nanosample.py:124-149)
and the graft-independent half of row f-3 (``batch_inpaint_input_element`` sample.py:283-310).  The reference's sampler
modules are imported through oracle/ref_import.py; the only thing replaced is the third-party numbering call in front
of them (``get_pad_seq`` -> anarci / abnumber, sample.py:78-90; ``graft_chain`` -> abnumber germline grafting,
sample.py:209-226), which is fed from a table of pre-numbered residue dictionaries.  Everything downstream -- slot
placement, mask tables, the framework-gap rule, ``loc``, region / chain tensors -- is the reference's code running.

Numbered dictionaries come from (a) the built-in slotter on real rows of the reference's evaluation CSVs and
(b) hand-made edge cases (unknown insertion codes inside and outside CDRs, framework gaps, empty chains, lambda light
chains).  Output: tests/golden/input_prep.json -- data only (inputs + the reference's integer outputs).
"""
from __future__ import annotations

import contextlib
import importlib.util
import io
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import ref_import  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "input_prep.json")
REF = ref_import.REFERENCE_ROOT


def _load_module(name, relpath):
    ref_import.install()
    for extra in (os.path.join(REF, "antibody_scripts"), os.path.join(REF, "nanobody_scripts")):
        if extra not in sys.path:
            sys.path.append(extra)           # sample.py imports its sibling patent_eval
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, relpath))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)             # the CLI body sits under `if __name__ == '__main__'`
    return mod


def _cases():
    """[(name, kind, payload)] -- numbered inputs."""
    import pandas as pd
    from hudiff_amd.numbering import number_sequence_builtin
    cases = []
    hu = pd.read_csv(os.path.join(REF, "data/antibody_eval_data/HuAb348_data/humanization_pair_data_filter.csv"))
    mouse = hu[hu["type"] == "mouse"].reset_index(drop=True)
    for i in (0, 1, 7, 42, 100, 173, 250, 347):
        h, ht = number_sequence_builtin(mouse.loc[i, "h_seq"])
        l, lt = number_sequence_builtin(mouse.loc[i, "l_seq"])
        cases.append((f"huab348_{i}", "ab", {"h": h, "l": l, "h_chain": ht, "l_chain": lt}))
    hm = pd.read_csv(os.path.join(REF, "data/antibody_eval_data/Humab25_data/parental_mouse.csv"))
    for i in (0, 5, 24):
        h, ht = number_sequence_builtin(hm.loc[i, "h_seq"])
        l, lt = number_sequence_builtin(hm.loc[i, "l_seq"])
        cases.append((f"humab25_{i}", "ab", {"h": h, "l": l, "h_chain": ht, "l_chain": lt}))
    vhh = pd.read_csv(os.path.join(REF, "data/nanobody_eval_data/abnativ_select_vhh.csv"))
    for i in (0, 1, 2, 77, 150, 299):
        h, _ = number_sequence_builtin(vhh.loc[i, "vhhseq"])
        cases.append((f"vhh_{i}", "nb", {"h": h}))
    # ---- hand-made edge cases -------------------------------------------------------------------------
    base_h, base_l = dict(cases[0][2]["h"]), dict(cases[0][2]["l"])
    odd_h = dict(base_h)
    odd_h.update({"111M": "W", "60A": "W", "85A": "P", "112M": "Y"})          # not in HEAVY_POSITIONS_dict: CDR + framework
    odd_l = dict(base_l)
    odd_l.update({"111G": "W", "40A": "C"})                                    # light table stops at 111F
    cases.append(("unknown_insertions", "ab", {"h": odd_h, "l": odd_l, "h_chain": "H", "l_chain": "L"}))
    gap_h = {k: v for k, v in base_h.items() if k not in ("1", "2", "3", "45", "46", "83", "84", "85", "127", "128")}
    gap_l = {k: v for k, v in base_l.items() if k not in ("1", "10", "41", "70", "71", "72", "127")}
    cases.append(("framework_gaps", "ab", {"h": gap_h, "l": gap_l, "h_chain": "H", "l_chain": "K"}))
    cases.append(("empty_light", "ab", {"h": base_h, "l": {}, "h_chain": "H", "l_chain": "K"}))
    cases.append(("x_residues", "ab", {"h": {**base_h, "5": "X", "50": "X"}, "l": base_l, "h_chain": "H", "l_chain": "L"}))
    cases.append(("vhh_unknown_insertions", "nb", {"h": odd_h}))
    cases.append(("vhh_framework_gaps", "nb", {"h": gap_h}))
    cases.append(("vhh_empty", "nb", {"h": {}}))
    return cases


def _identity_list(seq_dict, rng, chain):
    """What ``graft_chain`` returns besides the grafted dict (sample.py:216-225): position names (IMGT, no chain
    letter) that are CDR positions or framework positions where the grafted germline equals the mouse residue.
    Here: every CDR-IMGT position plus a seeded ~70 % of the framework positions."""
    out = []
    for key in seq_dict:
        n = int("".join(c for c in key if c.isdigit()))
        in_cdr = 27 <= n <= 38 or 56 <= n <= 65 or 105 <= n <= 117
        if in_cdr or rng.random() < 0.7:
            out.append(key)
    return out


def main():
    sample = _load_module("ref_sample", "antibody_scripts/sample.py")
    nano = _load_module("ref_nanosample", "nanobody_scripts/nanosample.py")
    import logging
    sample.logger = logging.getLogger("ref_sample")        # get_inpaint_input logs through a global set in __main__
    table = {}

    def ab_get_pad_seq(key):
        return dict(table[key][0]), table[key][1]

    def nb_get_pad_seq(key):
        return dict(table[key][0])

    def graft_chain(key):
        d, ct, ident = table[key]
        return dict(d), list(ident), ct
    sample.get_pad_seq = ab_get_pad_seq
    nano.get_pad_seq = nb_get_pad_seq
    sample.graft_chain = graft_chain

    rng = np.random.default_rng(2024)
    records = []
    for name, kind, payload in _cases():
        rec = {"name": name, "kind": kind, "input": payload, "expect": {}}
        sink = io.StringIO()
        with contextlib.redirect_stdout(sink):
            if kind == "ab":
                table["H"] = (payload["h"], payload["h_chain"])
                table["L"] = (payload["l"], payload["l_chain"])
                for mode, finetune in (("finetune", True), ("pretrain", False)):
                    for pad_region in (0, 7):
                        tok, reg, chain, batch, loc, _ = sample.batch_input_element("H", "L", batch_size=3, pad_region=pad_region,
                                                                                    finetune=finetune)
                        assert (tok[0] == tok[2]).all() and (reg[0] == reg[1]).all()
                        rec["expect"][f"{mode}_pad{pad_region}"] = {
                            "tokens": tok[0].tolist(), "region": reg[0].tolist(), "chain": chain.tolist(),
                            "batch": batch.tolist(), "loc": [int(x) for x in loc]}
                ih, il = _identity_list(payload["h"], rng, "H"), _identity_list(payload["l"], rng, "L")
                rec["input"]["identity_h"], rec["input"]["identity_l"] = ih, il
                table["H"] = (payload["h"], payload["h_chain"], ih)
                table["L"] = (payload["l"], payload["l_chain"], il)
                tok, reg, chain, batch, loc, _ = sample.batch_inpaint_input_element("H", "L", batch_size=2, pad_region=0)
                rec["expect"]["inpaint_pad0"] = {"tokens": tok[0].tolist(), "region": reg[0].tolist(), "chain": chain.tolist(),
                                                 "batch": batch.tolist(), "loc": [int(x) for x in loc]}
            else:
                table["H"] = (payload["h"],)
                for mode, inpaint in (("plain", False), ("inpaint", True)):
                    tok, reg, loc, _ = nano.batch_input_element("H", inpaint_sample=inpaint, batch_size=2)
                    assert (tok[0] == tok[1]).all()
                    rec["expect"][mode] = {"tokens": tok[0].tolist(), "region": reg[0].tolist(), "loc": [int(x) for x in loc]}
        rec["reference_stdout"] = sink.getvalue().splitlines()
        records.append(rec)
    with open(OUT, "w") as f:
        json.dump({"source": "oracle/make_golden_inputs.py: antibody_scripts/sample.py batch_input_element / "
                             "batch_inpaint_input_element and nanobody_scripts/nanosample.py batch_input_element of the "
                             "reference, numbering calls fed from `input`",
                   "cases": records}, f, separators=(",", ":"))
    print(f"{OUT}: {len(records)} cases, {os.path.getsize(OUT) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
