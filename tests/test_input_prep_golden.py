"""hudiff_amd.inputs against the REFERENCE's own ``batch_input_element`` (antibody FR finetune /
pretrain, nanobody plain / inpaint) and ``batch_inpaint_input_element`` (antibody inpaint after grafting).

tests/golden/input_prep.json is written by oracle/make_golden_inputs.py, which runs the reference's functions
(antibody_scripts/sample.py:142-179, 283-310; nanobody_scripts/nanosample.py:124-149) with only the numbering /
grafting calls in front of them fed from the `input` dictionaries.  Integer work: bit-exact."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN
from hudiff_amd import inputs as I


@pytest.fixture(scope="module")
def cases():
    return json.load(open(os.path.join(GOLDEN, "input_prep.json")))["cases"]


def _same(got_tok, got_reg, got_loc, want):
    assert got_tok.tolist() == want["tokens"]
    assert got_reg.tolist() == want["region"]
    assert [int(x) for x in got_loc] == want["loc"]


def test_antibody_rows_match_reference(cases):
    n = 0
    for c in cases:
        if c["kind"] != "ab":
            continue
        inp = c["input"]
        for key, want in c["expect"].items():
            mode, pad = key.split("_pad")
            if mode == "inpaint":
                tok, reg, chain, loc = I.antibody_inpaint_row(inp["h"], inp["l"], inp["identity_h"], inp["identity_l"],
                                                               inp["l_chain"], pad_region=int(pad))
                B = 2
            else:
                tok, reg, chain, loc = I.antibody_row(inp["h"], inp["l"], inp["l_chain"], finetune=mode == "finetune",
                                                      pad_region=int(pad))
                B = 3
            _same(tok, reg, loc, want)
            # chain tensor of the reference: [heavy id] * B + [light id] * B  (sample.py:174-175)
            assert want["chain"] == [chain[0]] * B + [chain[1]] * B, (c["name"], key)
            assert want["batch"] == [0] * B + [1] * B
            assert (tok == 22).sum() == len(loc) and (tok[loc] == 22).all()
            n += 1
    assert n >= 5 * 12


def test_nanobody_rows_match_reference(cases):
    n = 0
    for c in cases:
        if c["kind"] != "nb":
            continue
        for mode, want in c["expect"].items():
            tok, reg, loc = I.nanobody_row(c["input"]["h"], inpaint_sample=mode == "inpaint")
            _same(tok, reg, loc, want)
            n += 1
    assert n >= 2 * 9


def test_unknown_position_messages_match_reference(cases, capsys):
    """The reference prints (and otherwise ignores) residues whose IMGT position is not in its slot tables
    (sample.py:111-131); slot_residues(quiet=False) says the same things in the same order."""
    c = next(c for c in cases if c["name"] == "unknown_insertions")
    capsys.readouterr()
    I.slot_residues(c["input"]["h"], "H", quiet=False)
    I.slot_residues(c["input"]["l"], "L", quiet=False)
    got = [ln for ln in capsys.readouterr().out.splitlines() if ln]
    want = [ln for ln in c["reference_stdout"] if ln and not ln.isdigit()]
    # the reference ran finetune/pretrain x pad 0/7 (4 identical passes) + the inpaint pass, which logs instead of printing
    per_pass = len(want) // 4
    assert per_pass >= 4 and want[:per_pass] == want[per_pass:2 * per_pass]
    assert got == want[:per_pass]
