"""Slot / mask / region tables and tokenizer against fixtures extracted from the reference."""
import numpy as np
import pytest

from conftest import load_golden
from hudiff_amd import tables
from hudiff_amd.tokenizer import Tokenizer


def test_tables_match_reference_fixture():
    z = load_golden("tables.npz")
    assert list(z["heavy_positions"]) == tables.HEAVY_POSITIONS
    assert list(z["light_positions"]) == tables.LIGHT_POSITIONS
    for name in ("HEAVY_CDR_INDEX", "LIGHT_CDR_INDEX", "HEAVY_CDR_KABAT_NO_VERNIER", "LIGHT_CDR_KABAT_NO_VERNIER",
                 "INPAINT_HEAVY_CDR_INDEX", "HEAVY_REGION_INDEX", "LIGHT_REGION_INDEX"):
        assert z[name.lower()].tolist() == getattr(tables, name), name
    # SURVEY.md App. A maskable counts
    count0 = lambda t: sum(1 for v in t if v == 0)
    assert (count0(tables.HEAVY_CDR_INDEX), count0(tables.LIGHT_CDR_INDEX)) == (93, 92)
    assert (count0(tables.HEAVY_CDR_KABAT_NO_VERNIER), count0(tables.LIGHT_CDR_KABAT_NO_VERNIER)) == (79, 78)
    assert count0(tables.INPAINT_HEAVY_CDR_INDEX) == 87
    assert tables.HEAVY_POSITIONS_dict["112A"] == 134 and tables.LIGHT_POSITIONS_dict["127"] == 138


def test_tokenizer_roundtrip():
    tk = Tokenizer()
    assert tk.n_toks == 23 and tk.idx_pad == 21 and tk.idx_msk == 22 and tk.tok2idx("X") == 20
    for seq in ("EVQLVESGGGLVQPGGSLRLSCAAS", "ACDEFGHIKLMNPQRSTVWYX"):
        ids = tk.seq2idx(seq)
        assert tk.idx2seq(ids) == seq
    padded = list("AC--D") + ["<msk>"]
    ids = tk.seq2idx(padded)
    assert ids.tolist() == [0, 1, 21, 21, 2, 22]
    assert tk.idx2seq(ids) == "ACD<msk>" and tk.idx2seq_pad(ids) == "AC--D<msk>"
    with pytest.raises(KeyError):
        tk.seq2idx("AB")
    assert [tk.chain_type_idx(c) for c in "HLK"] == [0, 1, 2]
    with pytest.raises(TypeError):
        tk.chain_type_idx("Z")
