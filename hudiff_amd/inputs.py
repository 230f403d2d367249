"""Input preparation of the samplers: IMGT-numbered residues -> slot tokens, mask, visiting order.

Mirrors (reference file:line) ``get_input_element`` / ``batch_input_element``
(antibody_scripts/sample.py:94-179) and their nanobody counterparts (nanobody_scripts/nanosample.py:91-149)
from the point AFTER numbering: the hot path starts from int arrays.  Numbering itself
(``get_pad_seq``: anarci.number + abnumber.Chain, sample.py:78-90) is the "next" row §8f-1; when
``abnumber``/``anarci`` are importable ``number_sequence`` uses them exactly as the reference does,
otherwise the built-in slotter (hudiff_amd/numbering.py) or pre-numbered residues supplied by the caller.
"""
from __future__ import annotations

import re
from typing import Dict, Optional, Tuple

import numpy as np

from . import tables as T
from .tokenizer import Tokenizer

_TK = Tokenizer()


def numbering_backend(choice: str = "auto") -> str:
    """'anarci' (the reference's own stack) when importable, else 'builtin' (hudiff_amd/numbering.py)."""
    if choice in ("anarci", "builtin"):
        return choice
    try:
        import anarci      # noqa: F401
        import abnumber    # noqa: F401
        return "anarci"
    except ImportError:
        return "builtin"


def number_sequence(aa_seq: str, backend: str = "auto") -> Tuple[Dict[str, str], str]:
    """sample.py:78-90 ``get_pad_seq``: raw sequence -> ({IMGT position: residue}, chain type 'H'|'K'|'L').

    backend 'anarci' runs the reference's third-party stack (ANARCI + HMMER, abnumber) exactly as the
    reference does; 'builtin' is the dependency-free slotter of hudiff_amd/numbering.py (parity with ANARCI
    unpinned); 'auto' prefers anarci when it is importable."""
    if numbering_backend(backend) == "builtin":
        from .numbering import number_sequence_builtin
        return number_sequence_builtin(aa_seq)
    try:
        from anarci import number
        from abnumber import Chain
    except ImportError as e:
        raise RuntimeError("--numbering anarci needs `anarci` and `abnumber` (not installed here)") from e
    seq_dict = {}
    results = number(aa_seq, scheme="imgt")
    for key, value in results[0]:
        seq_dict[str(key[0]) + key[1].strip()] = value
    return seq_dict, Chain(aa_seq, scheme="imgt").chain_type


def slot_residues(seq_dict: Dict[str, str], chain: str, quiet: bool = True):
    """Drop numbered residues into the 152 (heavy) / 139 (light) slots; unknown insertion codes are skipped
    (sample.py:107-131: printed and ignored by the reference as well)."""
    table = T.HEAVY_POSITIONS_dict if chain == "H" else T.LIGHT_POSITIONS_dict
    out = ["-"] * len(table)
    for key, value in seq_dict.items():
        idx = table.get(key)
        if idx is not None:
            out[idx] = value
        elif not quiet:
            n = int(re.findall(r"\d+", key)[0])
            if 27 <= n <= 38 or 56 <= n <= 65 or 105 <= n <= 117:
                print(("Heavy" if chain == "H" else "Light") + " CDR has problem.")
            else:
                print(("H" if chain == "H" else "L") + f" Position {key} is not in predefine dict, which can be ignored.")
    return out


def antibody_row(h_dict, l_dict, l_chain_type: str, finetune: bool = True, pad_region: int = 0):
    """-> tokens[291] (masked), region[291], chain (heavy id, light id), loc (maskable slots, ascending).
    sample.py:142-179 for one replica."""
    slots = slot_residues(h_dict, "H") + slot_residues(l_dict, "L")
    return antibody_row_from_tokens(_TK.seq2idx(slots), _TK.chain_type_idx(l_chain_type), finetune, pad_region)


def antibody_row_from_tokens(slot_tokens, l_chain_id: int, finetune: bool = True, pad_region: int = 0):
    """Same, from already slotted token ids [291] (21 = empty slot) and the light chain's type id."""
    tok = np.asarray(slot_tokens).astype(np.int64)
    if not finetune:
        mask = np.array(T.HEAVY_CDR_INDEX + T.LIGHT_CDR_INDEX) == 0
    else:
        mask = np.array(T.HEAVY_CDR_KABAT_NO_VERNIER + T.LIGHT_CDR_KABAT_NO_VERNIER) == 0
        mask = mask & ~((tok == _TK.idx_pad) & mask)          # framework gap slots are not sampled (:161-165)
    loc = np.arange(T.AB_LEN)[mask]
    tok = tok.copy()
    tok[mask] = _TK.idx_msk
    chain = (_TK.chain_type_idx("H"), int(l_chain_id))
    return tok.astype(np.int32), T.ab_region(pad_region).astype(np.int32), chain, loc


def graft_chain(aa_seq: str):
    """sample.py:209-226 ``graft_chain`` -- needs abnumber (its human-germline database does the grafting): CDRs of
    ``aa_seq`` on the closest human germline -> (numbered residues of the graft, identity position names, chain type).
    Without abnumber pre-grafted input can be supplied instead (``--grafted_fpath`` of hudiff_amd.cli.sample)."""
    try:
        from abnumber import Chain
    except ImportError as e:
        raise RuntimeError("CDR grafting needs `abnumber` (not installed here); pass --grafted_fpath with "
                           "pre-grafted numbered chains instead") from e
    seq_chain = Chain(aa_seq, scheme="imgt")
    grafted = seq_chain.graft_cdrs_onto_human_germline()
    align = seq_chain.align(grafted)
    identity = []
    for pos in align.positions:
        if pos.is_in_cdr():
            identity.append(str(pos)[1:])
        else:
            a1, a2 = align[pos]
            if a1 == a2:
                identity.append(str(pos)[1:])
    seq_dict, chain_type = number_sequence(grafted.seq, "anarci")
    return seq_dict, identity, chain_type


def cdr_pair_grafting(mouse_h_seq: str, mouse_l_seq: str, back_mutation: bool = False, scheme: str = "kabat"):
    """sample.py:370-376 (``--traditional_method``): plain CDR grafting of both chains onto their closest human
    germlines with abnumber, optionally back-mutating the Vernier zone.  No model involved."""
    try:
        from abnumber import Chain
    except ImportError as e:
        raise RuntimeError("--traditional_method is abnumber's CDR grafting (sample.py:370-376); `abnumber` is not "
                           "installed here") from e
    h = Chain(mouse_h_seq, scheme=scheme).graft_cdrs_onto_human_germline(backmutate_vernier=back_mutation)
    l = Chain(mouse_l_seq, scheme=scheme).graft_cdrs_onto_human_germline(backmutate_vernier=back_mutation)
    return h.seq, l.seq


def antibody_inpaint_row(h_dict, l_dict, identity_h, identity_l, l_chain_type: str, pad_region: int = 0):
    """``--sample_method inpaint`` after grafting (sample.py:229-310 for one replica): ``h_dict`` / ``l_dict`` are the
    numbered residues of the CDR-GRAFTED chains (mouse CDRs on the closest human germline, ``graft_chain``
    sample.py:209-226 -- abnumber's germline database, not part of this package) and ``identity_*`` the position names
    kept from the graft: all CDR positions plus the framework positions where graft and mouse agree.  Only those
    residues are placed; every other CDR-IMGT framework slot -- mismatches AND slots the graft leaves empty -- is masked
    and sampled (:291-299).  -> tokens[291] (masked), region[291], chain ids, loc."""
    keep_h, keep_l = set(identity_h), set(identity_l)
    slots = slot_residues({k: v for k, v in h_dict.items() if k in keep_h}, "H") + \
        slot_residues({k: v for k, v in l_dict.items() if k in keep_l}, "L")
    tok = _TK.seq2idx(slots)
    mask = (np.array(T.HEAVY_CDR_INDEX + T.LIGHT_CDR_INDEX) == 0) & (tok == _TK.idx_pad)
    loc = np.arange(T.AB_LEN)[mask]
    tok = tok.copy()
    tok[mask] = _TK.idx_msk
    chain = (_TK.chain_type_idx("H"), _TK.chain_type_idx(l_chain_type))
    return tok.astype(np.int32), T.ab_region(pad_region).astype(np.int32), chain, loc


def nanobody_row(h_dict, inpaint_sample: bool = False):
    """nanosample.py:124-149 for one replica -> tokens[152], region[152], loc."""
    return nanobody_row_from_tokens(_TK.seq2idx(slot_residues(h_dict, "H")), inpaint_sample)


def nanobody_row_from_tokens(slot_tokens, inpaint_sample: bool = False):
    """Same, from already slotted token ids [152] (21 = empty slot)."""
    tok = np.asarray(slot_tokens).astype(np.int64)
    table = np.array(T.INPAINT_HEAVY_CDR_INDEX if inpaint_sample else T.HEAVY_CDR_INDEX)
    mask = (table == 0) & (tok != _TK.idx_pad)
    loc = np.arange(T.H_LEN)[mask]
    tok = tok.copy()
    tok[mask] = _TK.idx_msk
    return tok.astype(np.int32), T.nb_region().astype(np.int32), loc


def untokenize_antibody(row) -> Tuple[str, str]:
    row = np.asarray(row)
    return _TK.idx2seq(row[:T.H_LEN]), _TK.idx2seq(row[T.H_LEN:])


def untokenize_nanobody(row) -> str:
    return _TK.idx2seq(np.asarray(row))


def slot_identity(ref_tokens, tokens) -> float:
    """Identity of a sample to its parental sequence over the aligned IMGT slots (both rows use the same
    slot layout, so no alignment is needed): equal / positions where either has a residue.  Stand-in for
    ``cal_all_preservation`` on abnumber alignments (antibody_scripts/patent_eval.py:150-159)."""
    a, b = np.asarray(ref_tokens), np.asarray(tokens)
    pos = (a != _TK.idx_pad) | (b != _TK.idx_pad)
    return float(((a == b) & pos).sum()) / float(max(int(pos.sum()), 1))
