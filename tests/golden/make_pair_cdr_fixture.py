"""TEST INFRASTRUCTURE -- f-1 evidence held by the reference's own DATA (VERDICT r2 "Next" #6b).

    python tests/golden/make_pair_cdr_fixture.py        (build container: reads /root/reference/data)

HuAb348 ships every mouse antibody WITH its experimentally humanized partner (humanization_pair_data_filter.csv: rows
`type == mouse` and the row that follows with the same order index, `type != mouse`).  Humanization by CDR grafting keeps
the CDR residues and replaces frameworks, so under a CONSISTENT numbering the two partners must carry (nearly) the same
residues in the same CDR slots -- whatever the frameworks look like.  ANARCI cannot be run offline; this check cannot
prove the built-in slotter equal to it, but a slotter that shifts a loop in either partner shows up here as a pair whose
slot-for-slot CDR identity collapses.

Written (data only): the humanized partners slotted by hudiff_amd.numbering (int8 tokens [348, 291], chain type of the light
chain), next to the mouse rows of hudiff_amd/data/real_rows.npz; and pair_cdr_review.json, the pairs an ANARCI run should
look at first.  The identities themselves are recomputed by tests/test_numbering_pairs.py.

Yard-stick without any numbering: a Needleman-Wunsch alignment of the two raw chains.  A pair is a "graft pair" when >= 95 %
of the mouse residues in Kabat-CDR slots are aligned to an identical partner residue; for such a pair the slot-for-slot CDR
identity must be >= 95 % as well (59 of the 348 pairs are not grafts in this sense: re-engineered CDRs, swapped chains).
"""
import os
import sys

import numpy as np
import pandas as pd

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hudiff_amd import inputs as I  # noqa: E402
from hudiff_amd.numbering import number_sequence_builtin  # noqa: E402

CSV = "/root/reference/data/antibody_eval_data/HuAb348_data/humanization_pair_data_filter.csv"
OUT = os.path.join(ROOT, "tests", "golden", "huab348_humanized_partners.npz")


def nw_pairs(a, b):
    """Global alignment (match +1, mismatch -1, gap -1) of two id sequences -> {index in a: aligned index in b}."""
    n, m = len(a), len(b)
    A, B = np.asarray(a), np.asarray(b)
    S = np.zeros((n + 1, m + 1), np.int32)
    S[:, 0] = -np.arange(n + 1)
    S[0, :] = -np.arange(m + 1)
    for i in range(1, n + 1):
        row = np.maximum(S[i - 1, :-1] + np.where(B == A[i - 1], 1, -1), S[i - 1, 1:] - 1)
        for j in range(1, m + 1):
            left = S[i, j - 1] - 1
            S[i, j] = row[j - 1] if row[j - 1] >= left else left
    i, j, pairs = n, m, {}
    while i > 0 or j > 0:
        if i > 0 and j > 0 and S[i, j] == S[i - 1, j - 1] + (1 if a[i - 1] == b[j - 1] else -1):
            pairs[i - 1] = j - 1
            i, j = i - 1, j - 1
        elif i > 0 and S[i, j] == S[i - 1, j] - 1:
            i -= 1
        else:
            j -= 1
    return pairs


def pair_cdr_numbers(mouse_tokens, human_tokens):
    """-> (CDR residues of the mouse pair, slot-for-slot identity, alignment identity, fraction aligned into the SAME slot)
    over the Kabat CDR 1-3 slots of both chains."""
    from hudiff_amd import tables as T
    kab = np.array(T.HEAVY_CDR_KABAT_NO_VERNIER + T.LIGHT_CDR_KABAT_NO_VERNIER)
    cdr = (kab >= 1) & (kab <= 3)
    tot = slot_match = same_res = same_slot = 0
    for lo, hi in ((0, T.H_LEN), (T.H_LEN, T.AB_LEN)):
        ms = [s for s in range(lo, hi) if mouse_tokens[s] != 21]
        hs = [s for s in range(lo, hi) if human_tokens[s] != 21]
        a, b = [int(mouse_tokens[s]) for s in ms], [int(human_tokens[s]) for s in hs]
        exact = all(mouse_tokens[s] == human_tokens[s] for s in ms if cdr[s])
        pr = None if exact else nw_pairs(a, b)
        for i, s in enumerate(ms):
            if not cdr[s]:
                continue
            tot += 1
            slot_match += mouse_tokens[s] == human_tokens[s]
            if exact:                       # identical residue in the identical slot: both yard-sticks agree trivially
                same_res += 1
                same_slot += 1
            elif i in pr:
                same_res += a[i] == b[pr[i]]
                same_slot += hs[pr[i]] == s
    return tot, slot_match / tot, same_res / tot, same_slot / tot


def main():
    df = pd.read_csv(CSV)
    mouse = df[df["type"] == "mouse"].reset_index(drop=True)
    human = df[df["type"] != "mouse"].reset_index(drop=True)
    assert len(mouse) == len(human) == 348
    toks, lch, names = [], [], []
    for m, h in zip(mouse.itertuples(), human.itertuples()):
        assert m.order_name.split("_")[0] == h.order_name.split("_")[0], (m.order_name, h.order_name)
        hd, ht = number_sequence_builtin(h.h_seq)
        ld, lt = number_sequence_builtin(h.l_seq)
        assert ht == "H" and lt in "KL"
        toks.append(np.array(I._TK.seq2idx(I.slot_residues(hd, "H") + I.slot_residues(ld, "L"))).astype(np.int8))
        lch.append(I._TK.chain_type_idx(lt))
        names.append(f"{m.name}|{h.name}")
    np.savez_compressed(OUT, tokens=np.stack(toks), lchain=np.array(lch, np.int8), names=np.array(names))
    print(OUT, os.path.getsize(OUT) // 1024, "KiB")
    import json
    from hudiff_amd import evalsets as E
    M = E.load_rows()["huab348_tokens"]
    review = []
    for i in range(348):
        tot, slot_id, nw_id, same_slot = pair_cdr_numbers(M[i], toks[i])
        if nw_id >= 0.95 and (slot_id < 1.0 and same_slot < 1.0):
            review.append({"pair": names[i], "cdr_residues": tot, "slot_identity": round(slot_id, 4), "alignment_identity": round(nw_id, 4),
                           "aligned_into_same_slot": round(same_slot, 4),
                           "why": "graft pair whose CDR residues do not all sit in the same slots in both partners (loop lengths differ, or "
                                  "the slotter placed one partner's loop differently): compare with ANARCI here first"})
    json.dump({"source": "tests/golden/make_pair_cdr_fixture.py over HuAb348 humanization_pair_data_filter.csv", "pairs": review},
              open(os.path.join(ROOT, "tests", "golden", "pair_cdr_review.json"), "w"), indent=1)
    print("review list:", len(review), "pairs")


if __name__ == "__main__":
    main()
