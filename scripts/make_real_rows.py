#!/usr/bin/env python
"""Numbers the reference's evaluation sets with the built-in IMGT slotter and writes INT fixtures.

    python scripts/make_real_rows.py                (build container: reads /root/reference/data/...)

BASELINE.json's configs are quoted on real rows -- HuAb348 ``humanization_pair_data_filter.csv`` (348 mouse pairs),
Humab25 ``parental_mouse.csv`` (25 pairs), ``abnativ_select_vhh.csv`` (300 VHH) -- which do not exist on the GPU box.
This script slots every sequence (hudiff_amd.numbering, the same front-end the CLIs use offline) and stores the
result as small integer arrays in hudiff_amd/data/real_rows.npz:

    huab348_tokens  int8 [348, 291]   slot tokens of VH (152) + VL (139), 21 = empty slot, nothing masked
    huab348_lchain  int8 [348]        light chain type id (1 = lambda, 2 = kappa; utils/tokenizer.py chain ids)
    humab25_tokens / humab25_lchain / humab25_names
    vhh_tokens      int8 [300, 152]

bench.py builds its batches from them (masks / loc / region follow from the tables), the GPU tests rebuild the
CSV inputs of the CLIs from them (a sequence is its non-empty slots in order).  Data only: inputs for our own code.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import pandas as pd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hudiff_amd import inputs as I  # noqa: E402
from hudiff_amd.numbering import number_sequence_builtin  # noqa: E402

REF = "/root/reference/data"
OUT = os.path.join(ROOT, "hudiff_amd", "data", "real_rows.npz")


def slot_pair(h_seq, l_seq):
    h, ht = number_sequence_builtin(h_seq)
    l, lt = number_sequence_builtin(l_seq)
    assert ht == "H" and lt in "KL", (ht, lt)
    tok = np.array(I._TK.seq2idx(I.slot_residues(h, "H") + I.slot_residues(l, "L")))
    # every residue found a slot: the sequence is recoverable from the fixture
    assert I.untokenize_antibody(tok) == (h_seq, l_seq), "a residue fell outside the slot tables"
    return tok.astype(np.int8), I._TK.chain_type_idx(lt)


def main():
    out = {}
    hu = pd.read_csv(os.path.join(REF, "antibody_eval_data/HuAb348_data/humanization_pair_data_filter.csv"))
    hu = hu[hu["type"] == "mouse"]
    hm = pd.read_csv(os.path.join(REF, "antibody_eval_data/Humab25_data/parental_mouse.csv"))
    hm = hm[hm["type"] == "mouse"]
    for key, df in (("huab348", hu), ("humab25", hm)):
        toks, lch, lossy = [], [], 0
        for line in df.itertuples():
            try:
                t, c = slot_pair(line.h_seq, line.l_seq)
            except AssertionError:
                # keep the row (the CLI would also run it) but note that its CSV text cannot be rebuilt exactly
                h, _ = number_sequence_builtin(line.h_seq)
                l, lt = number_sequence_builtin(line.l_seq)
                t = np.array(I._TK.seq2idx(I.slot_residues(h, "H") + I.slot_residues(l, "L"))).astype(np.int8)
                c = I._TK.chain_type_idx(lt)
                lossy += 1
            toks.append(t); lch.append(c)
        out[f"{key}_tokens"] = np.stack(toks)
        out[f"{key}_lchain"] = np.array(lch, np.int8)
        print(key, out[f"{key}_tokens"].shape, "rows whose raw text is not recoverable from the slots:", lossy)
    out["humab25_names"] = np.array([str(n) for n in hm["name"]])
    vhh = pd.read_csv(os.path.join(REF, "nanobody_eval_data/abnativ_select_vhh.csv"))
    toks, lossy = [], 0
    for seq in vhh["vhhseq"]:
        h, _ = number_sequence_builtin(seq)
        t = np.array(I._TK.seq2idx(I.slot_residues(h, "H"))).astype(np.int8)
        lossy += I.untokenize_nanobody(t) != seq
        toks.append(t)
    out["vhh_tokens"] = np.stack(toks)
    print("vhh", out["vhh_tokens"].shape, "rows whose raw text is not recoverable from the slots:", lossy)
    np.savez_compressed(OUT, **out)
    print(OUT, f"{os.path.getsize(OUT) / 1024:.0f} KiB")
    for key, fin in (("huab348", True), ("humab25", True)):
        T = [len(I.antibody_row_from_tokens(t, c, finetune=fin)[3]) for t, c in zip(out[f"{key}_tokens"], out[f"{key}_lchain"])]
        print(key, "finetune T: min", min(T), "max", max(T), "mean", round(float(np.mean(T)), 2))


if __name__ == "__main__":
    main()
