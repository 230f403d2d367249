"""Amino-acid tokenizer with the reference's vocabulary (utils/tokenizer.py:43-149).

ids: 20 standard residues 'ACDEFGHIKLMNPQRSTVWY' -> 0..19, 'X' -> 20, '-' (gap / pad) -> 21, '<msk>' -> 22.
The sampler draws over ids 0..21 (sample.py:510), so 'X' and '-' can be emitted; ``idx2seq`` drops '-'.
"""
from __future__ import annotations

import numpy as np

RESIDUES = "ACDEFGHIKLMNPQRSTVWY"


class Tokenizer:
    def __init__(self, has_bos: bool = False, has_eos: bool = False):
        if has_bos or has_eos:
            raise NotImplementedError("the sampling path never uses <bos>/<eos>")
        self.tok_msk = "<msk>"
        self.tok_pad = "-"
        self.toks = [*RESIDUES, "X", self.tok_pad, self.tok_msk]
        self.tok2idx_dict = {t: i for i, t in enumerate(self.toks)}
        self.idx_msk = self.tok2idx_dict[self.tok_msk]
        self.idx_pad = self.tok2idx_dict[self.tok_pad]

    @property
    def n_toks(self) -> int:
        return len(self.toks)

    def tok2idx(self, tok: str) -> int:
        return self.tok2idx_dict[tok]

    def seq2idx(self, aa_seq) -> np.ndarray:
        """Sequence (str or list of tokens) -> int64 ids; unknown symbols raise KeyError like the reference."""
        return np.array([self.tok2idx_dict[x] for x in [*aa_seq]], dtype=np.int64)

    def idx2seq(self, idx_vec) -> str:
        ids = np.asarray(idx_vec).reshape(-1).tolist()
        return "".join(self.toks[i] for i in ids if i != self.idx_pad)

    def idx2seq_pad(self, idx_vec) -> str:
        return "".join(self.toks[i] for i in np.asarray(idx_vec).reshape(-1).tolist())

    def idx2seq_batch(self, idx_mat):
        return [self.idx2seq(row) for row in np.asarray(idx_mat)]

    def idx2seq_pad_batch(self, idx_mat):
        return [self.idx2seq_pad(row) for row in np.asarray(idx_mat)]

    @staticmethod
    def chain_type_idx(chain: str) -> int:
        try:
            return {"H": 0, "L": 1, "K": 2}[chain]
        except KeyError:
            raise TypeError("Chain Type has problem.")
