"""The reference's evaluation sets as pre-slotted integer rows (hudiff_amd/data/real_rows.npz, written by
scripts/make_real_rows.py from data/antibody_eval_data/HuAb348_data/humanization_pair_data_filter.csv,
Humab25_data/parental_mouse.csv and data/nanobody_eval_data/abnativ_select_vhh.csv with the built-in IMGT slotter).

BASELINE.json's metric is quoted "on HuAb348": bench.py and the full-size GPU tests build their device batches from
these rows -- antibody a = global_row % n_antibodies, replica = global_row // n_antibodies, so a batch of 256 rows per
GPU cycles through the 348 pairs with distinct replica noise.  Masks, visiting order
and region ids come from the same code the CLIs use (hudiff_amd.inputs); T is ragged as in the real data.
"""
from __future__ import annotations

import os

import numpy as np

from . import inputs as I
from . import tables

_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "real_rows.npz")      # package data: ships with hudiff_amd
_cache = None

DATASETS = {"huab348": "ab", "humab25": "ab", "vhh": "nb"}


def available() -> bool:
    return os.path.exists(_PATH)


def load_rows():
    global _cache
    if _cache is None:
        if not available():
            raise FileNotFoundError(f"{_PATH} is missing: the evaluation rows are package data of hudiff_amd "
                                    "(regenerate with scripts/make_real_rows.py where the reference's CSVs exist)")
        _cache = dict(np.load(_PATH))
    return _cache


def sequences(dataset: str):
    """Raw sequences recovered from the slots (a sequence is its non-empty slots in order): [(vh, vl)] or [vhh]."""
    z = load_rows()
    toks = z[f"{dataset}_tokens"]
    if DATASETS[dataset] == "ab":
        return [I.untokenize_antibody(t) for t in toks]
    return [I.untokenize_nanobody(t) for t in toks]


def eval_batch(dataset: str, B: int, mode: str | None = None, row0: int = 0, seed: int = 2023):
    """Rows row0 .. row0+B of the cycled data set -> dict like hudiff_amd.synthetic.synthetic_batch.

    dataset 'huab348' | 'humab25' (mode 'finetune' | 'pretrain') or 'vhh' (mode 'plain' | 'inpaint')."""
    z = load_rows()
    kind = DATASETS[dataset]
    toks = z[f"{dataset}_tokens"].astype(np.int32)
    n = toks.shape[0]
    L = tables.AB_LEN if kind == "ab" else tables.H_LEN
    mode = mode or ("finetune" if kind == "ab" else "plain")
    tokens = np.zeros((B, L), np.int32)
    truth = np.zeros((B, L), np.int32)
    region = np.zeros((B, L), np.int32)
    chain = np.zeros(2 * B, np.int32) if kind == "ab" else None
    source = np.zeros(B, np.int32)
    orders, Ts = [], []
    for r in range(B):
        g = row0 + r
        a = g % n
        source[r] = a
        if kind == "ab":
            tok, reg, ch, loc = I.antibody_row_from_tokens(toks[a], int(z[f"{dataset}_lchain"][a]), finetune=mode == "finetune")
            chain[r], chain[B + r] = ch
        else:
            tok, reg, loc = I.nanobody_row_from_tokens(toks[a], inpaint_sample=mode == "inpaint")
        loc = loc.copy()
        np.random.default_rng(seed + g).shuffle(loc)          # per-row visiting order (sample.py:497-498 shuffles per antibody)
        tokens[r], region[r], truth[r] = tok, reg, toks[a]
        orders.append(loc)
        Ts.append(len(loc))
    Tmax = max(Ts) if Ts else 0
    order = np.zeros((B, max(Tmax, 1)), np.int32)
    for r, loc in enumerate(orders):
        order[r, :len(loc)] = loc
    return dict(tokens=tokens, region=region, chain=chain, order=order, T=np.array(Ts, np.int32), truth=truth, mode=mode,
                source=source, dataset=dataset, n_sequences=n)
