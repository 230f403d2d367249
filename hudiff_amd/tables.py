"""IMGT slot layout and mask / region tables of the HuDiff input contract, stated compactly.

The reference keeps these as literal lists (dataset/preprocess.py:195-362 slot dictionaries and CDR
index tables; dataset/oas_pair_dataset_new.py:25-40 region tables).  Here each table is a run-length
description expanded at import; ``tests/test_tables.py`` checks every entry against a fixture
extracted from the reference (tests/golden/tables.npz).

Heavy chain: 152 slots = IMGT 1..111, insertions 111A..111L, 112L..112A, 112..128.
Light chain: 139 slots = IMGT 1..111, insertions 111A..111F, 112F..112A, 112..127.
"""
from __future__ import annotations

import numpy as np

H_LEN = 152
L_LEN = 139
AB_LEN = H_LEN + L_LEN


def _imgt_slots(n_ins: int, last: int):
    letters = "ABCDEFGHIJKL"[:n_ins]
    names = [str(i) for i in range(1, 112)]
    names += ["111" + c for c in letters]
    names += ["112" + c for c in reversed(letters)]
    names += [str(i) for i in range(112, last + 1)]
    return names


HEAVY_POSITIONS = _imgt_slots(12, 128)
LIGHT_POSITIONS = _imgt_slots(6, 127)
HEAVY_POSITIONS_dict = {name: i for i, name in enumerate(HEAVY_POSITIONS)}
LIGHT_POSITIONS_dict = {name: i for i, name in enumerate(LIGHT_POSITIONS)}


def _expand(*runs):
    out = []
    for value, count in runs:
        out += [value] * count
    return out


# value 0 = framework (maskable), 1/2/3 = CDR1/2/3, 4/5 = other positions kept fixed
HEAVY_CDR_INDEX = _expand((0, 26), (1, 12), (0, 17), (2, 10), (0, 39), (3, 37), (0, 11))
LIGHT_CDR_INDEX = _expand((0, 26), (1, 12), (0, 17), (2, 10), (0, 39), (3, 25), (0, 10))
HEAVY_CDR_KABAT_NO_VERNIER = _expand((0, 26), (1, 14), (0, 14), (2, 20), (0, 30), (3, 37), (0, 9), (4, 2))
LIGHT_CDR_KABAT_NO_VERNIER = _expand((0, 23), (1, 17), (0, 11), (5, 4), (2, 14), (0, 35), (3, 25), (0, 9), (4, 1))
INPAINT_HEAVY_CDR_INDEX = _expand((0, 26), (1, 12), (0, 3), (4, 1), (0, 6), (4, 2), (0, 1), (4, 1), (0, 2),
                                  (2, 12), (0, 38), (3, 37), (0, 11))
# FR1 CDR1 FR2 CDR2 FR3 CDR3 FR4
HEAVY_REGION_INDEX = _expand((0, 26), (1, 12), (2, 17), (3, 10), (4, 39), (5, 37), (6, 11))
LIGHT_REGION_INDEX = _expand((0, 26), (1, 12), (2, 17), (3, 10), (4, 39), (5, 25), (6, 10))

assert len(HEAVY_POSITIONS) == len(HEAVY_CDR_INDEX) == len(HEAVY_CDR_KABAT_NO_VERNIER) == \
    len(INPAINT_HEAVY_CDR_INDEX) == len(HEAVY_REGION_INDEX) == H_LEN
assert len(LIGHT_POSITIONS) == len(LIGHT_CDR_INDEX) == len(LIGHT_CDR_KABAT_NO_VERNIER) == \
    len(LIGHT_REGION_INDEX) == L_LEN


def ab_region(pad_region: int = 0) -> np.ndarray:
    """Region ids of the 291 antibody slots; light ids are offset by pad_region (sample.py:462-465, 103-105)."""
    return np.array(HEAVY_REGION_INDEX + [r + pad_region for r in LIGHT_REGION_INDEX], dtype=np.int64)


def nb_region() -> np.ndarray:
    return np.array(HEAVY_REGION_INDEX, dtype=np.int64)
