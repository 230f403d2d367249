"""Batched driver of the device-side sampling loop for many independent sequences.

The reference humanizes one antibody at a time with ``batch_size`` replicas sharing one visiting order
(antibody_scripts/sample.py:475-538).  Here every (antibody, replica) is one independent row of a large
device batch: rows carry their own order and length, noise is keyed by the global row id, and rows are
sharded over ranks (hudiff_amd/dist.py).  Visiting orders are produced exactly as the reference does --
one ``np.random.shuffle(loc)`` per sequence in file order from the globally seeded numpy RNG
(sample.py:497-498, utils/misc.py:27-31).
"""
from __future__ import annotations

import random
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np

from . import dist as D


def seed_all(seed: int):
    """utils/misc.py:27-31 (torch's generator is not used by this implementation)."""
    np.random.seed(seed)
    random.seed(seed)


@dataclass
class Job:
    """One sequence to humanize: masked tokens, region ids, chain ids (antibody) and its visiting order."""
    tokens: np.ndarray
    region: np.ndarray
    loc: np.ndarray                      # already shuffled (or not) by the caller
    chain: Optional[tuple] = None        # (heavy id, light id) for antibodies
    name: str = ""
    parent: dict = field(default_factory=dict)


def sample_jobs(model, jobs: Sequence[Job], replicas: int, seed: int, *, passes: int = 1,
                device_batch: int = 256, dropout: str = "faithful", q_noise=None) -> np.ndarray:
    """Sample ``replicas`` rows per job; returns int32 [len(jobs), passes, replicas, L] on rank 0 (every rank
    when single-process).  ``passes`` > 1 re-runs the loop over the already filled tokens, which is what the
    reference's ``while sample_number > 0`` loop does (sample.py:499, nanosample.py:316)."""
    L = model.max_len
    n_rows = len(jobs) * replicas
    rank, world, _ = D.env_rank_world()
    lo, hi = D.shard_bounds(n_rows, rank, world)
    is_ab = model.kind == "ab"
    Tmax = max([len(j.loc) for j in jobs] + [1])
    out = np.zeros((passes, hi - lo, L), np.int32)
    for s in range(lo, hi, device_batch):
        e = min(s + device_batch, hi)
        ids = np.arange(s, e)
        jb = [jobs[i // replicas] for i in ids]
        tok = np.stack([j.tokens for j in jb]).astype(np.int32)
        reg = np.stack([j.region for j in jb]).astype(np.int32)
        order = np.zeros((len(jb), Tmax), np.int32)
        T = np.zeros(len(jb), np.int32)
        for r, j in enumerate(jb):
            order[r, :len(j.loc)] = j.loc
            T[r] = len(j.loc)
        chain = np.array([j.chain[0] for j in jb] + [j.chain[1] for j in jb], np.int32) if is_ab else None
        for p in range(passes):
            tok = model.sample(tok, reg, chain, order, T, seed=seed + 1000003 * p, row0=s, dropout=dropout,
                               q_noise=None if q_noise is None else q_noise[p][:, s:e])
            out[p, s - lo:e - lo] = tok
    gathered = [D.gather_rows(out[p], n_rows, L) for p in range(passes)]
    if gathered[0] is None:
        return None
    return np.stack(gathered, axis=0).reshape(passes, len(jobs), replicas, L).transpose(1, 0, 2, 3)
