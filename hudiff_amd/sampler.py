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
    tokens: np.ndarray                   # [L], or [replicas, L] when replicas continue from different states
    region: np.ndarray
    loc: np.ndarray                      # already shuffled (or not) by the caller
    chain: Optional[tuple] = None        # (heavy id, light id) for antibodies
    name: str = ""
    parent: dict = field(default_factory=dict)


def _id_runs(gids: np.ndarray, max_rows: int):
    """Split positions 0..len(gids) into chunks of <= max_rows whose global row ids are consecutive (one device
    launch keys its rows as row0 + b)."""
    s = 0
    while s < len(gids):
        e = s + 1
        while e < len(gids) and e - s < max_rows and gids[e] == gids[e - 1] + 1:
            e += 1
        yield s, e
        s = e


def sample_jobs(model, jobs: Sequence[Job], replicas: int, seed: int, *, passes: int = 1,
                device_batch: int = 256, dropout: str = "faithful", q_noise=None, all_ranks: bool = False,
                job_ids: Optional[Sequence[int]] = None) -> np.ndarray:
    """Sample ``replicas`` rows per job; returns int32 [len(jobs), passes, replicas, L] on rank 0 (every rank
    when single-process or ``all_ranks``).  ``passes`` > 1 re-runs the loop over the already filled tokens, which is what the
    reference's ``while sample_number > 0`` loop does (sample.py:499, nanosample.py:316).

    ``job_ids``: the id each job's noise is keyed by (default: its position in ``jobs``).  Row (job, replica) draws
    its noise as global row ``job_ids[job] * replicas + replica``, whatever else is in the batch."""
    L = model.max_len
    n_rows = len(jobs) * replicas
    rank, world, _ = D.env_rank_world()
    lo, hi = D.shard_bounds(n_rows, rank, world)
    is_ab = model.kind == "ab"
    Tmax = max([len(j.loc) for j in jobs] + [1])
    out = np.zeros((passes, hi - lo, L), np.int32)
    jid = np.arange(len(jobs), dtype=np.int64) if job_ids is None else np.asarray(job_ids, dtype=np.int64)
    pos = np.arange(lo, hi)                                      # positions in the packed (job-major) row list
    gids = jid[pos // replicas] * replicas + pos % replicas      # the ids the noise is keyed by
    for cs, ce in _id_runs(gids, device_batch):
        ids = pos[cs:ce]
        s, e = int(ids[0]), int(ids[-1]) + 1
        jb = [jobs[i // replicas] for i in ids]
        tok = np.stack([j.tokens if np.ndim(j.tokens) == 1 else j.tokens[i % replicas]
                        for i, j in zip(ids, jb)]).astype(np.int32)
        reg = np.stack([j.region for j in jb]).astype(np.int32)
        order = np.zeros((len(jb), Tmax), np.int32)
        T = np.zeros(len(jb), np.int32)
        for r, j in enumerate(jb):
            order[r, :len(j.loc)] = j.loc
            T[r] = len(j.loc)
        chain = np.array([j.chain[0] for j in jb] + [j.chain[1] for j in jb], np.int32) if is_ab else None
        for p in range(passes):
            tok = model.sample(tok, reg, chain, order, T, seed=seed + 1000003 * p, row0=int(gids[cs]), dropout=dropout,
                               q_noise=None if q_noise is None else np.ascontiguousarray(q_noise[p][:Tmax, gids[cs:ce]]))
            out[p, s - lo:e - lo] = tok
    gathered = [D.gather_rows(out[p], n_rows, L, all_ranks) for p in range(passes)]
    if gathered[0] is None:
        return None
    return np.stack(gathered, axis=0).reshape(passes, len(jobs), replicas, L).transpose(1, 0, 2, 3)


def noise_in_reference_order(q_flat, jobs: Sequence[Job], replicas: int) -> np.ndarray:
    """Recorded ``torch.multinomial`` noise of a REFERENCE run -> the ``q_noise`` argument of ``sample_jobs``.

    The reference draws input row by input row, step by step, one ``[batch_size, 22]`` Exp(1) tensor per step
    (sample.py:499-513, nanosample.py:316-329): ``q_flat`` is that sequence, ``[sum_j T_j, replicas, 22]`` for ONE sweep over
    every input row.  Here row (job j, replica r) is global row ``j * replicas + r`` -> ``[1, Tmax, len(jobs) * replicas, 22]``."""
    q_flat = np.asarray(q_flat, np.float32)
    Ts = [len(j.loc) for j in jobs]
    if q_flat.shape != (sum(Ts), replicas, 22):
        raise ValueError(f"recorded noise has shape {q_flat.shape}; this input needs {(sum(Ts), replicas, 22)} "
                         "(one [batch_size, 22] draw per visited slot of every input row)")
    out = np.ones((1, max(Ts + [1]), len(jobs) * replicas, 22), np.float32)
    at = 0
    for j, T in enumerate(Ts):
        out[0, :T, j * replicas:(j + 1) * replicas] = q_flat[at:at + T]
        at += T
    return out


def sample_jobs_with_retry(model, jobs: Sequence[Job], replicas: int, seed: int, *, want: int, tries: int, accept,
                           device_batch: int = 256, dropout: str = "faithful", log=None, q_noise=None) -> List[List[np.ndarray]]:
    """The nanobody sampler's accept / re-sweep loop (nanobody_scripts/nanosample.py:316-353), batched.

    Per input sequence the reference keeps ``sample_number`` (rows still wanted) and ``try_num``: while both are
    positive it sweeps all ``batch_size`` replicas once more (from their current, already filled tokens), then
    walks the replicas in order: stop when nothing is wanted; an accepted row is written and counted; a rejected
    row is written only when ``try_num == 1``; ``try_num`` drops by one per row looked at.  Here every sweep runs
    all still-active sequences as one device batch; the decisions are taken on every rank from all-gathered
    tokens.  Returns, per job, the rows written, in order."""
    state = [{"left": want, "tries": tries, "tokens": None, "out": []} for _ in jobs]
    active = [j for j in range(len(jobs)) if want > 0 and tries > 0]
    sweep = 0
    while active:
        sub = [Job(tokens=jobs[j].tokens if state[j]["tokens"] is None else state[j]["tokens"], region=jobs[j].region,
                   loc=jobs[j].loc, chain=jobs[j].chain, name=jobs[j].name) for j in active]
        # noise is keyed by the ORIGINAL job index: a sequence's samples do not depend on which other inputs were
        # accepted earlier (or are in the file at all)
        # (q_noise: injected noise of sweep 0 only, [1, Tmax, len(jobs) * replicas, 22] keyed like the generated noise: parity runs)
        res = sample_jobs(model, sub, replicas, seed + 1000003 * sweep, device_batch=device_batch, dropout=dropout,
                          all_ranks=True, job_ids=active, q_noise=q_noise if sweep == 0 else None)
        still = []
        for a, j in enumerate(active):
            st = state[j]
            st["tokens"] = res[a, 0]
            for row in res[a, 0]:
                if st["left"] == 0:
                    break
                ok = bool(accept(row))
                if ok:
                    st["out"].append(row)
                    st["left"] -= 1
                else:
                    if st["tries"] == 1:
                        st["out"].append(row)
                    if log is not None:
                        log(jobs[j], row)
                st["tries"] -= 1
            if st["left"] > 0 and st["tries"] > 0:
                still.append(j)
        active = still
        sweep += 1
    return [st["out"] for st in state]
