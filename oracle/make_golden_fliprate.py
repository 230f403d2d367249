"""TEST INFRASTRUCTURE ONLY -- final tokens of the REFERENCE at production width over many rows (flip-rate evidence).

    python oracle/make_golden_fliprate.py [ab|nb] [rows]      (build container: needs /root/reference; ~20 min of CPU on 8 cores)

north_star asks for "identical top-1 humanized residues" against the reference's fp32 CPU path.  A draw can flip when two ratios
p / q of a step lie within the logit error of each other, so the honest statement is a COUNT over many rows, with the margin of
every draw that differs.  This script runs the reference's own ``AntiTFNet`` / ``NanoAntiTFNet`` (yml shapes, dropout 0: the only
noise is ``torch.multinomial``'s) through the loop of antibody_scripts/sample.py:499-513, one evaluation row at a time exactly as
the reference does, on 64 HuAb348 pairs / 64 VHH sequences (hudiff_amd/data/real_rows.npz), and stores per row: its index in the
evaluation set, the visiting order, the recorded Exp(1) noise, the final tokens, every per-step draw and the reference's own
near-tie margin of every draw (log of best ratio - log of second best).  No weights (``synthetic.random_state_dict``; SHA-256
stored), no reference source.  tests/test_fliprate.py replays the noise through the oracle and the three HIP routes.
"""
from __future__ import annotations

import multiprocessing as mp
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

OUT = os.path.join(ROOT, "tests", "golden")
# kind -> (weight seed, first evaluation row, [(masking mode, rows)], torch seed of the first row)
SETS = {"ab": (0, 100, [("finetune", 64)], 7001), "nb": (0, 100, [("inpaint", 32), ("plain", 32)], 7101)}


def _rows(kind, n_total):
    wseed, row0, parts, tseed = SETS[kind]
    out, r = [], row0
    for mode, n in parts:
        n = min(n, max(0, n_total - len(out)))
        out += [(r + i, mode) for i in range(n)]
        r += n
    return out


def _work(args):
    kind, jobs, threads = args
    import torch
    import make_golden as mg
    import make_golden_deep as deep
    from hudiff_amd import evalsets as E
    from hudiff_amd import synthetic as S
    torch.set_num_threads(threads)
    wseed, _, _, tseed = SETS[kind]
    cfg = dict(S.AB_CONFIG if kind == "ab" else S.NB_CONFIG, dropout=0.0)
    sd = S.random_state_dict(kind, cfg, seed=wseed)
    model = deep.build(kind, cfg, sd)
    res = []
    for j, (row, mode) in jobs:
        batch = E.eval_batch("huab348" if kind == "ab" else "vhh", 1, row0=row, mode=mode)
        loc = batch["order"][0, :batch["T"][0]].astype(np.int64)
        tokens = batch["tokens"][0:1].astype(np.int64)
        region = batch["region"][0:1].astype(np.int64)
        chain = None if batch["chain"] is None else np.array([batch["chain"][0], batch["chain"][1]], np.int64)
        torch.manual_seed(tseed + j)
        with mg.Recorder() as rec:
            final, steps = mg.ref_sample_loop(model, tokens, region, chain, loc, rec)
        q = np.stack(rec.q)[:, 0]                                     # [T, 22]
        margin = np.zeros(len(loc), np.float32)
        for t, (_, _, soft, s) in enumerate(steps):
            ratio = np.log(np.maximum(soft[0].astype(np.float64), 1e-300)) - np.log(q[t].astype(np.float64))
            top = np.sort(ratio)[::-1]
            margin[t] = np.float32(top[0] - top[1])
        res.append(dict(j=j, row=row, mode=mode, tokens=tokens[0], region=region[0], chain=chain, loc=loc, q=q,
                        sampled=np.array([s[3][0] for s in steps]), final=final[0], margin=margin))
        print(kind, "row", row, mode, "steps", len(loc), "min margin %.2e" % margin.min(), flush=True)
    return res, deep.weights_digest(sd)


def main():
    kinds = [a for a in sys.argv[1:] if a in SETS] or list(SETS)
    nums = [int(a) for a in sys.argv[1:] if a.isdigit()]
    n_total = nums[0] if nums else 64
    nproc = int(os.environ.get("FLIP_PROCS", "4"))
    threads = max(1, (os.cpu_count() or 8) // nproc)
    for kind in kinds:
        jobs = list(enumerate(_rows(kind, n_total)))
        chunks = [(kind, jobs[i::nproc], threads) for i in range(nproc)]
        with mp.get_context("spawn").Pool(nproc) as pool:
            parts = pool.map(_work, chunks)
        digest = parts[0][1]
        rows = sorted((r for p, _ in parts for r in p), key=lambda r: r["j"])
        n = len(rows)
        Tmax = max(len(r["loc"]) for r in rows)
        L = rows[0]["tokens"].shape[0]
        q = np.ones((Tmax, n, 22), np.float32)
        order = np.zeros((n, Tmax), np.int64)
        sampled = np.zeros((n, Tmax), np.int64)
        margin = np.full((n, Tmax), np.inf, np.float32)
        for i, r in enumerate(rows):
            T = len(r["loc"])
            q[:T, i] = r["q"]; order[i, :T] = r["loc"]; sampled[i, :T] = r["sampled"]; margin[i, :T] = r["margin"]
        chain = np.zeros(0, np.int64) if rows[0]["chain"] is None else np.concatenate(
            [np.array([r["chain"][0] for r in rows]), np.array([r["chain"][1] for r in rows])]).astype(np.int64)
        np.savez_compressed(
            os.path.join(OUT, f"fliprate_{kind}.npz"), weight_seed=np.int64(SETS[kind][0]), weight_sha256=np.array(digest),
            eval_row=np.array([r["row"] for r in rows], np.int64), mode=np.array([r["mode"] for r in rows]),
            tokens=np.stack([r["tokens"] for r in rows]).astype(np.int8), region=np.stack([r["region"] for r in rows]).astype(np.int8),
            chain=chain.astype(np.int8), order=order.astype(np.int16), T=np.array([len(r["loc"]) for r in rows], np.int16), q=q,
            sampled=sampled.astype(np.int8), final=np.stack([r["final"] for r in rows]).astype(np.int8), margin=margin)
        print(kind, n, "rows written; L", L, "Tmax", Tmax, flush=True)


if __name__ == "__main__":
    main()
