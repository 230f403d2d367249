"""TEST INFRASTRUCTURE ONLY -- a complete sampling trace of the REFERENCE at PRODUCTION width.

    python oracle/make_golden_prod_trace.py            (build container: needs /root/reference; ~5 min of CPU)

The reference's own ``AntiTFNet`` / ``NanoAntiTFNet`` (configs/antibody_train.yml / heavy_train.yml shapes, dropout 0 so that
the only noise is ``torch.multinomial``'s) run the loop of antibody_scripts/sample.py:499-513 on two real rows (HuAb348 pairs /
VHH sequences from hudiff_amd/data/real_rows.npz) with the recorded Exp(1) noise; inputs, noise, per-step draws and final tokens
are stored.  Weights are NOT stored: ``hudiff_amd.synthetic.random_state_dict(kind, cfg, seed)`` (SHA-256 in the fixture).
The GPU tests replay the noise through the fp32 kernels and -- padded with filler rows to a launch large enough -- through the
split-precision kernels: both must reproduce the reference's tokens bit for bit.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import make_golden as mg  # noqa: E402
import make_golden_deep as deep  # noqa: E402
from hudiff_amd import evalsets as E  # noqa: E402
from hudiff_amd import synthetic as S  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
# (file suffix, kind) -> weight seed, first evaluation row, masking mode, torch seed of row 0.  The "_b" set (round 5) runs other
# weights, other rows and -- for the nanobody model -- the un-masked sampling mode of configs[3] (--inpaint_sample False).
SETS = {("", "ab"): (0, 5, "finetune", 2023), ("", "nb"): (0, 5, "inpaint", 2023),
        ("_b", "ab"): (1, 57, "finetune", 4051), ("_b", "nb"): (1, 61, "plain", 4051)}


def main():
    torch.set_num_threads(8)
    only = sys.argv[1:]                                  # e.g. "_b" to write the second set only
    for (suffix, kind), (wseed, row0, mode, tseed) in SETS.items():
        if only and suffix not in only:
            continue
        cfg = dict(S.AB_CONFIG if kind == "ab" else S.NB_CONFIG, dropout=0.0)
        sd = S.random_state_dict(kind, cfg, seed=wseed)
        model = deep.build(kind, cfg, sd)
        B = 2
        batch = E.eval_batch("huab348" if kind == "ab" else "vhh", B, row0=row0, mode=mode)
        # the reference visits ONE shuffled order for all replicas of an antibody; the two rows here are different antibodies,
        # so each keeps its own order: run them one at a time, exactly as sample.py does per input row
        finals, qs, sampled, locs = [], [], [], []
        for r in range(B):
            loc = batch["order"][r, :batch["T"][r]].astype(np.int64)
            tokens = batch["tokens"][r:r + 1].astype(np.int64)
            region = batch["region"][r:r + 1].astype(np.int64)
            chain = None if batch["chain"] is None else np.array([batch["chain"][r], batch["chain"][B + r]], np.int64)
            torch.manual_seed(tseed + r)
            with mg.Recorder() as rec:
                final, steps = mg.ref_sample_loop(model, tokens, region, chain, loc, rec)
            finals.append(final[0]); qs.append(np.stack(rec.q)[:, 0]); sampled.append(np.array([s[3][0] for s in steps])); locs.append(loc)
            print(kind, suffix, "row", r, "steps", len(loc), flush=True)
        Tmax = max(len(l) for l in locs)
        q = np.ones((Tmax, B, 22), np.float32)
        order = np.zeros((B, Tmax), np.int64)
        for r in range(B):
            q[:len(locs[r]), r] = qs[r]
            order[r, :len(locs[r])] = locs[r]
        np.savez_compressed(
            os.path.join(OUT, f"prod_{kind}_trace{suffix}.npz"), weight_seed=np.int64(wseed), weight_sha256=np.array(deep.weights_digest(sd)),
            tokens=batch["tokens"].astype(np.int64), region=batch["region"].astype(np.int64),
            chain=(np.zeros(0, np.int64) if batch["chain"] is None else batch["chain"].astype(np.int64)),
            order=order, T=np.array([len(l) for l in locs], np.int64), q=q,
            sampled=np.stack([np.pad(s, (0, Tmax - len(s))) for s in sampled]), final=np.stack(finals))
        print(kind, suffix, "production-width trace written")


if __name__ == "__main__":
    main()
