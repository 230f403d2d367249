"""TEST INFRASTRUCTURE ONLY -- "adversarial statistics" golden vectors from the REFERENCE's own classes (VERDICT r2 "Next" #2).

    python oracle/make_golden_adversarial.py           (build container: needs /root/reference; ~1 min of CPU)

Every other reference-generated fixture uses freshly initialised weights: zero-mean rows, no outlier channels, |x| = O(1).
Two design choices of the HIP path are only as safe as that statistic -- LayerNorm folded into column-centred weights
(x W'' on the RAW residual row) and the unscaled fp16 (hi, lo) split of activations (HUDIFF_X3=1).  Here the PRODUCTION
architecture (configs/antibody_train.yml, heavy_train.yml) is loaded with ``hudiff_amd.synthetic.adversarial_state_dict``
(row mean >> row std; massive channels; |x| beyond the fp16 range; |x| << 2^-3) into the reference's AntiTFNet /
NanoAntiTFNet, and its outputs on two real evaluation rows are recorded:

  logits      float32 forward, dropout off  (what "the reference PyTorch CPU path" returns)
  logits_f64  the same module in float64     (how much of a difference is the reference's own round-off)
  stats       row mean / row std and max |x| in front of the first attention and of the two folded LayerNorms
  q, sampled, final   a 6-step sampling trace under the recorded Exp(1) noise

Weights are not stored (seed + recipe + SHA-256).  Only data is written.
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
SEED = 5
TRACE_STEPS = 6


def main():
    torch.set_num_threads(8)
    for kind in ("ab", "nb"):
        cfg = dict(S.AB_CONFIG if kind == "ab" else S.NB_CONFIG, dropout=0.0)
        batch = E.eval_batch("huab348" if kind == "ab" else "vhh", 2, row0=3, mode="finetune" if kind == "ab" else "plain")
        tok, reg = batch["tokens"].astype(np.int64), batch["region"].astype(np.int64)
        chain = None if batch["chain"] is None else batch["chain"].astype(np.int64)
        for variant in S.ADVERSARIAL_VARIANTS:
            sd = S.adversarial_state_dict(kind, cfg, SEED, variant)
            model = deep.build(kind, cfg, sd)
            stats = {}

            def hook(name):
                def f(mod, inp):
                    x = inp[0].detach().double()
                    mu, sdv = x.mean(-1), x.std(-1)
                    prev = stats.get(name, (0.0, 0.0, np.inf))
                    stats[name] = (max(prev[0], float((mu.abs() / sdv).max())), max(prev[1], float(x.abs().max())), min(prev[2], float(sdv.min())))
                return f
            handles = []
            for blk in model.self_at.layers:
                handles += [blk.attn_hl.register_forward_pre_hook(hook("attn1_in")), blk.norm_hl1.register_forward_pre_hook(hook("norm1_in")),
                            blk.norm_hl2.register_forward_pre_hook(hook("norm2_in"))]
            logits = mg.ref_forward(model, tok, reg, chain)
            for h in handles:
                h.remove()
            m64 = deep.build(kind, cfg, sd).double()
            with torch.no_grad():
                l64 = m64(torch.from_numpy(tok), torch.from_numpy(reg), None if chain is None else torch.from_numpy(chain)).numpy()
            own = float(np.abs(logits - l64).max())
            assert own < 5e-5, (kind, variant, own)          # the function itself is well conditioned: 1e-4 is a meaningful bar
            # short trace: the two rows are different antibodies, each with its own visiting order (one at a time, as sample.py)
            finals, qs, sampled, locs = [], [], [], []
            for r in range(2):
                loc = batch["order"][r, :TRACE_STEPS].astype(np.int64)
                ch = None if chain is None else np.array([chain[r], chain[2 + r]], np.int64)
                torch.manual_seed(99 + r)
                with mg.Recorder() as rec:
                    final, steps = mg.ref_sample_loop(model, tok[r:r + 1], reg[r:r + 1], ch, loc, rec)
                finals.append(final[0]); qs.append(np.stack(rec.q)[:, 0]); sampled.append(np.array([s[3][0] for s in steps])); locs.append(loc)
            np.savez_compressed(
                os.path.join(OUT, f"adv_{kind}_{variant}.npz"), weight_seed=np.int64(SEED), weight_sha256=np.array(deep.weights_digest(sd)),
                tokens=tok, region=reg, chain=(np.zeros(0, np.int64) if chain is None else chain), logits=logits,
                logits_f64=l64.astype(np.float32), reference_f32_vs_f64=np.float32(own),
                stat_names=np.array(sorted(stats)), stats=np.array([stats[k] for k in sorted(stats)], np.float64),
                order=np.stack(locs), q=np.stack(qs, 1), sampled=np.stack(sampled, 1), final=np.stack(finals))
            print(kind, variant, "max|logit| %.3g  reference f32 vs f64 %.2e " % (np.abs(l64).max(), own),
                  {k: "mean/std %.1f max|x| %.3g min std %.3g" % v for k, v in sorted(stats.items())}, flush=True)


if __name__ == "__main__":
    main()
