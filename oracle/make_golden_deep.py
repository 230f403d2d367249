"""TEST INFRASTRUCTURE ONLY -- "deep-micro" golden vectors from the REFERENCE's own classes.

    python oracle/make_golden_deep.py              (build container: needs /root/reference)

The micro fixtures of oracle/make_golden.py have 2 conv layers, 2 attention blocks and 2 heads, so only dilations 1
and 2 and a two-head attention are pinned to the reference there.  These fixtures keep the PRODUCTION depth and head
structure (configs/antibody_train.yml:3-24, heavy_train.yml:3-21: 6 + 6 ByteNet blocks = dilations 1, 2, 4, 8, 16, 32,
5 SelfAttBlocks, 8 heads x 64 = att_model 512) at a small width (d_model 16, sum_d_model 48 / 32, feed-forward 32),
so every layer kind of the production stack is compared with the reference's arithmetic, not only with the oracle.

Weights are NOT stored: they are ``hudiff_amd.synthetic.random_state_dict(kind, DEEP_CFG, seed)`` (numpy PCG64, stable
across machines); the fixture carries their SHA-256 so that a test can tell a weight mismatch from a parity failure.
Stored (data only): inputs, the reference's logits with dropout off, its logits with dropout on + the keep-masks it
drew, and a 16-step sampling trace (recorded Exp(1) noise, per-step draws, final tokens).
"""
from __future__ import annotations

import hashlib
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import ref_import  # noqa: E402
import make_golden as mg  # noqa: E402  (Recorder, canonical_masks, make_inputs, ref_forward, ref_sample_loop)
from hudiff_amd import synthetic as S  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")

DEEP_AB = dict(S.AB_CONFIG, d_embedding=16, d_model=16, s_model=16, r_model=16, n_pos_model=16, sum_d_model=48,
               dim_feedforward=32, dropout=0.0)
DEEP_NB = dict(S.NB_CONFIG, d_embedding=16, d_model=16, r_model=16, n_pos_model=16, sum_d_model=32,
               dim_feedforward=32, dropout=0.0)
WEIGHT_SEED = {"ab": 77, "nb": 78}


def weights_digest(sd):
    h = hashlib.sha256()
    for k in sorted(sd):
        h.update(k.encode())
        h.update(np.ascontiguousarray(sd[k], dtype=np.float32).tobytes())
    return h.hexdigest()


def build(kind, cfg, sd):
    AntiTFNet, NanoAntiTFNet = ref_import.reference_models()
    model = (AntiTFNet if kind == "ab" else NanoAntiTFNet)(**cfg)
    own = model.state_dict()
    missing = [k for k in own if k not in sd]
    assert all(k.endswith(".rope") or k.endswith("pos_embedding.pe") for k in missing), missing   # recomputable buffers
    assert not [k for k in sd if k not in own]
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    model.eval()
    return model


def main():
    tables = ref_import.reference_tables()
    for kind, base in (("ab", DEEP_AB), ("nb", DEEP_NB)):
        sd = S.random_state_dict(kind, base, seed=WEIGHT_SEED[kind])
        p = 0.2 if kind == "ab" else 0.5
        model0 = build(kind, base, sd)
        model1 = build(kind, dict(base, dropout=p), sd)
        rng = np.random.default_rng(31 if kind == "ab" else 37)
        B = 2
        mode = "finetune" if kind == "ab" else "inpaint"
        tokens, region, chain, loc = mg.make_inputs(kind, B, rng, tables, mode)
        fill = rng.choice(loc, size=len(loc) // 2, replace=False)            # row 1: half-way through a sample
        tokens[1, fill] = rng.integers(0, 22, size=len(fill))
        if kind == "ab":
            chain = np.array([0, 0, 2, 1], np.int64)
        # per-component activations of the same forward (round 6: tests/test_gpu_components.py holds the HIP buffers to them row by row
        # of SURVEY.md section 8a): forward hooks on the reference's own sub-modules, written to a SEPARATE file so that deep_{kind}.npz
        # regenerates bit for bit as before
        acts, hooks = {}, []

        def grab(key):
            def hook(mod, inp, out):
                o = out if not isinstance(out, (tuple, list)) else torch.cat(list(out), dim=1)       # DualConv returns (h, l)
                acts[key] = o.detach().numpy().astype(np.float32)
            return hook
        conv_name = "dual_conv_block" if kind == "ab" else "nano_conv_block"
        for mod_name, key in (("aa_encoder", "aa_encoder"), ("region_encoder", "region"), ("pos_encoder", "pos"), (conv_name, "conv"),
                              ("last_norm", "last_norm")) + ((("side_encoder", "side"),) if kind == "ab" else ()):
            hooks.append(getattr(model0, mod_name).register_forward_hook(grab(key)))
        for n, layer in enumerate(model0.self_at.layers):
            hooks.append(layer.register_forward_hook(grab(f"att{n}")))
            hooks.append(layer.attn_hl.register_forward_hook(grab(f"att{n}_a1")))          # AttLayer output (before the residual)
        logits = mg.ref_forward(model0, tokens, region, chain)
        for h in hooks:
            h.remove()
        np.savez_compressed(os.path.join(OUT, f"deep_{kind}_acts.npz"), weight_sha256=np.array(weights_digest(sd)),
                            **{"act_" + k: v for k, v in acts.items()})
        print(kind, "component activations:", {k: v.shape for k, v in acts.items()})
        torch.manual_seed(4321)
        with mg.Recorder() as rec:
            logits_d = mg.ref_forward(model1, tokens, region, chain)
        enc, conv = mg.canonical_masks(kind, rec.masks, dict(base, dropout=p), B)
        # short trace, dropout off, shared shuffled order as the reference does it (sample.py:497-513)
        s_tokens, s_region, s_chain, s_loc = mg.make_inputs(kind, B, rng, tables, mode)
        np.random.seed(11)
        np.random.shuffle(s_loc)
        s_loc = s_loc[:16]
        torch.manual_seed(2023)
        with mg.Recorder() as rec2:
            final, steps = mg.ref_sample_loop(model0, s_tokens, s_region, s_chain, s_loc, rec2)
        np.savez_compressed(
            os.path.join(OUT, f"deep_{kind}.npz"),
            config_keys=np.array(sorted(base)), config_vals=np.array([str(base[k]) for k in sorted(base)]),
            weight_seed=np.int64(WEIGHT_SEED[kind]), weight_sha256=np.array(weights_digest(sd)),
            tokens=tokens, region=region, chain=(np.zeros(0, np.int64) if chain is None else chain), logits=logits,
            p=np.float32(p), enc_masks=np.packbits(enc), conv_masks=np.packbits(conv), enc_shape=np.array(enc.shape),
            conv_shape=np.array(conv.shape), logits_dropout=logits_d,
            s_tokens=s_tokens, s_region=s_region, s_chain=(np.zeros(0, np.int64) if s_chain is None else s_chain),
            s_loc=np.asarray(s_loc, np.int64), q=np.stack(rec2.q), step_logits=np.stack([s[1] for s in steps]),
            step_sampled=np.stack([s[3] for s in steps]), final=final)
        print(kind, "deep golden written; params:", sum(v.size for v in sd.values()),
              "max|logit|", float(np.abs(logits).max()), "dropout shift", float(np.abs(logits_d - logits).max()))


if __name__ == "__main__":
    main()
