"""TEST INFRASTRUCTURE ONLY -- generates tests/golden/*.npz by running the REFERENCE's own classes.

Run in the build container (where /root/reference exists):

    python oracle/make_golden.py

The reference has no tests or golden vectors for the sampling path, so the vectors
that pin the oracle are produced here by importing ``AntiTFNet`` / ``NanoAntiTFNet`` from
/root/reference (through oracle/ref_import.py) on a *micro* configuration (same architecture,
small widths) with seeded random weights, and recording inputs, outputs and the noise the
reference consumed:

* ``F.dropout`` (functional, training=True -- active at inference, model/encoder/model.py:176-178,
  295-303) is wrapped so that the keep-mask of every call is recorded while the global torch RNG is
  consumed exactly as the unwrapped call would;
* ``torch.multinomial`` is wrapped to record the Exp(1) noise ``q`` (CPU multinomial == argmax(p/q));
  bit-identity with the real ``torch.multinomial`` under the same generator state is asserted.

Only data is written (inputs / weights / expected outputs); no reference source is copied.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_import  # noqa: E402
import hudiff_oracle as ho  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")

MICRO_AB = dict(n_tokens=23, d_embedding=16, d_model=16, n_encoder_layers=2, aa_kernel_size=7, r=128,
                n_side=3, s_embedding=4, s_model=16, n_region=7, r_embedding=4, r_model=16,
                n_pos_model=16, max_len=291, sum_d_model=48, dual_layers=2, att_model=128,
                dim_feedforward=32, nhead=2, cs_layers=2, dropout=0.0, activation="gelu")
MICRO_NB = dict(n_tokens=23, d_embedding=16, d_model=16, n_encoder_layers=2, aa_kernel_size=7, r=128,
                n_region=7, r_embedding=4, r_model=16, n_pos_model=16, max_len=152, sum_d_model=32,
                dual_layers=2, att_model=128, dim_feedforward=32, nhead=2, cs_layers=2, dropout=0.0,
                activation="gelu")


# ------------------------------------------------------------------ noise recorders
class Recorder:
    def __init__(self):
        self.masks = []      # list of (shape, p, uint8 keep-mask) in consumption order
        self.q = []          # list of float32 [B, 22]
        self._real_dropout = F.dropout
        self._real_multinomial = torch.multinomial

    def dropout(self, x, p=0.5, training=True, inplace=False):
        if not training:                      # nn.Dropout modules in eval mode (MLP.dropout)
            return self._real_dropout(x, p, False, inplace)
        assert not inplace
        noise = self._real_dropout(torch.ones_like(x), p, True)        # == mask / (1-p), same RNG use
        keep = (noise != 0)
        self.masks.append((tuple(x.shape), float(p), keep.numpy().astype(np.uint8)))
        return x * noise

    def multinomial(self, probs, num_samples=1, replacement=False, *, generator=None):
        assert num_samples == 1 and generator is None
        state = torch.get_rng_state()
        real = self._real_multinomial(probs, 1)
        torch.set_rng_state(state)
        q = torch.empty_like(probs).exponential_(1)
        mine = torch.argmax(probs / q, dim=-1, keepdim=True)
        assert torch.equal(real, mine), "torch.multinomial is no longer argmax(p/Exp(1))"
        self.q.append(q.numpy().astype(np.float32))
        return mine

    def __enter__(self):
        F.dropout = self.dropout
        torch.multinomial = self.multinomial
        return self

    def __exit__(self, *a):
        F.dropout = self._real_dropout
        torch.multinomial = self._real_multinomial


def canonical_masks(kind, rec_masks, cfg, B):
    """Re-order masks from the reference's RNG consumption order into the
    canonical [n_layers, B, L, width] slot layout (H slots 0..151 then L slots 152..290)."""
    L, d, D = cfg["max_len"], cfg["d_model"], cfg["sum_d_model"]
    ne, nc = cfg["n_encoder_layers"], cfg["dual_layers"]
    enc = np.zeros((ne, B, L, d), np.uint8)
    conv = np.zeros((nc, B, L, D), np.uint8)
    it = iter(rec_masks)
    if kind == "ab":
        for n in range(ne):                       # ByteNetTime._convolve: per layer H then L
            sh, p, m = next(it); assert sh == (B, 152, d) and abs(p - cfg["dropout"]) < 1e-12
            enc[n, :, :152] = m
            sh, p, m = next(it); assert sh == (B, 139, d)
            enc[n, :, 152:] = m
        for n in range(nc):                       # DualConv.forward: whole H stack first ...
            sh, p, m = next(it); assert sh == (B, 152, D) and p == 0.5
            conv[n, :, :152] = m
        for n in range(nc):                       # ... then the whole L stack
            sh, p, m = next(it); assert sh == (B, 139, D) and p == 0.5
            conv[n, :, 152:] = m
    else:
        for n in range(ne):
            sh, p, m = next(it); assert sh == (B, L, d)
            enc[n] = m
        for n in range(nc):
            sh, p, m = next(it); assert sh == (B, L, D) and p == 0.5
            conv[n] = m
    assert next(it, None) is None
    return enc, conv


# ------------------------------------------------------------------ model / input builders
def build_reference(kind, cfg, seed):
    AntiTFNet, NanoAntiTFNet = ref_import.reference_models()
    torch.manual_seed(seed)
    model = (AntiTFNet if kind == "ab" else NanoAntiTFNet)(**cfg)
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():                          # make LN affine / biases non-trivial
        for name, p in model.named_parameters():
            if p.dim() == 1:
                if name.endswith("weight"):
                    p.copy_(1.0 + 0.2 * torch.randn(p.shape, generator=g))
                else:
                    p.copy_(0.2 * torch.randn(p.shape, generator=g))
            elif "embed" in name:
                p.copy_(torch.randn(p.shape, generator=g))
    model.eval()
    return model


def state_to_numpy(model):
    out = {}
    for k, v in model.state_dict().items():
        if v.is_complex():
            continue                               # '...rope' buffers: recomputable (App. B)
        out[k] = v.detach().numpy().astype(np.float32)
    return out


def make_inputs(kind, B, rng, tables, mode):
    """Synthetic pre-slotted sequences -> (tokens, region, chain, loc). No ANARCI here."""
    t = tables
    if kind == "ab":
        L = 291
        region = np.array(t["HEAVY_REGION_INDEX"] + t["LIGHT_REGION_INDEX"], np.int64)
        maskable = np.array(t["HEAVY_CDR_KABAT_NO_VERNIER"] + t["LIGHT_CDR_KABAT_NO_VERNIER"]) == 0 \
            if mode == "finetune" else np.array(t["HEAVY_CDR_INDEX"] + t["LIGHT_CDR_INDEX"]) == 0
        gaps = list(range(111, 135)) + [9] + [152 + i for i in range(111, 123)]
    else:
        L = 152
        region = np.array(t["HEAVY_REGION_INDEX"], np.int64)
        maskable = np.array(t["INPAINT_HEAVY_CDR_INDEX"] if mode == "inpaint" else t["HEAVY_CDR_INDEX"]) == 0
        gaps = list(range(111, 135)) + [9]
    tok = rng.integers(0, 20, size=L)
    keep_some = rng.choice(gaps, size=len(gaps) // 3, replace=False)
    tok[gaps] = 21
    tok[keep_some] = rng.integers(0, 20, size=len(keep_some))
    mask = maskable & (tok != 21) if (kind == "nb" or mode == "finetune") else maskable
    loc = np.arange(L)[mask]
    tok = tok.copy()
    tok[mask] = 22
    tokens = np.repeat(tok[None], B, 0)
    regions = np.repeat(region[None], B, 0)
    chain = np.array([0] * B + [2] * B, np.int64) if kind == "ab" else None
    return tokens, regions, chain, loc


def ref_forward(model, tokens, region, chain):
    with torch.no_grad():
        out = model(torch.from_numpy(tokens), torch.from_numpy(region),
                    None if chain is None else torch.from_numpy(chain))
    return out.numpy().astype(np.float32)


def ref_sample_loop(model, tokens, region, chain, loc, rec):
    """The reference loop, verbatim in behaviour (sample.py:499-513 / nanosample.py:316-329)."""
    tok = torch.from_numpy(tokens.copy())
    reg = torch.from_numpy(region)
    chn = None if chain is None else torch.from_numpy(chain)
    steps = []
    with torch.no_grad():
        for i in loc:
            pred = model(tok, reg, chn)
            lg = pred[:, i, :22]
            soft = torch.nn.functional.softmax(lg, dim=1)
            s = torch.multinomial(soft, num_samples=1)
            tok[:, i] = s.squeeze(-1)
            steps.append((int(i), lg.numpy().astype(np.float32), soft.numpy().astype(np.float32),
                          s.squeeze(-1).numpy().astype(np.int64)))
    return tok.numpy(), steps


def main():
    os.makedirs(OUT, exist_ok=True)
    tables = ref_import.reference_tables()
    # ---- tables fixture (data only) -----------------------------------------------------------
    np.savez_compressed(
        os.path.join(OUT, "tables.npz"),
        heavy_positions=np.array(sorted(tables["HEAVY_POSITIONS_dict"], key=tables["HEAVY_POSITIONS_dict"].get)),
        light_positions=np.array(sorted(tables["LIGHT_POSITIONS_dict"], key=tables["LIGHT_POSITIONS_dict"].get)),
        **{k.lower(): np.array(v, np.int8) for k, v in tables.items() if not k.endswith("_dict")})

    for kind, base_cfg in (("ab", MICRO_AB), ("nb", MICRO_NB)):
        rng = np.random.default_rng(7 if kind == "ab" else 11)
        cfg0 = dict(base_cfg)
        p_drop = 0.2 if kind == "ab" else 0.5          # configs/antibody_train.yml:23, heavy_train.yml:20
        cfg1 = dict(base_cfg, dropout=p_drop)
        model0 = build_reference(kind, cfg0, seed=101 if kind == "ab" else 202)
        AntiTFNet, NanoAntiTFNet = ref_import.reference_models()
        model1 = (AntiTFNet if kind == "ab" else NanoAntiTFNet)(**cfg1)
        model1.load_state_dict(model0.state_dict(), strict=True)
        model1.eval()
        sd = state_to_numpy(model0)
        np.savez_compressed(os.path.join(OUT, f"micro_{kind}_weights.npz"), **sd)

        # ---- forward, dropout off, with intermediates ----------------------------------------
        B = 3
        mode = "finetune" if kind == "ab" else "plain"
        tokens, region, chain, loc = make_inputs(kind, B, rng, tables, mode)
        # make the rows differ: partially fill some masked slots at random
        for b in range(B):
            fill = rng.choice(loc, size=(len(loc) * b) // B, replace=False)
            tokens[b, fill] = rng.integers(0, 22, size=len(fill))
        if kind == "ab":
            chain = np.array([0, 0, 0, 2, 1, 2], np.int64)
        acts = {}
        hooks = []
        names = {"aa_encoder": "aa_encoder", "pos_encoder": "pos", "self_at": "att_out",
                 "last_norm": "last_norm"}
        for mod_name, key in names.items():
            hooks.append(getattr(model0, mod_name).register_forward_hook(
                lambda m, i, o, key=key: acts.__setitem__(key, o.detach().numpy().astype(np.float32))))
        logits = ref_forward(model0, tokens, region, chain)
        for h in hooks:
            h.remove()
        np.savez_compressed(os.path.join(OUT, f"micro_{kind}_forward.npz"), tokens=tokens, region=region,
                            chain=(np.zeros(0, np.int64) if chain is None else chain), logits=logits,
                            **{"act_" + k: v for k, v in acts.items()})

        # ---- forward, dropout ON (cfg.dropout > 0), recorded masks -----------------------------
        torch.manual_seed(1234)
        with Recorder() as rec:
            logits_d = ref_forward(model1, tokens, region, chain)
        enc, conv = canonical_masks(kind, rec.masks, cfg1, B)
        np.savez_compressed(os.path.join(OUT, f"micro_{kind}_forward_dropout.npz"), tokens=tokens,
                            region=region, chain=(np.zeros(0, np.int64) if chain is None else chain),
                            p=np.float32(p_drop), enc_masks=np.packbits(enc), conv_masks=np.packbits(conv),
                            enc_shape=np.array(enc.shape), conv_shape=np.array(conv.shape), logits=logits_d)

        # ---- full sampling trace, dropout off ---------------------------------------------------
        B = 2
        modes = ["finetune"] if kind == "ab" else ["plain", "inpaint"]
        for mode in modes:
            tokens, region, chain, loc = make_inputs(kind, B, rng, tables, mode)
            np.random.seed(2023)
            np.random.shuffle(loc)                                   # sample.py:497-498
            torch.manual_seed(2023)
            with Recorder() as rec:
                final, steps = ref_sample_loop(model0, tokens, region, chain, loc, rec)
            assert not rec.masks
            np.savez_compressed(
                os.path.join(OUT, f"micro_{kind}_sample_{mode}.npz"), tokens=tokens, region=region,
                chain=(np.zeros(0, np.int64) if chain is None else chain), loc=loc.astype(np.int64),
                q=np.stack(rec.q), step_logits=np.stack([s[1] for s in steps]),
                step_probs=np.stack([s[2] for s in steps]), step_sampled=np.stack([s[3] for s in steps]),
                final=final)

        # ---- short sampling trace, dropout ON ---------------------------------------------------
        tokens, region, chain, loc = make_inputs(kind, B, rng, tables, "finetune" if kind == "ab" else "plain")
        np.random.seed(7)
        np.random.shuffle(loc)
        loc = loc[:10]
        torch.manual_seed(99)
        enc_all, conv_all, q_all = [], [], []
        tok_t = tokens.copy()
        step_sampled = []
        for i in loc:                                               # one recorder per step to split masks
            with Recorder() as rec:
                tok_t, steps = ref_sample_loop(model1, tok_t, region, chain, [i], rec)
            e, c = canonical_masks(kind, rec.masks, cfg1, B)
            enc_all.append(e); conv_all.append(c); q_all.append(rec.q[0]); step_sampled.append(steps[0][3])
        enc_all, conv_all = np.stack(enc_all), np.stack(conv_all)
        np.savez_compressed(
            os.path.join(OUT, f"micro_{kind}_sample_dropout.npz"), tokens=tokens, region=region,
            chain=(np.zeros(0, np.int64) if chain is None else chain), loc=np.asarray(loc, np.int64),
            p=np.float32(p_drop), q=np.stack(q_all), enc_masks=np.packbits(enc_all),
            conv_masks=np.packbits(conv_all), enc_shape=np.array(enc_all.shape),
            conv_shape=np.array(conv_all.shape), step_sampled=np.stack(step_sampled), final=tok_t)

        # ---- antibody PRETRAIN mask (sample.py:148-151: every slot outside the IMGT CDRs, gaps included -> T = 185, the
        #      longest schedule), full trace, dropout off.  Added in round 3 with its OWN generator so that every fixture
        #      above stays bit-identical to what rounds 1-2 committed.
        if kind == "ab":
            rng_p = np.random.default_rng(31)
            tokens, region, chain, loc = make_inputs(kind, B, rng_p, tables, "pretrain")
            assert len(loc) == 185
            np.random.seed(2024)
            np.random.shuffle(loc)
            torch.manual_seed(2024)
            with Recorder() as rec:
                final, steps = ref_sample_loop(model0, tokens, region, chain, loc, rec)
            assert not rec.masks
            np.savez_compressed(
                os.path.join(OUT, "micro_ab_sample_pretrain.npz"), tokens=tokens, region=region, chain=chain,
                loc=loc.astype(np.int64), q=np.stack(rec.q), step_logits=np.stack([s[1] for s in steps]),
                step_probs=np.stack([s[2] for s in steps]), step_sampled=np.stack([s[3] for s in steps]), final=final)

        # ---- antibody INPAINT mask (`--sample_method inpaint`, sample.py:283-310, 486-489): the inputs are what the reference's own
        #      batch_inpaint_input_element built for a CDR-grafted HuAb348 pair (tests/golden/input_prep.json, written by
        #      oracle/make_golden_inputs.py from that function: framework positions that differ from the graft template masked, the
        #      grafted CDRs kept); the reference loop then samples them.  Own generator, as above.
        if kind == "ab":
            import json
            with open(os.path.join(OUT, "input_prep.json")) as f:
                case = next(c for c in json.load(f)["cases"] if c["name"] == "huab348_7")
            e = case["expect"]["inpaint_pad0"]
            Bg = 3
            tokens = np.repeat(np.array(e["tokens"], np.int64)[None], Bg, 0)
            region = np.repeat(np.array(e["region"], np.int64)[None], Bg, 0)
            chain = np.array([e["chain"][0]] * Bg + [e["chain"][-1]] * Bg, np.int64)
            loc = np.array(e["loc"], np.int64)
            assert (tokens[0, loc] == 22).all() and (tokens[0] == 22).sum() == len(loc) == 62
            np.random.seed(2025)
            np.random.shuffle(loc)
            torch.manual_seed(2025)
            with Recorder() as rec:
                final, steps = ref_sample_loop(model0, tokens, region, chain, loc, rec)
            assert not rec.masks
            np.savez_compressed(
                os.path.join(OUT, "micro_ab_sample_graft.npz"), tokens=tokens, region=region, chain=chain,
                loc=loc.astype(np.int64), q=np.stack(rec.q), step_logits=np.stack([s[1] for s in steps]),
                step_probs=np.stack([s[2] for s in steps]), step_sampled=np.stack([s[3] for s in steps]), final=final)

        np.savez_compressed(os.path.join(OUT, f"micro_{kind}_config.npz"),
                            **{k: np.array(v) for k, v in cfg0.items()})
        print(kind, "golden written; params:", sum(v.size for v in sd.values()))


if __name__ == "__main__":
    main()
