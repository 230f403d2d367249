"""Synthetic weights and pre-slotted inputs for tests and bench.py (no checkpoints / ANARCI offline).

* ``AB_CONFIG`` / ``NB_CONFIG``: the `model:` sections of configs/antibody_train.yml:3-24 and
  configs/heavy_train.yml:3-21 (the production architectures BASELINE.json names).
* ``random_state_dict``: seeded random weights with the reference's state_dict keys and shapes
   -- what ``load_state_dict`` of a real checkpoint would deliver.
* ``synthetic_batch``: rows shaped like numbered HuAb348 / VHH inputs (SURVEY.md §8d "Synthetic
  inputs"): residues i.i.d. over the 20 amino acids, a fixed gap pattern around the IMGT 111/112
  insertion block, the mode's framework mask set to <msk>, a per-row shuffled visiting order.
"""
from __future__ import annotations

import numpy as np

from . import tables

AB_CONFIG = dict(n_tokens=23, d_embedding=256, d_model=256, n_encoder_layers=6, aa_kernel_size=7, r=128,
                 n_side=3, s_embedding=4, s_model=256, n_region=7, r_embedding=4, r_model=256, n_pos_model=256,
                 max_len=291, sum_d_model=768, dual_layers=6, att_model=512, dim_feedforward=256, nhead=8,
                 cs_layers=5, dropout=0.2, activation="gelu")
NB_CONFIG = dict(n_tokens=23, d_embedding=256, d_model=256, n_encoder_layers=6, aa_kernel_size=7, r=128,
                 n_region=7, r_embedding=4, r_model=256, n_pos_model=256, max_len=152, sum_d_model=512,
                 dual_layers=6, att_model=512, dim_feedforward=256, nhead=8, cs_layers=5, dropout=0.5,
                 activation="gelu")


def _bytenet_shapes(prefix, d, dh, k):
    return {
        prefix + "sequence1.0.weight": (d,), prefix + "sequence1.0.bias": (d,),
        prefix + "sequence1.2.conv.weight": (dh, d, 1), prefix + "sequence1.2.conv.bias": (dh,),
        prefix + "sequence1.3.weight": (dh,), prefix + "sequence1.3.bias": (dh,),
        prefix + "conv.weight": (dh, dh, k), prefix + "conv.bias": (dh,),
        prefix + "sequence2.0.weight": (dh,), prefix + "sequence2.0.bias": (dh,),
        prefix + "sequence2.2.conv.weight": (d, dh, 1), prefix + "sequence2.2.conv.bias": (d,),
    }


def state_dict_shapes(kind: str, cfg: dict) -> dict:
    d, D, A, Fd, k = cfg["d_model"], cfg["sum_d_model"], cfg["att_model"], cfg["dim_feedforward"], cfg["aa_kernel_size"]
    re = cfg["r_embedding"]
    sh = {"aa_encoder.embedder.weight": (cfg["n_tokens"], d)}
    stacks = ("h_layers", "l_layers") if kind == "ab" else ("layers",)
    conv = "dual_conv_block" if kind == "ab" else "nano_conv_block"
    for s in stacks:
        for n in range(cfg["n_encoder_layers"]):
            sh.update(_bytenet_shapes(f"aa_encoder.{s}.{n}.", d, d // 2, k))
        for n in range(cfg["dual_layers"]):
            sh.update(_bytenet_shapes(f"{conv}.{s}.{n}.", D, D // 2, k))
    if kind == "ab":
        se = cfg["s_embedding"]
        sh.update({"side_encoder.side_embeddinng.weight": (cfg["n_side"], se),
                   "side_encoder.side_mlp.0.weight": (d, se), "side_encoder.side_mlp.0.bias": (d,),
                   "side_encoder.side_mlp.1.weight": (d,), "side_encoder.side_mlp.1.bias": (d,),
                   "side_encoder.side_mlp.3.weight": (d, d), "side_encoder.side_mlp.3.bias": (d,)})
    sh.update({"region_encoder.region_embedding.weight": (cfg["n_region"], re),
               "region_encoder.region_layer1.0.weight": (re,), "region_encoder.region_layer1.0.bias": (re,),
               "region_encoder.region_layer1.2.conv.weight": (d, re, 1), "region_encoder.region_layer1.2.conv.bias": (d,),
               "region_encoder.region_layer1.3.weight": (d,), "region_encoder.region_layer1.3.bias": (d,),
               "pos_encoder.pos_lin.ln1.weight": (2 * d, d), "pos_encoder.pos_lin.ln1.bias": (2 * d,),
               "pos_encoder.pos_lin.ln2.weight": (d, 2 * d), "pos_encoder.pos_lin.ln2.bias": (d,)})
    for n in range(cfg["cs_layers"]):
        p = f"self_at.layers.{n}."
        for a in ("attn_hl.", "attn_hl_c."):
            for nm in ("query", "key", "value"):
                sh[p + a + nm + ".weight"] = (A, D)
                sh[p + a + nm + ".bias"] = (A,)
            sh[p + a + "out_put.weight"] = (D, A)
            sh[p + a + "out_put.bias"] = (D,)
        for nm in ("norm_hl1", "norm_hl2"):
            sh[p + nm + ".weight"] = (D,)
            sh[p + nm + ".bias"] = (D,)
        sh[p + "ff_hl.0.weight"] = (Fd, D); sh[p + "ff_hl.0.bias"] = (Fd,)
        sh[p + "ff_hl.2.weight"] = (D, Fd); sh[p + "ff_hl.2.bias"] = (D,)
    sh.update({"last_norm.weight": (D,), "last_norm.bias": (D,),
               "decoder.weight": (cfg["n_tokens"], D), "decoder.bias": (cfg["n_tokens"],)})
    return sh


def random_state_dict(kind: str, cfg: dict, seed: int = 0) -> dict:
    """Seeded float32 weights: matrices ~ N(0, 1/fan_in), LayerNorm gamma ~ 1 + 0.1 N, biases / beta ~ 0.1 N,
    embeddings ~ N(0, 1); the last projection of every residual branch (ByteNet sequence2.2, attention
    out_put, ff_hl.2) is scaled by 0.1 so that the residual stream stays O(1) through all 100+ layers even
    with the x2 inference-time dropout scaling -- as it does in a trained network."""
    rng = np.random.default_rng(seed)
    out = {}
    for key, shape in state_dict_shapes(kind, cfg).items():
        if len(shape) == 1:
            is_gamma = key.endswith("weight")
            v = (1.0 if is_gamma else 0.0) + 0.1 * rng.standard_normal(shape)
        elif "embed" in key:
            v = rng.standard_normal(shape)
        else:
            fan_in = int(np.prod(shape[1:]))
            v = rng.standard_normal(shape) / np.sqrt(fan_in)
            if key.endswith(("sequence2.2.conv.weight", "out_put.weight", "ff_hl.2.weight")):
                v *= 0.1
        out[key] = v.astype(np.float32)
    return out


_H_GAP_BLOCK = list(range(111, 135))      # IMGT 111A..112A insertion slots of the heavy / VHH layout
_L_GAP_BLOCK = list(range(111, 123))      # light chain 111A..112A


def _one_chain(rng, n_slots, gap_block, cdr3_len):
    tok = rng.integers(0, 20, size=n_slots)
    gaps = list(gap_block)
    keep = max(0, min(len(gaps), cdr3_len))
    # keep `keep` insertion slots occupied, centred like IMGT fills them (from both ends inwards)
    lo = keep // 2 + keep % 2
    occupied = set(gaps[:lo] + gaps[len(gaps) - (keep - lo):]) if keep else set()
    for g in gaps:
        if g not in occupied:
            tok[g] = 21
    tok[9] = 21                               # IMGT position 10 is absent in most V domains
    return tok


def synthetic_batch(kind: str, B: int, seed: int = 2023, mode: str | None = None, row0: int = 0):
    """-> dict(tokens[B,L] int32 (masked), region[B,L], chain[2B] | None, order[B,Tmax], T[B], truth[B,L]).

    kind 'ab': mode 'finetune' (Kabat-no-vernier framework minus gaps, sample.py:152-170) or 'pretrain'
    (IMGT framework incl. gap slots, :148-151).  kind 'nb': mode 'plain' | 'inpaint' (nanosample.py:129-147).
    Row r is generated from seed + row0 + r, so shards reproduce the same global rows.
    """
    if kind == "ab":
        mode = mode or "finetune"
        L = tables.AB_LEN
        region = tables.ab_region()
        table = (tables.HEAVY_CDR_KABAT_NO_VERNIER + tables.LIGHT_CDR_KABAT_NO_VERNIER) if mode == "finetune" \
            else (tables.HEAVY_CDR_INDEX + tables.LIGHT_CDR_INDEX)
    else:
        mode = mode or "plain"
        L = tables.H_LEN
        region = tables.nb_region()
        table = tables.INPAINT_HEAVY_CDR_INDEX if mode == "inpaint" else tables.HEAVY_CDR_INDEX
    maskable = np.array(table) == 0
    tokens = np.zeros((B, L), np.int32)
    truth = np.zeros((B, L), np.int32)
    orders, Ts = [], []
    chain = np.zeros(2 * B, np.int32) if kind == "ab" else None
    for r in range(B):
        rng = np.random.default_rng(seed + row0 + r)
        h = _one_chain(rng, tables.H_LEN, _H_GAP_BLOCK, int(rng.integers(2, 11)))
        if kind == "ab":
            l = _one_chain(rng, tables.L_LEN, _L_GAP_BLOCK, int(rng.integers(0, 4)))
            tok = np.concatenate([h, l])
            chain[B + r] = 1 if rng.random() < 0.05 else 2      # 5 % lambda, else kappa
        else:
            tok = h
        truth[r] = tok
        m = maskable & (tok != 21) if not (kind == "ab" and mode == "pretrain") else maskable
        loc = np.arange(L)[m]
        rng.shuffle(loc)
        tok = tok.copy()
        tok[m] = 22
        tokens[r] = tok
        orders.append(loc)
        Ts.append(len(loc))
    Tmax = max(Ts) if Ts else 0
    order = np.zeros((B, max(Tmax, 1)), np.int32)
    for r, loc in enumerate(orders):
        order[r, :len(loc)] = loc
    return dict(tokens=tokens, region=np.repeat(region[None].astype(np.int32), B, 0), chain=chain,
                order=order, T=np.array(Ts, np.int32), truth=truth, mode=mode)


ADVERSARIAL_VARIANTS = ("dc", "massive", "huge", "tiny", "plain")


def adversarial_state_dict(kind: str, cfg: dict, seed: int, variant: str) -> dict:
    """``random_state_dict`` bent so that the residual stream takes the statistics trained checkpoints are known for and
    freshly initialised weights never show (VERDICT r2 "Next" #2) -- the function stays well conditioned (the reference's own
    float32 result is within ~4e-5 of its float64 evaluation), only the numbers the kernels see get ugly:

    ``dc``       a constant added to every residual-branch output bias of the conv and attention stages: row mean / row std of
                 the stream reaches ~80 in front of the un-normalised first attention and ~50 in front of the LayerNorms
                 whose affine part is folded into the next projection (value projections damped so that the offset does
                 not turn into channel variance)
    ``massive``  four channels carry |x| ~ 2e4 (out_put biases + 1e4) and two LayerNorm gains are x 30
    ``huge``     the whole stream scaled by 2^17 (|x| ~ 1e6 > 65504, the fp16 range): embeddings, static branch and every
                 residual-branch output x 2^17, the un-normalised attention inputs / 2^17
    ``tiny``     the same with 2^-12 (|x| ~ 2e-3, row std ~ 3e-4 << 2^-3; below the LayerNorm epsilon)
    ``plain``    nothing bent: the freshly initialised weights themselves (round 5: the control vector -- reference float32 / float64 logits
                 and a short trace at production width with ordinary statistics)

    Used by oracle/make_golden_adversarial.py (loaded into the reference's classes) and by the parity tests."""
    if variant not in ADVERSARIAL_VARIANTS:
        raise ValueError(variant)
    sd = random_state_dict(kind, cfg, seed)
    if variant == "plain":
        return sd
    rng = np.random.default_rng(seed + 1000)
    last_w = ("sequence2.2.conv.weight", "out_put.weight", "ff_hl.2.weight")
    last_b = ("sequence2.2.conv.bias", "out_put.bias", "ff_hl.2.bias")
    if variant == "dc":
        for k in sd:
            if k.endswith(last_b) and ("self_at" in k or "conv_block" in k):
                sd[k] = (sd[k] + 10.0).astype(np.float32)
            if k.endswith("value.weight"):
                sd[k] = (sd[k] * 0.1).astype(np.float32)
    elif variant == "massive":
        ch = rng.choice(cfg["sum_d_model"], size=4, replace=False)
        for k in sd:
            if "self_at" in k and k.endswith("out_put.bias"):
                sd[k][ch] += 1.0e4
            if k.endswith(("norm_hl1.weight", "norm_hl2.weight")):
                sd[k][ch[:2]] *= 30.0
    else:
        s = 2.0 ** (17 if variant == "huge" else -12)
        static = ("region_encoder.region_layer1.3.weight", "region_encoder.region_layer1.3.bias", "side_encoder.side_mlp.3.weight",
                  "side_encoder.side_mlp.3.bias", "pos_encoder.pos_lin.ln1.bias", "pos_encoder.pos_lin.ln2.bias")
        for k in sd:
            if k.endswith(last_w) or k.endswith(last_b) or k == "aa_encoder.embedder.weight" or k in static:
                sd[k] = (sd[k] * s).astype(np.float32)
            if ".attn_hl." in k and k.endswith(("query.weight", "key.weight", "value.weight")):
                sd[k] = (sd[k] / s).astype(np.float32)
        # the sinusoid table is a registered buffer of the checkpoint (model/encoder/model.py:70-78): scaled with the rest
        from .model import _sinusoid_pe
        sd["pos_encoder.pos_embedding.pe"] = (_sinusoid_pe(cfg["max_len"], cfg["d_model"])[:, None, :] * s).astype(np.float32)
    return sd
