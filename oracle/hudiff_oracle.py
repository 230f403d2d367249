"""TEST INFRASTRUCTURE ONLY -- CPU (numpy, float32) restatement of HuDiff's sampling hot path.

This file is the *checker*: only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it.  The product path
(``hudiff_amd``) never does and fails loudly if its HIP library is missing.

What is restated (reference file:line, relative to /root/reference):

* ``AntiTFNet.forward``      model/encoder/model.py:366-384  (``_encoder`` :351-359, ``_att`` :361-364)
* ``NanoAntiTFNet.forward``  model/nanoencoder/model.py:325-343
* ``ByteNetTime``/``NanoByteNetTime``  model/encoder/model.py:155-180, model/nanoencoder/model.py:150-170
* ``SideEmbedder`` :197-205, ``RegionEmbedder`` :222-230, ``PositionalEncoding`` :62-87,
  ``MLP`` :19-33, ``PosEmbedder`` :242-246, ``DualConv`` :277-304, ``NanoConv`` nanoencoder/model.py:255-270
* ``precompute_freqs_cis`` / ``apply_rotary_emb`` / ``AttLayer`` / ``SelfAttBlock`` / ``SelfAttNet``
  model/encoder/cross_attention.py:35-56, 59-88, 149-173, 273-287, 291-310
* the sampling loop  antibody_scripts/sample.py:499-513, nanobody_scripts/nanosample.py:316-329
  (``torch.multinomial(p,1)`` on CPU == ``argmax(p / q)``, ``q ~ Exp(1)``)
* ``ByteNetBlock`` / ``PositionFeedForward`` / ``MaskedConv1d``: third-party ``sequence-models``
  (PyPI, un-pinned in environment.yaml:23, NOT vendored in the reference).  Restated from the
  published upstream semantics; **parity at that boundary is unpinned** (no reference test covers it).

Pinning status: the reference ships NO golden vectors or tests for this path.
The oracle is pinned instead against outputs of the reference's own classes imported in the
build container (``oracle/make_golden.py`` -> ``tests/golden/*.npz``; checked by
``tests/test_oracle_golden.py``), including injected dropout masks and Exp(1) noise.

Reference quirks reproduced on purpose: functional dropout is active at inference
whenever cfg.dropout > 0 (p = cfg.dropout in the token encoder, p = 0.5 in Dual/NanoConv);
SelfAttBlock's last residual comes from the block INPUT; DualConv uses ReLU while NanoConv and the
token encoder use the configured / default GELU; no attention or padding mask anywhere.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import numpy as np
from scipy.special import erf as _erf

F32 = np.float32
N_TOKENS = 23
N_SAMPLE = 22          # sample.py:510  logits[:, i, :len(toks)-1]
TOK_GAP = 21
TOK_MSK = 22
AB_H_LEN = 152
AB_L_LEN = 139

# --------------------------------------------------------------------------------------
# Counter-based noise contract shared with the HIP library (include/hudiff_hip.h "Noise").
# Philox4x32-10, key = (seed_lo, seed_hi).
#   dropout : (k0, k1) = Philox(counter = (0, 0, step, site))[0:2] ; rk = mix32(k0 ^ mix32(global_row + k1))
#             w = mix32(rk + (slot * width + feature) * 0x9E3779B9) ; keep  <=>  w >= floor(p * 2^32)
#             site = layer index for the token encoder, 64 + layer index for Dual/NanoConv
#   sampling: counter = (j >> 2, global_row, step, 0xFFFFFFFF)     word = j & 3 , j in [0, 22)
#             u = ((u32 >> 8) + 0.5) * 2^-24 ;  q = -log(u)  (float32)
# --------------------------------------------------------------------------------------
PHILOX_M0 = np.uint64(0xD2511F53)
PHILOX_M1 = np.uint64(0xCD9E8D57)
PHILOX_W0 = 0x9E3779B9
PHILOX_W1 = 0xBB67AE85
SITE_CONV_BASE = 64
SITE_SAMPLE = 0xFFFFFFFF


def philox4x32(c0, c1, c2, c3, seed: int):
    """Vectorised Philox4x32-10. c* broadcastable uint32 arrays -> 4 uint32 arrays."""
    c0, c1, c2, c3 = np.broadcast_arrays(*(np.asarray(c, dtype=np.uint64) for c in (c0, c1, c2, c3)))
    c0, c1, c2, c3 = c0.copy(), c1.copy(), c2.copy(), c3.copy()
    k0 = seed & 0xFFFFFFFF
    k1 = (seed >> 32) & 0xFFFFFFFF
    mask = np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0 = PHILOX_M0 * c0
        p1 = PHILOX_M1 * c2
        hi0, lo0 = p0 >> np.uint64(32), p0 & mask
        hi1, lo1 = p1 >> np.uint64(32), p1 & mask
        n0 = hi1 ^ c1 ^ np.uint64(k0)
        n2 = hi0 ^ c3 ^ np.uint64(k1)
        c0, c1, c2, c3 = n0, lo1, n2, lo0
        k0 = (k0 + PHILOX_W0) & 0xFFFFFFFF
        k1 = (k1 + PHILOX_W1) & 0xFFFFFFFF
    return tuple(c.astype(np.uint32) for c in (c0, c1, c2, c3))


def mix32(x):
    """lowbias32 finaliser on uint32 arrays (wrap-around arithmetic)."""
    x = np.asarray(x, dtype=np.uint64) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x7FEB352D)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(15)
    x = (x * np.uint64(0x846CA68B)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(16)
    return x


def philox_keep_mask(seed: int, rows: np.ndarray, step: int, site: int, n_slots: int, width: int,
                     p: float) -> np.ndarray:
    """uint8 keep-mask [len(rows), n_slots, width] for one dropout site (see the contract above)."""
    k0, k1, _, _ = philox4x32(np.uint32(0), np.uint32(0), np.uint32(step), np.uint32(site), seed)
    k0, k1 = np.uint64(int(k0)), np.uint64(int(k1))
    m32 = np.uint64(0xFFFFFFFF)
    rows = np.asarray(rows, dtype=np.uint64) & m32
    rk = mix32(k0 ^ mix32((rows + k1) & m32))                            # [R]
    elem = np.arange(n_slots * width, dtype=np.uint64)
    w = mix32((rk[:, None] + elem[None, :] * np.uint64(0x9E3779B9)) & m32)
    thresh = np.uint64(min(int(math.floor(p * 4294967296.0)), 0xFFFFFFFF))
    return (w >= thresh).astype(np.uint8).reshape(len(rows), n_slots, width)


def philox_exp_noise(seed: int, rows: np.ndarray, step: int) -> np.ndarray:
    """float32 Exp(1) noise [len(rows), 22] for one sampling step."""
    j = np.arange(N_SAMPLE, dtype=np.uint32)
    out = philox4x32((j >> 2)[None, :], np.asarray(rows, dtype=np.uint32)[:, None], np.uint32(step),
                     np.uint32(SITE_SAMPLE), seed)
    words = np.stack(out, axis=-1)
    u32 = np.take_along_axis(words, (j & 3).astype(np.int64)[None, :, None], axis=-1)[..., 0]
    u = ((u32 >> np.uint32(8)).astype(F32) + F32(0.5)) * F32(2.0 ** -24)
    return (-np.log(u, dtype=F32)).astype(F32)


# --------------------------------------------------------------------------------------
# Elementary ops (float32 throughout)
# --------------------------------------------------------------------------------------
def layer_norm(x, w, b, eps=1e-5):
    mu = x.mean(axis=-1, keepdims=True, dtype=x.dtype)
    xc = x - mu
    var = (xc * xc).mean(axis=-1, keepdims=True, dtype=x.dtype)
    return (xc / np.sqrt(var + x.dtype.type(eps))) * w + b


def gelu(x):
    t = x.dtype.type
    return (t(0.5) * x * (t(1.0) + _erf(x * t(1.0 / math.sqrt(2.0))))).astype(x.dtype)


def relu(x):
    return np.maximum(x, x.dtype.type(0.0))


ACT = {"gelu": gelu, "relu": relu}


def linear(x, w, b):
    """torch nn.Linear: w [out, in]."""
    return x @ w.T + b


def dilated_conv(x, w, b, dilation):
    """MaskedConv1d with input_mask=None: x [B, Lc, C], w [Cout, Cin, K], zero padding
    dilation*(K-1)//2 on both ends (sequence_models.convolutional.MaskedConv1d)."""
    B, Lc, _ = x.shape
    K = w.shape[2]
    half = (K - 1) // 2
    wt = _tap_major(w)                                  # [K, Cin, Cout] contiguous (BLAS-friendly)
    out = np.zeros((B, Lc, w.shape[0]), dtype=x.dtype)
    for tap in range(K):
        s = (tap - half) * dilation
        lo, hi = max(0, -s), min(Lc, Lc - s)
        if hi <= lo:
            continue
        out[:, lo:hi, :] += np.ascontiguousarray(x[:, lo + s:hi + s, :]) @ wt[tap]
    return out + b


_TAP_CACHE: dict = {}


def _tap_major(w):
    key = id(w)
    hit = _TAP_CACHE.get(key)
    if hit is None or hit[0] is not w:
        hit = (w, np.ascontiguousarray(w.transpose(2, 1, 0)))
        _TAP_CACHE[key] = hit
    return hit[1]


def sinusoid_pe(max_len, d_model):
    """PositionalEncoding buffer 'pe' (model/encoder/model.py:70-78), float32 like torch."""
    position = np.arange(max_len, dtype=F32)[:, None]
    div = np.exp(np.arange(0, d_model, 2).astype(F32) * F32(-math.log(10000.0) / d_model)).astype(F32)
    pe = np.zeros((max_len, d_model), dtype=F32)
    ang = (position * div).astype(F32)
    pe[:, 0::2] = np.sin(ang)
    pe[:, 1::2] = np.cos(ang)
    return pe


def rope_table(head_dim, length, theta=10000.0):
    """precompute_freqs_cis (cross_attention.py:35-56) -> (cos, sin) float32 [length, head_dim/2]."""
    freqs = (F32(1.0) / (F32(theta) ** (np.arange(0, head_dim, 2)[: head_dim // 2].astype(F32) / F32(head_dim)))).astype(F32)
    ang = np.outer(np.arange(length, dtype=F32), freqs).astype(F32)
    return np.cos(ang).astype(F32), np.sin(ang).astype(F32)


def apply_rope(x, cos, sin):
    """apply_rotary_emb (cross_attention.py:59-88): x [B, L, H, hd]; complex pairs (2k, 2k+1)."""
    xr, xi = x[..., 0::2], x[..., 1::2]
    c, s = cos[None, :, None, :], sin[None, :, None, :]
    out = np.empty_like(x)
    out[..., 0::2] = xr * c - xi * s
    out[..., 1::2] = xr * s + xi * c
    return out


def dilations(n_layers, r):
    log2 = int(np.log2(r)) + 1
    return [2 ** (n % log2) for n in range(n_layers)]


# --------------------------------------------------------------------------------------
# Dropout plumbing
# --------------------------------------------------------------------------------------
@dataclass
class Dropout:
    """How the functional dropout sites behave in one forward.

    mode 'off'    : sites are identity (reference behaviour when cfg.dropout == 0)
    mode 'inject' : enc_masks [n_enc, B, L, d] / conv_masks [n_conv, B, L, D] uint8 keep-masks in the
                    canonical slot order (H slots then L slots), as hd_forward/hd_sample take them
    mode 'philox' : masks generated from the counter-based contract above
    """
    mode: str = "off"
    enc_masks: Optional[np.ndarray] = None
    conv_masks: Optional[np.ndarray] = None
    seed: int = 0
    rows: Optional[np.ndarray] = None
    step: int = 0

    def keep(self, site_kind: str, layer: int, B: int, L: int, width: int, p: float):
        if self.mode == "off":
            return None
        if self.mode == "inject":
            m = (self.enc_masks if site_kind == "enc" else self.conv_masks)[layer]
            assert m.shape == (B, L, width), (m.shape, (B, L, width))
            return m
        if self.mode == "philox":
            site = layer if site_kind == "enc" else SITE_CONV_BASE + layer
            rows = self.rows if self.rows is not None else np.arange(B)
            return philox_keep_mask(self.seed, rows, self.step, site, L, width, p)
        raise ValueError(self.mode)


def apply_dropout(x, keep, p):
    if keep is None:
        return x
    return x * (keep.astype(x.dtype) * x.dtype.type(1.0 / (1.0 - p)))


# --------------------------------------------------------------------------------------
# The denoiser
# --------------------------------------------------------------------------------------
class OracleNet:
    """numpy restatement of AntiTFNet ('ab') / NanoAntiTFNet ('nb').

    ``sd`` maps the reference's state_dict keys to numpy arrays.
    """

    def __init__(self, kind: str, cfg: dict, sd: Dict[str, np.ndarray], dtype=F32):
        """dtype=np.float64 evaluates the same float32 weights in double precision (used by tests to
        separate float32 round-off of the path itself from implementation differences)."""
        assert kind in ("ab", "nb")
        self.kind = kind
        self.cfg = dict(cfg)
        self.dt = np.dtype(dtype).type
        self.sd = {k: np.asarray(v) for k, v in sd.items() if not np.iscomplexobj(v)}
        self.sd = {k: v.astype(F32).astype(self.dt) for k, v in self.sd.items()}
        c = self.cfg
        self.L = int(c["max_len"])
        self.d = int(c["d_model"])
        self.D = int(c["sum_d_model"])
        self.nhead = int(c["nhead"])
        self.att = int(c["att_model"])
        self.p_enc = float(c.get("dropout", 0.0))
        self.enc_act = ACT[c.get("activation", "relu")]
        # DualConv is built with its default activation='relu' (model/encoder/model.py:345),
        # NanoConv with its default 'gelu' (model/nanoencoder/model.py:242, 308).
        self.conv_act = relu if kind == "ab" else gelu
        self.conv_prefix = "dual_conv_block" if kind == "ab" else "nano_conv_block"
        self.segs = [(0, AB_H_LEN, "h_layers"), (AB_H_LEN, AB_H_LEN + AB_L_LEN, "l_layers")] \
            if kind == "ab" else [(0, self.L, "layers")]
        if kind == "ab":
            assert self.L == AB_H_LEN + AB_L_LEN
        self.enc_dil = dilations(int(c["n_encoder_layers"]), int(c["r"]))
        self.conv_dil = dilations(int(c["dual_layers"]), int(c["r"]))
        # buffers travel in the state_dict (register_buffer: model/encoder/model.py:70-78 `pe`, cross_attention.py:145-146
        # `rope`): load_state_dict overwrites the constructor's tables with the checkpoint's, so a state_dict that carries
        # them wins here too; otherwise they are rebuilt the way the constructor builds them
        if "pos_encoder.pos_embedding.pe" in sd:
            self.pe = np.asarray(sd["pos_encoder.pos_embedding.pe"]).reshape(self.L, -1).astype(F32).astype(self.dt)
        else:
            self.pe = sinusoid_pe(self.L, int(c["n_pos_model"])).astype(self.dt)
        self.sd.pop("pos_encoder.pos_embedding.pe", None)
        ropes = [np.asarray(v) for k, v in sd.items() if k.endswith(".rope")]
        if ropes:
            r = ropes[0]
            r = np.stack([r.real, r.imag], -1) if np.iscomplexobj(r) else r
            self.cos, self.sin = r[..., 0].astype(F32).astype(self.dt), r[..., 1].astype(F32).astype(self.dt)
        else:
            self.cos, self.sin = (t.astype(self.dt) for t in rope_table(self.att // self.nhead, self.L))
        self.sd = {k: v for k, v in self.sd.items() if not k.endswith(".rope")}
        self.trace: Optional[dict] = None     # set to {} to record intermediate activations

    # -- helpers -------------------------------------------------------------------
    def _rec(self, name, val):
        if self.trace is not None:
            self.trace[name] = np.array(val, copy=True)

    def _bytenet_block(self, x, pre, dil, act):
        """ByteNetBlock.forward: x + sequence2(conv(sequence1(x)))."""
        s = self.sd
        h = act(layer_norm(x, s[pre + "sequence1.0.weight"], s[pre + "sequence1.0.bias"]))
        h = linear(h, s[pre + "sequence1.2.conv.weight"][:, :, 0], s[pre + "sequence1.2.conv.bias"])
        h = act(layer_norm(h, s[pre + "sequence1.3.weight"], s[pre + "sequence1.3.bias"]))
        h = dilated_conv(h, s[pre + "conv.weight"], s[pre + "conv.bias"], dil)
        h = act(layer_norm(h, s[pre + "sequence2.0.weight"], s[pre + "sequence2.0.bias"]))
        h = linear(h, s[pre + "sequence2.2.conv.weight"][:, :, 0], s[pre + "sequence2.2.conv.bias"])
        return x + h

    def _conv_stack(self, x, prefix, dils, act, p, site_kind, drop: Dropout):
        """Per-segment ByteNet stacks + functional dropout after every block."""
        B, L, W = x.shape
        out = np.empty_like(x)
        active = p > 0.0 and drop.mode != "off"
        keeps = [drop.keep(site_kind, n, B, L, W, p) if active else None for n in range(len(dils))]
        for lo, hi, name in self.segs:
            xs = x[:, lo:hi, :]
            for n, dil in enumerate(dils):
                xs = self._bytenet_block(xs, f"{prefix}.{name}.{n}.", dil, act)
                if keeps[n] is not None:
                    xs = apply_dropout(xs, keeps[n][:, lo:hi, :], p)
            out[:, lo:hi, :] = xs
        return out

    def static_embed(self, region, chain):
        """Token-independent branch: pos [B,L,d] (+ chn [B,L,d] for 'ab')."""
        s = self.sd
        B = region.shape[0]
        x = s["region_encoder.region_embedding.weight"][region]
        x = relu(layer_norm(x, s["region_encoder.region_layer1.0.weight"], s["region_encoder.region_layer1.0.bias"]))
        x = linear(x, s["region_encoder.region_layer1.2.conv.weight"][:, :, 0], s["region_encoder.region_layer1.2.conv.bias"])
        x = relu(layer_norm(x, s["region_encoder.region_layer1.3.weight"], s["region_encoder.region_layer1.3.bias"]))
        x = x + self.pe[None, :, :]
        m = gelu(linear(x, s["pos_encoder.pos_lin.ln1.weight"], s["pos_encoder.pos_lin.ln1.bias"]))
        m = linear(m, s["pos_encoder.pos_lin.ln2.weight"], s["pos_encoder.pos_lin.ln2.bias"])
        pos = x + m
        chn = None
        if self.kind == "ab":
            chain = np.asarray(chain).reshape(-1)
            assert chain.shape[0] == 2 * B
            e = s["side_encoder.side_embeddinng.weight"][chain]                      # [2B, 4]
            e = linear(e, s["side_encoder.side_mlp.0.weight"], s["side_encoder.side_mlp.0.bias"])
            e = relu(layer_norm(e, s["side_encoder.side_mlp.1.weight"], s["side_encoder.side_mlp.1.bias"]))
            e = linear(e, s["side_encoder.side_mlp.3.weight"], s["side_encoder.side_mlp.3.bias"])
            # SideEmbedder.forward (model.py:197-205): rows with side==0 are tiled over the 152 heavy
            # slots, rows with side!=0 over the 139 light slots, in order of appearance.
            h_rows, l_rows = e[chain == 0], e[chain != 0]
            assert h_rows.shape[0] == B and l_rows.shape[0] == B
            chn = np.concatenate([np.repeat(h_rows[:, None, :], AB_H_LEN, axis=1),
                                  np.repeat(l_rows[:, None, :], AB_L_LEN, axis=1)], axis=1)
        return pos.astype(self.dt), (None if chn is None else chn.astype(self.dt))

    def _attn(self, x, pre):
        """AttLayer.forward (cross_attention.py:149-173), context=None."""
        s = self.sd
        B, L, _ = x.shape
        H, hd = self.nhead, self.att // self.nhead
        q = linear(x, s[pre + "query.weight"], s[pre + "query.bias"]).reshape(B, L, H, hd)
        k = linear(x, s[pre + "key.weight"], s[pre + "key.bias"]).reshape(B, L, H, hd)
        v = linear(x, s[pre + "value.weight"], s[pre + "value.bias"]).reshape(B, L, H, hd)
        q, k = apply_rope(q, self.cos, self.sin), apply_rope(k, self.cos, self.sin)
        q, k, v = (t.transpose(0, 2, 1, 3) for t in (q, k, v))
        w = (q @ k.transpose(0, 1, 3, 2)) / self.dt(math.sqrt(self.att / self.nhead))
        w = w - w.max(axis=-1, keepdims=True)
        w = np.exp(w)
        w = w / w.sum(axis=-1, keepdims=True, dtype=self.dt)
        o = (w @ v).transpose(0, 2, 1, 3).reshape(B, L, H * hd)
        return linear(o, s[pre + "out_put.weight"], s[pre + "out_put.bias"])

    def _self_att_block(self, x, n):
        """SelfAttBlock.forward (cross_attention.py:273-287) -- last residual is the block INPUT."""
        s = self.sd
        pre = f"self_at.layers.{n}."
        at = x + self._attn(x, pre + "attn_hl.")
        at = at + self._attn(layer_norm(at, s[pre + "norm_hl1.weight"], s[pre + "norm_hl1.bias"]), pre + "attn_hl_c.")
        f = layer_norm(at, s[pre + "norm_hl2.weight"], s[pre + "norm_hl2.bias"])
        f = relu(linear(f, s[pre + "ff_hl.0.weight"], s[pre + "ff_hl.0.bias"]))
        f = linear(f, s[pre + "ff_hl.2.weight"], s[pre + "ff_hl.2.bias"])
        return f + x

    # -- forward -------------------------------------------------------------------
    def forward(self, tokens, region, chain=None, dropout: Optional[Dropout] = None, static=None):
        """-> logits float32 [B, L, 23]."""
        drop = dropout or Dropout("off")
        s = self.sd
        tokens = np.asarray(tokens).astype(np.int64)
        region = np.asarray(region).astype(np.int64)
        B, L = tokens.shape
        assert L == self.L, "RoPE asserts L == rolength (cross_attention.py:29-30)"
        e = s["aa_encoder.embedder.weight"][tokens]
        self._rec("embed", e)
        e = self._conv_stack(e, "aa_encoder", self.enc_dil, self.enc_act, self.p_enc, "enc", drop)
        self._rec("aa_encoder", e)
        pos, chn = static if static is not None else self.static_embed(region, chain)
        self._rec("pos", pos)
        if self.kind == "ab":
            self._rec("chn", chn)
            emb = e + pos + chn
            feat = np.concatenate([emb, pos, chn], axis=-1)
        else:
            emb = e + pos
            feat = np.concatenate([emb, pos], axis=-1)
        self._rec("feature", feat)
        p_conv = 0.5 if self.p_enc > 0.0 else 0.0      # F.dropout(x) default p, gated by cfg.dropout > 0
        h = self._conv_stack(feat.astype(self.dt), self.conv_prefix, self.conv_dil, self.conv_act, p_conv, "conv", drop)
        self._rec("conv", h)
        for n in range(int(self.cfg["cs_layers"])):
            h = self._self_att_block(h, n)
            self._rec(f"att{n}", h)
        h = layer_norm(h, s["last_norm.weight"], s["last_norm.bias"])
        logits = linear(h, s["decoder.weight"], s["decoder.bias"])
        return logits.astype(self.dt)

    __call__ = forward


# --------------------------------------------------------------------------------------
# Sampling
# --------------------------------------------------------------------------------------
def categorical_from_logits(logits22, q):
    """softmax over the first 22 logits, then torch.multinomial(p,1) == argmax(p / q) (first max wins)."""
    z = logits22 - logits22.max(axis=-1, keepdims=True)
    ez = np.exp(z)
    p = ez / ez.sum(axis=-1, keepdims=True, dtype=F32)
    return np.argmax(p / q, axis=-1), p


def sample(net: OracleNet, tokens, region, chain, order, T, *, seed=0, row0=0, q_noise=None,
           dropout_mode="off", enc_masks=None, conv_masks=None, trace: Optional[list] = None):
    """Order-agnostic autoregressive sampling (sample.py:499-513) for independent rows.

    tokens [B, L] int (returned copy is filled in), order [B, Tmax] slot visited by row b at step t,
    T [B] number of steps of row b (rows with t >= T[b] are left untouched at step t).
    q_noise [Tmax, B, 22] float32 injects the Exp(1) noise; otherwise the Philox contract is used with
    global row ids row0 + b.  dropout_mode: 'off' | 'philox' | 'inject' (masks [Tmax, n, B, L, w]).
    """
    tokens = np.array(tokens, dtype=np.int64, copy=True)
    order = np.asarray(order)
    T = np.asarray(T)
    B = tokens.shape[0]
    rows = np.arange(B) + row0
    static = net.static_embed(np.asarray(region).astype(np.int64), chain)
    for t in range(int(T.max()) if B else 0):
        if dropout_mode == "inject":
            drop = Dropout("inject", enc_masks=enc_masks[t], conv_masks=conv_masks[t])
        elif dropout_mode == "philox":
            drop = Dropout("philox", seed=seed, rows=rows, step=t)
        else:
            drop = Dropout("off")
        logits = net.forward(tokens, region, chain, dropout=drop, static=static)
        active = t < T
        slot = np.where(active, order[:, min(t, order.shape[1] - 1)], 0)
        lg = logits[np.arange(B), slot, :N_SAMPLE]
        q = q_noise[t] if q_noise is not None else philox_exp_noise(seed, rows, t)
        s, p = categorical_from_logits(lg, q)
        if trace is not None:
            trace.append({"slot": slot.copy(), "active": active.copy(), "probs": p.copy(),
                          "sampled": s.copy(), "logits": lg.copy()})
        tokens[np.arange(B)[active], slot[active]] = s[active]
    return tokens


def flops_per_forward(cfg: dict) -> float:
    """Algorithmic FLOPs of one forward of one sequence (SURVEY.md §8d formula)."""
    L, d, D, A, Fd = (int(cfg[k]) for k in ("max_len", "d_model", "sum_d_model", "att_model", "dim_feedforward"))
    dh, Dh = d // 2, D // 2
    k = int(cfg["aa_kernel_size"])
    return L * (int(cfg["n_encoder_layers"]) * (4 * d * dh + 2 * k * dh * dh)
                + int(cfg["dual_layers"]) * (4 * D * Dh + 2 * k * Dh * Dh)
                + 2 * int(cfg["cs_layers"]) * (8 * D * A + 4 * L * A)
                + int(cfg["cs_layers"]) * 4 * D * Fd + 2 * D * N_TOKENS)
