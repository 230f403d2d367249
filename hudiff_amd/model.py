"""Host-side mirror of the reference's model-call boundary, backed by libhudiff_hip.so.

Reference interface mirrored (file:line under /root/reference):
  * ``AntiTFNet(**config.model)`` / ``NanoAntiTFNet(**config.model)``   model/encoder/model.py:325-349,
    model/nanoencoder/model.py:290-311 -- same constructor keywords
  * ``model_selected(config)``                                          utils/train_utils.py:43-55
  * ``model.load_state_dict(ckpt['model'])`` / ``.eval()`` / ``.to(device)``   antibody_scripts/sample.py:456-458
  * ``model(H_L_seq, H_L_region_type, H_L_chn_type) -> logits[B, L, n_tokens]``  model/encoder/model.py:366-384
so the reference-shaped loop (sample.py:499-513) can drive it unchanged; ``.sample`` runs the whole loop
on the device (hd_sample).  torch is used only to read checkpoints / to hand tensors back in the
caller's type -- no torch op is on the hot path.
"""
from __future__ import annotations

import ctypes as C
from typing import Mapping, Optional

import numpy as np

from . import _lib as L

_ACT = {"relu": L.HD_ACT_RELU, "gelu": L.HD_ACT_GELU}
AB_H_LEN = 152


def _get(cfg, key, default=None):
    if isinstance(cfg, Mapping):
        return cfg.get(key, default)
    return getattr(cfg, key, default)


def _to_numpy(x):
    if x is None:
        return None, False
    if hasattr(x, "detach") and hasattr(x, "cpu"):       # torch.Tensor without importing torch
        return x.detach().cpu().numpy(), True
    return np.asarray(x), False


def _sinusoid_pe(max_len, d_model):
    f32 = np.float32
    position = np.arange(max_len, dtype=f32)[:, None]
    div = np.exp(np.arange(0, d_model, 2).astype(f32) * f32(-np.log(10000.0) / d_model)).astype(f32)
    ang = (position * div).astype(f32)
    pe = np.zeros((max_len, d_model), dtype=f32)
    pe[:, 0::2], pe[:, 1::2] = np.sin(ang), np.cos(ang)
    return pe


def _rope_table(head_dim, length, theta=10000.0):
    f32 = np.float32
    freqs = (f32(1.0) / (f32(theta) ** (np.arange(0, head_dim, 2)[: head_dim // 2].astype(f32) / f32(head_dim)))).astype(f32)
    ang = np.outer(np.arange(length, dtype=f32), freqs).astype(f32)
    return np.stack([np.cos(ang), np.sin(ang)], axis=-1).astype(f32)


class _Denoiser:
    kind = None            # 'ab' | 'nb'

    def __init__(self, n_tokens, d_embedding, d_model, n_encoder_layers, aa_kernel_size, r,
                 n_region, r_embedding, r_model, n_pos_model, max_len, sum_d_model, dual_layers,
                 att_model, dim_feedforward, nhead, cs_layers, n_side=3, s_embedding=4, s_model=None,
                 rank=None, n_frozen_embs=None, padding_idx=None, causal=False, dropout=0.0, slim=True,
                 activation="relu", down_embed=False, timesteps=None, device=0, precision=None, options=None):
        if rank is not None or n_frozen_embs is not None or causal or not slim or down_embed or padding_idx is not None:
            raise NotImplementedError("only the configuration HuDiff ships (rank=None, causal=False, slim=True, "
                                      "down_embed=False, padding_idx=None) is implemented")
        if not (d_embedding == d_model == r_model == n_pos_model) or (s_model not in (None, d_model)):
            raise ValueError("d_embedding, d_model, r_model, n_pos_model (and s_model) must be equal")
        self.config = dict(n_tokens=n_tokens, d_embedding=d_embedding, d_model=d_model,
                           n_encoder_layers=n_encoder_layers, aa_kernel_size=aa_kernel_size, r=r,
                           n_region=n_region, r_embedding=r_embedding, r_model=r_model, n_pos_model=n_pos_model,
                           max_len=max_len, sum_d_model=sum_d_model, dual_layers=dual_layers, att_model=att_model,
                           dim_feedforward=dim_feedforward, nhead=nhead, cs_layers=cs_layers, dropout=float(dropout),
                           activation=activation)
        if self.kind == "ab":
            self.config.update(n_side=n_side, s_embedding=s_embedding, s_model=d_model)
        c = L.HdConfig()
        c.abi_version = L.HD_ABI_VERSION
        c.kind = L.HD_KIND_ANTIBODY if self.kind == "ab" else L.HD_KIND_NANOBODY
        c.n_tokens, c.max_len = n_tokens, max_len
        c.h_len = AB_H_LEN if self.kind == "ab" else max_len
        c.d_model, c.sum_d_model = d_model, sum_d_model
        c.n_encoder_layers, c.dual_layers, c.kernel_size, c.r = n_encoder_layers, dual_layers, aa_kernel_size, r
        c.att_model, c.nhead, c.dim_feedforward, c.cs_layers = att_model, nhead, dim_feedforward, cs_layers
        c.n_region, c.r_embedding = n_region, r_embedding
        c.n_side, c.s_embedding = (n_side, s_embedding) if self.kind == "ab" else (0, 0)
        c.enc_act = _ACT[activation]
        # DualConv is constructed with its default activation='relu' (model/encoder/model.py:345),
        # NanoConv with its default 'gelu' (model/nanoencoder/model.py:242, 308)
        c.conv_act = L.HD_ACT_RELU if self.kind == "ab" else L.HD_ACT_GELU
        c.dropout = float(dropout)
        self._cfg = c
        self.max_len, self.n_tokens = max_len, n_tokens
        self._lib = L.load()
        self._h = C.c_void_p()
        L.check(self._lib.hd_create(C.byref(c), int(device), C.byref(self._h)))
        # precision route (include/hudiff_hip.h "precision routes"; not a keyword of the reference's constructor, which computes in
        # fp32 on whatever torch gives it): None / "default" = the library default (split precision; the environment may override
        # the default), "split" | "f32_gemm" | "f32_all" = that route whatever the environment says
        if precision is not None:
            if precision not in L.PRECISIONS:
                raise ValueError(f"precision must be one of {sorted(L.PRECISIONS)}, got {precision!r}")
            L.check(self._lib.hd_set_precision(self._h, L.PRECISIONS[precision]))
        self._loaded = False
        self._seen_fallbacks = (0, 0)
        self.device_index = int(device)
        # tuning options (include/hudiff_hip.h "tuning options"): {name: value}, e.g. {"lanes": 1, "lnsync_level": 0}
        for k, v in (options or {}).items():
            self.set_option(k, v)

    # -- tuning options (hd_set_option / hd_get_option) --------------------------------------------
    @staticmethod
    def _option_id(name):
        if isinstance(name, str):
            key = name.lower().removeprefix("hd_opt_")
            if key not in L.OPTIONS:
                raise ValueError(f"unknown option {name!r}; known: {sorted(L.OPTIONS)}")
            return L.OPTIONS[key]
        return int(name)

    def set_option(self, name, value):
        """An explicit choice of a kernel-selecting knob; wins over the HUDIFF_* environment variable of the same meaning."""
        L.check(self._lib.hd_set_option(self._h, self._option_id(name), int(value)))
        return self

    def get_option(self, name):
        v = C.c_int64()
        L.check(self._lib.hd_get_option(self._h, self._option_id(name), C.byref(v)))
        return int(v.value)

    def options(self):
        return {n: self.get_option(i) for n, i in L.OPTIONS.items()}

    # -- nn.Module-shaped surface ---------------------------------------------------------------
    def load_state_dict(self, state_dict, strict=True):
        if self._loaded:
            raise RuntimeError("weights already loaded into this handle")
        if not strict:
            raise NotImplementedError("strict=False is not supported")
        keys = set()
        for key, val in state_dict.items():
            arr, _ = _to_numpy(val)
            if np.iscomplexobj(arr):             # '...rope' complex64 [L, hd/2] -> float32 [L, hd/2, 2]
                arr = np.stack([arr.real, arr.imag], axis=-1)
            self._load(key, arr)
            keys.add(key[7:] if key.startswith("module.") else key)
        # Buffers a stripped state_dict may lack: built the way torch builds them (float32 arithmetic),
        # model/encoder/model.py:70-78 and model/encoder/cross_attention.py:35-56.
        if "pos_encoder.pos_embedding.pe" not in keys:
            self._load("pos_encoder.pos_embedding.pe", _sinusoid_pe(self.max_len, self.config["d_model"])[:, None, :])
        if not any(k.endswith(".rope") for k in keys):
            self._load("self_at.layers.0.attn_hl.rope",
                       _rope_table(self.config["att_model"] // self.config["nhead"], self.max_len))
        L.check(self._lib.hd_finalize(self._h))
        self._loaded = True
        return self

    def _load(self, key, arr):
        arr = np.ascontiguousarray(arr, dtype=np.float32)
        shape = (C.c_int64 * max(arr.ndim, 1))(*arr.shape)
        L.check(self._lib.hd_load_tensor(self._h, key.encode(), L.ptr(arr, C.c_float), shape, arr.ndim))

    def eval(self):
        return self

    def to(self, *a, **k):
        return self

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.hd_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _warn_on_guard(self):
        """A guard of the split-precision kernels repeated the last call on safer kernels (results are correct; the handle is slower
        from here on until precision_reset()): say so once per event -- the C library only counts it (hd_precision_report)."""
        r = L.HdPrecisionInfo()
        if self._lib.hd_precision_report(self._h, C.byref(r), C.sizeof(r)) != L.HD_OK:
            return
        now = (int(r.range_fallbacks), int(r.lnsync_fallbacks))
        if now != self._seen_fallbacks:
            import warnings
            what = []
            if now[0] > self._seen_fallbacks[0]:
                what.append("an activation left the fp16 range (|x| >= 65504): the call was repeated on the all-fp32 kernels and the handle stays on them")
            if now[1] > self._seen_fallbacks[1]:
                what.append("an ln_sync meeting failed: the call was repeated with separate LayerNorm passes and the handle keeps them")
            self._seen_fallbacks = now
            warnings.warn("hudiff_amd: " + "; ".join(what) + " (precision_info(); precision_reset() re-arms the split kernels)", RuntimeWarning, stacklevel=3)

    # -- forward ---------------------------------------------------------------------------------
    def _prep(self, tokens, region, chain):
        tok, was_torch = _to_numpy(tokens)
        reg, _ = _to_numpy(region)
        chn, _ = _to_numpy(chain)
        if tok.ndim != 2 or tok.shape[1] != self.max_len:
            # the reference asserts L == rolength in RoPE (cross_attention.py:29-30)
            raise AssertionError(f"tokens must be [B, {self.max_len}], got {tok.shape}")
        B = tok.shape[0]
        tok = L.as_i32(tok)
        reg = L.as_i32(reg, (B, self.max_len))
        if self.kind == "ab":
            if chn is None:
                raise ValueError("AntiTFNet needs H_L_chn_type [2B]")
            chn = L.as_i32(np.asarray(chn).reshape(-1), (2 * B,))
        else:
            chn = None
        return tok, reg, chn, B, was_torch

    @staticmethod
    def _flags(dropout, graph=True, prune=True, lanes=2):
        f = {"faithful": L.HD_DROPOUT_FAITHFUL, "off": L.HD_DROPOUT_OFF, "inject": L.HD_DROPOUT_INJECT}[dropout]
        # graph: True = one hipGraph per step replayed T times, "loop" = the whole T-step loop as one hipGraph, False = eager
        return (f | (0 if graph else L.HD_NO_GRAPH) | (L.HD_LOOP_GRAPH if graph == "loop" else 0) | (0 if prune else L.HD_NO_PRUNE)
                | (0 if lanes == 2 else L.HD_ONE_LANE))

    def forward(self, H_L_seq, H_L_region_type, H_L_chn_type=None, *, dropout="faithful", seed=0, row0=0,
                step=0, enc_masks=None, conv_masks=None):
        if not self._loaded:
            raise RuntimeError("load_state_dict first")
        tok, reg, chn, B, was_torch = self._prep(H_L_seq, H_L_region_type, H_L_chn_type)
        logits = np.empty((B, self.max_len, self.n_tokens), dtype=np.float32)
        em = None if enc_masks is None else np.ascontiguousarray(enc_masks, dtype=np.uint8)
        cm = None if conv_masks is None else np.ascontiguousarray(conv_masks, dtype=np.uint8)
        L.check(self._lib.hd_forward(self._h, L.ptr(tok, C.c_int32), L.ptr(reg, C.c_int32), L.ptr(chn, C.c_int32),
                                     B, self._flags(dropout), int(seed), int(row0), int(step),
                                     L.ptr(em, C.c_uint8), L.ptr(cm, C.c_uint8), L.ptr(logits, C.c_float)))
        self._warn_on_guard()
        if was_torch:
            import torch
            return torch.from_numpy(logits)
        return logits

    __call__ = forward

    # -- whole loop on the device ----------------------------------------------------------------
    def _sample_args(self, tokens, region, chain, order, T, q_noise, enc_masks, conv_masks):
        tok, reg, chn, B, was_torch = self._prep(tokens, region, chain)
        order = np.asarray(order)
        Tmax = int(order.shape[1]) if order.ndim == 2 else 0
        order = L.as_i32(order.reshape(B, Tmax), (B, Tmax))
        T = L.as_i32(np.asarray(T).reshape(-1), (B,))
        q = None if q_noise is None else np.ascontiguousarray(q_noise, dtype=np.float32)
        if q is not None and q.shape != (Tmax, B, 22):
            raise ValueError(f"q_noise must be [{Tmax}, {B}, 22], got {q.shape}")
        em = None if enc_masks is None else np.ascontiguousarray(enc_masks, dtype=np.uint8)
        cm = None if conv_masks is None else np.ascontiguousarray(conv_masks, dtype=np.uint8)
        return tok, reg, chn, order, T, B, Tmax, q, em, cm, was_torch

    def sample(self, tokens, region, chain, order, T, *, seed=0, row0=0, q_noise=None, dropout="faithful",
               enc_masks=None, conv_masks=None, graph=True, prune=True, lanes=2):
        """Run the T-step loop (sample.py:499-513) for B independent rows; returns the filled tokens."""
        tok, reg, chn, order, T, B, Tmax, q, em, cm, was_torch = self._sample_args(
            tokens, region, chain, order, T, q_noise, enc_masks, conv_masks)
        out = tok.copy()
        L.check(self._lib.hd_sample(self._h, L.ptr(out, C.c_int32), L.ptr(reg, C.c_int32), L.ptr(chn, C.c_int32),
                                    L.ptr(order, C.c_int32), L.ptr(T, C.c_int32), B, Tmax,
                                    self._flags(dropout, graph, prune, lanes), int(seed), int(row0), L.ptr(q, C.c_float),
                                    L.ptr(em, C.c_uint8), L.ptr(cm, C.c_uint8)))
        self._warn_on_guard()
        if was_torch:
            import torch
            return torch.from_numpy(out.astype(np.int64))
        return out

    def sample_begin(self, tokens, region, chain, order, T, *, seed=0, row0=0, q_noise=None, dropout="faithful",
                     enc_masks=None, conv_masks=None, graph=True, prune=True, lanes=2):
        tok, reg, chn, order, T, B, Tmax, q, em, cm, _ = self._sample_args(
            tokens, region, chain, order, T, q_noise, enc_masks, conv_masks)
        L.check(self._lib.hd_sample_begin(self._h, L.ptr(tok, C.c_int32), L.ptr(reg, C.c_int32), L.ptr(chn, C.c_int32),
                                          L.ptr(order, C.c_int32), L.ptr(T, C.c_int32), B, Tmax,
                                          self._flags(dropout, graph, prune, lanes), int(seed), int(row0), L.ptr(q, C.c_float),
                                          L.ptr(em, C.c_uint8), L.ptr(cm, C.c_uint8)))
        self._session_B = B

    def sample_run(self, t0, t1):
        L.check(self._lib.hd_sample_run(self._h, int(t0), int(t1)))

    def sample_restart(self, seed):
        L.check(self._lib.hd_sample_restart(self._h, int(seed)))

    def sync(self):
        L.check(self._lib.hd_sync(self._h))

    def sample_end(self):
        out = np.empty((self._session_B, self.max_len), dtype=np.int32)
        L.check(self._lib.hd_sample_end(self._h, L.ptr(out, C.c_int32)))
        self._warn_on_guard()
        return out

    def sample_tokens(self):
        """Tokens of the open session as they stand (hd_sample_tokens: synchronises, the session stays open)."""
        out = np.empty((self._session_B, self.max_len), dtype=np.int32)
        L.check(self._lib.hd_sample_tokens(self._h, L.ptr(out, C.c_int32)))
        return out

    def last_run_ms(self):
        ms, steps = C.c_float(), C.c_int32()
        L.check(self._lib.hd_last_run_ms(self._h, C.byref(ms), C.byref(steps)))
        return float(ms.value), int(steps.value)

    def precision_info(self):
        """hd_precision_report: {'precision': 'split' | 'f32_gemm' | 'f32_all' (the resolved route), 'split_built' (bit 0 GEMM weight
        images, bit 1 attention core), 'split_in_use', 'lnsync_in_use', 'range_fallbacks' (calls repeated on the fp32 kernels because
        an operand left the fp16 range), 'lnsync_fallbacks' (calls repeated with separate LayerNorm passes because an ln_sync meeting
        failed), 'last_call_repeated'}."""
        r = L.HdPrecisionInfo()
        L.check(self._lib.hd_precision_report(self._h, C.byref(r), C.sizeof(r)))
        return {"precision": L.PRECISION_NAMES[int(r.precision)], "split_built": int(r.split_built), "split_in_use": bool(r.split_in_use),
                "lnsync_in_use": bool(r.lnsync_in_use), "range_fallbacks": int(r.range_fallbacks),
                "lnsync_fallbacks": int(r.lnsync_fallbacks), "last_call_repeated": bool(r.last_call_repeated),
                "lnsync_cross_xcd": bool(r.lnsync_cross_xcd)}

    def precision_reset(self):
        """hd_precision_reset: back on the configured route after a guard switched kernels off."""
        L.check(self._lib.hd_precision_reset(self._h))
        r = L.HdPrecisionInfo()
        if self._lib.hd_precision_report(self._h, C.byref(r), C.sizeof(r)) == L.HD_OK:
            self._seen_fallbacks = (int(r.range_fallbacks), int(r.lnsync_fallbacks))

    def debug_fail_next_lnsync(self):
        L.check(self._lib.hd_debug_fail_next_lnsync(self._h))

    def debug_scatter_lnsync(self, on=True):
        L.check(self._lib.hd_debug_scatter_lnsync(self._h, 1 if on else 0))

    def debug_stop_after(self, stage):
        L.check(self._lib.hd_debug_stop_after(self._h, int(stage)))

    def debug_read(self, name, B):
        width = {"FEAT": "sum_d_model", "Y": "sum_d_model", "AT": "sum_d_model", "ATX": "sum_d_model", "YX": "sum_d_model", "O": "att_model"}.get(name, "d_model")
        w = 3 * self.config["att_model"] if name == "QKV" else self.config[width]
        out = np.empty((B, self.max_len, w), dtype=np.float32)
        L.check(self._lib.hd_debug_read(self._h, name.encode(), B, L.ptr(out, C.c_float), out.size))
        return out

    def flops_per_row_forward(self):
        return float(self._lib.hd_flops_per_row_forward(C.byref(self._cfg)))

    def flops_per_row_sample_step(self):
        return float(self._lib.hd_flops_per_row_sample_step(C.byref(self._cfg)))


class AntiTFNet(_Denoiser):
    """model/encoder/model.py:325 -- antibody (VH + VL, 291 slots) denoiser."""
    kind = "ab"


class NanoAntiTFNet(_Denoiser):
    """model/nanoencoder/model.py:290 -- nanobody (VHH, 152 slots) denoiser."""
    kind = "nb"

    def __init__(self, n_tokens, d_embedding, d_model, n_encoder_layers, aa_kernel_size, r, n_region, r_embedding,
                 r_model, n_pos_model, max_len, sum_d_model, dual_layers, att_model, dim_feedforward, nhead, cs_layers,
                 **kw):
        super().__init__(n_tokens, d_embedding, d_model, n_encoder_layers, aa_kernel_size, r, n_region, r_embedding,
                         r_model, n_pos_model, max_len, sum_d_model, dual_layers, att_model, dim_feedforward, nhead,
                         cs_layers, **kw)


def model_selected(config, pretrained_model=None, tokenizer=None, device=0, precision=None, options=None):
    """utils/train_utils.py:43-55 for the two inference models (the training-only wrappers are out of scope)."""
    name = _get(config, "name")
    params = dict(_get(config, "model"))
    if name == "trans_oadm":
        return AntiTFNet(**params, device=device, precision=precision, options=options)
    if name == "nano":
        return NanoAntiTFNet(**params, device=device, precision=precision, options=options)
    raise NotImplementedError(f"config.name={name!r}: only 'trans_oadm' and 'nano' are sampling models")


def device_info(device=0):
    lib = L.load()
    name = C.create_string_buffer(256)
    cu, mem = C.c_int32(), C.c_int64()
    L.check(lib.hd_device_info(int(device), name, 256, C.byref(cu), C.byref(mem)))
    return {"name": name.value.decode(), "cu_count": int(cu.value), "hbm_bytes": int(mem.value)}


def device_count():
    return int(L.load().hd_device_count())
