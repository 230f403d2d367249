"""TEST INFRASTRUCTURE ONLY -- the CPU restatement of oracle/hudiff_oracle.py evaluated with PyTorch-CPU kernels.

SURVEY.md §8d asks for the CPU baseline "on the build's CPU restatement and, where torch is present on the box, the same
loop on PyTorch-CPU".  ``TorchOracleNet`` is the same network as ``hudiff_oracle.OracleNet`` (same state_dict keys, same
reference quirks, same counter-based dropout / sampling noise, all of which it inherits or receives from that module);
only the tensor algebra runs through ``torch.nn.functional`` (MKL / oneDNN, all host threads) instead of numpy + OpenBLAS.
``hudiff_oracle.sample(net, ...)`` drives it unchanged.  Only ``tests/`` and ``bench.py``'s ``cpu_baseline`` leg import
this file; the product path (``hudiff_amd``) never does.  tests/test_oracle_golden.py pins it to the numpy oracle.
"""
from __future__ import annotations

import math
from typing import Optional

import numpy as np

import hudiff_oracle as ho


class TorchOracleNet(ho.OracleNet):
    def __init__(self, kind, cfg, sd):
        super().__init__(kind, cfg, sd)                    # numpy float32 copies, dilations, pe, rope tables
        import torch
        self.torch = torch
        self.F = torch.nn.functional
        self.tw = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in self.sd.items()}
        self.pe_t = torch.from_numpy(np.ascontiguousarray(self.pe))
        self.cos_t = torch.from_numpy(np.ascontiguousarray(self.cos))[None, :, None, :]
        self.sin_t = torch.from_numpy(np.ascontiguousarray(self.sin))[None, :, None, :]
        self.act_t = {ho.gelu: self.F.gelu, ho.relu: self.F.relu}

    # -- helpers -------------------------------------------------------------------
    def _ln(self, x, name):
        return self.F.layer_norm(x, (x.shape[-1],), self.tw[name + ".weight"], self.tw[name + ".bias"], 1e-5)

    def _lin(self, x, name, conv1x1=False):
        w = self.tw[name + ".weight"]
        return self.F.linear(x, w[:, :, 0] if conv1x1 else w, self.tw[name + ".bias"])

    def _bytenet_block(self, x, pre, dil, act):
        """ByteNetBlock.forward (sequence_models): x + sequence2(conv(sequence1(x))); x [B, Lc, C]."""
        a = self.act_t[act]
        h = a(self._ln(x, pre + "sequence1.0"))
        h = self._lin(h, pre + "sequence1.2.conv", True)
        h = a(self._ln(h, pre + "sequence1.3"))
        w = self.tw[pre + "conv.weight"]
        k = w.shape[2]
        h = self.F.conv1d(h.transpose(1, 2), w, self.tw[pre + "conv.bias"], padding=dil * (k - 1) // 2,
                          dilation=dil).transpose(1, 2)
        h = a(self._ln(h, pre + "sequence2.0"))
        h = self._lin(h, pre + "sequence2.2.conv", True)
        return x + h

    def _conv_stack(self, x, prefix, dils, act, p, site_kind, drop):
        torch = self.torch
        B, L, W = x.shape
        active = p > 0.0 and drop.mode != "off"
        keeps = [drop.keep(site_kind, n, B, L, W, p) if active else None for n in range(len(dils))]
        outs = []
        for lo, hi, name in self.segs:
            xs = x[:, lo:hi, :]
            for n, dil in enumerate(dils):
                xs = self._bytenet_block(xs, f"{prefix}.{name}.{n}.", dil, act)
                if keeps[n] is not None:
                    xs = xs * (torch.from_numpy(np.ascontiguousarray(keeps[n][:, lo:hi, :])).to(xs.dtype) * (1.0 / (1.0 - p)))
            outs.append(xs)
        return torch.cat(outs, dim=1)

    def static_embed(self, region, chain):
        torch, F = self.torch, self.F
        region = torch.from_numpy(np.asarray(region).astype(np.int64))
        B = region.shape[0]
        x = self.tw["region_encoder.region_embedding.weight"][region]
        x = F.relu(self._ln(x, "region_encoder.region_layer1.0"))
        x = self._lin(x, "region_encoder.region_layer1.2.conv", True)
        x = F.relu(self._ln(x, "region_encoder.region_layer1.3"))
        x = x + self.pe_t[None]
        m = F.gelu(self._lin(x, "pos_encoder.pos_lin.ln1"))
        pos = x + self._lin(m, "pos_encoder.pos_lin.ln2")
        chn = None
        if self.kind == "ab":
            c = torch.from_numpy(np.asarray(chain).reshape(-1).astype(np.int64))
            e = self.tw["side_encoder.side_embeddinng.weight"][c]
            e = self._lin(e, "side_encoder.side_mlp.0")
            e = F.relu(self._ln(e, "side_encoder.side_mlp.1"))
            e = self._lin(e, "side_encoder.side_mlp.3")
            h_rows, l_rows = e[c == 0], e[c != 0]
            assert h_rows.shape[0] == B and l_rows.shape[0] == B
            chn = torch.cat([h_rows[:, None, :].expand(B, ho.AB_H_LEN, -1), l_rows[:, None, :].expand(B, ho.AB_L_LEN, -1)], dim=1)
        return pos, chn

    def _rope(self, x):
        xr, xi = x[..., 0::2], x[..., 1::2]
        out = self.torch.empty_like(x)
        out[..., 0::2] = xr * self.cos_t - xi * self.sin_t
        out[..., 1::2] = xr * self.sin_t + xi * self.cos_t
        return out

    def _attn(self, x, pre):
        B, L, _ = x.shape
        H, hd = self.nhead, self.att // self.nhead
        q = self._rope(self._lin(x, pre + "query").reshape(B, L, H, hd)).transpose(1, 2)
        k = self._rope(self._lin(x, pre + "key").reshape(B, L, H, hd)).transpose(1, 2)
        v = self._lin(x, pre + "value").reshape(B, L, H, hd).transpose(1, 2)
        w = self.torch.softmax((q @ k.transpose(-1, -2)) / math.sqrt(hd), dim=-1)
        o = (w @ v).transpose(1, 2).reshape(B, L, H * hd)
        return self._lin(o, pre + "out_put")

    def _self_att_block(self, x, n):
        pre = f"self_at.layers.{n}."
        at = x + self._attn(x, pre + "attn_hl.")
        self._rec(f"att{n}_at1", at.numpy())
        at = at + self._attn(self._ln(at, pre + "norm_hl1"), pre + "attn_hl_c.")
        self._rec(f"att{n}_at2", at.numpy())
        f = self.F.relu(self._lin(self._ln(at, pre + "norm_hl2"), pre + "ff_hl.0"))
        return self._lin(f, pre + "ff_hl.2") + x

    def forward(self, tokens, region, chain=None, dropout: Optional[ho.Dropout] = None, static=None):
        """-> logits float32 numpy [B, L, 23] (numpy in, numpy out: ``hudiff_oracle.sample`` drives this unchanged)."""
        torch = self.torch
        drop = dropout or ho.Dropout("off")
        with torch.no_grad():
            tok = torch.from_numpy(np.asarray(tokens).astype(np.int64))
            assert tok.shape[1] == self.L
            e = self.tw["aa_encoder.embedder.weight"][tok]
            e = self._conv_stack(e, "aa_encoder", self.enc_dil, self.enc_act, self.p_enc, "enc", drop)
            self._rec("aa_encoder", e.numpy())             # (the same trace keys as hudiff_oracle.OracleNet.forward)
            pos, chn = static if static is not None else self.static_embed(region, chain)
            self._rec("pos", pos.numpy())
            if self.kind == "ab":
                self._rec("chn", chn.numpy())
                feat = torch.cat([e + pos + chn, pos, chn], dim=-1)
            else:
                feat = torch.cat([e + pos, pos], dim=-1)
            p_conv = 0.5 if self.p_enc > 0.0 else 0.0
            h = self._conv_stack(feat, self.conv_prefix, self.conv_dil, self.conv_act, p_conv, "conv", drop)
            self._rec("conv", h.numpy())
            for n in range(int(self.cfg["cs_layers"])):
                h = self._self_att_block(h, n)
                self._rec(f"att{n}", h.numpy())
            h = self._ln(h, "last_norm")
            self._rec("last_norm", h.numpy())
            return self._lin(h, "decoder").numpy()

    __call__ = forward
