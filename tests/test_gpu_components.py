"""Per-component HIP parity (VERDICT r5 "Next" #4): every row a3 .. a11 of SURVEY.md section 8a has a named check of the HIP
buffers behind it -- not only of the logits at the end, where a compensating pair of errors would not be attributed.

Two yard-sticks:
* the REFERENCE's own sub-module outputs (forward hooks on AntiTFNet / NanoAntiTFNet, oracle/make_golden.py and
  oracle/make_golden_deep.py: tests/golden/micro_{ab,nb}_forward.npz `act_*`, tests/golden/deep_{ab,nb}_acts.npz) at the micro and
  deep (production depth, 8 heads, dilations 1 .. 32) shapes, which run the fp32 kernels;
* the oracle's PyTorch-CPU twin (pinned to those vectors by tests/test_oracle_golden.py) at PRODUCTION width, where the split-precision
  kernels run (32 antibodies / 64 nanobodies = launches of >= 8192 rows), on all three precision routes.
The HIP side is read through the C ABI: hd_debug_stop_after / hd_debug_read (include/hudiff_hip.h).
Reference lines: model/encoder/model.py:366-384 (forward), :118-180 (ByteNetTime), :249-304 (DualConv), cross_attention.py:149-173
(AttLayer), :273-287 (SelfAttBlock)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, chain_or_none, load_cfg, load_deep, load_golden, load_weights

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    import hudiff_amd
    if hudiff_amd.device_count() < 1:
        pytest.fail("no MI355X visible: GPU tests must run on the GPU box (there is no CPU fallback)")
    return hudiff_amd


def _close(name, got, want, rtol):
    """max |got - want| <= rtol x max |want| (activations have no natural unit: the bound scales with the tensor)."""
    assert got.shape == want.shape, (name, got.shape, want.shape)
    err, ref = float(np.abs(got - want).max()), float(np.abs(want).max())
    assert np.isfinite(got).all() and err <= rtol * max(ref, 1e-3), (name, err, ref, rtol)
    return err / max(ref, 1e-3)


def _stages(m, kind, tok, reg, chn, B, d, n_att):
    """The HIP buffers of one forward, stage by stage -> dict keyed like the oracle's trace."""
    out = {}
    m.debug_stop_after(1)                                                  # behind the token encoder (+ static add)
    m(tok, reg, chn, dropout="off")
    feat, extra = m.debug_read("FEAT", B), m.debug_read("EXTRA", B)
    out["pos"] = m.debug_read("POS", B)                                    # a5: region + position branch (PosEmbedder output)
    if kind == "ab":
        out["chn"] = feat[:, :, 2 * d:3 * d].copy()                        # a5: side embedder, broadcast over the chain's slots
    out["aa_encoder"] = feat[:, :, :d] - extra                            # a3 / a4: ByteNet token encoder (FEAT's first third = e + pos [+ chn])
    out["feature"] = feat                                                  # a6: the concat
    for n in range(n_att):
        m.debug_stop_after(2 + n)                                          # in front of attention block n
        m(tok, reg, chn, dropout="off")
        out["conv" if n == 0 else f"att{n - 1}"] = m.debug_read("Y", B)    # a7 (n = 0) / a10: SelfAttBlock n - 1 output
        m.debug_stop_after(100 + n)                                        # right behind the first attention of block n
        m(tok, reg, chn, dropout="off")
        out[f"att{n}_at1"] = m.debug_read("AT", B)                         # a8 / a9: x + AttLayer(x), RoPE inside
    m.debug_stop_after(0)
    out["logits"] = m(tok, reg, chn, dropout="off")
    out[f"att{n_att - 1}"] = m.debug_read("Y", B)
    out[f"att{n_att - 1}_at2"] = m.debug_read("AT", B)                     # second sum of the last block (split route: decoded from ATX)
    return out


def _decoder_on(sd, act_last_norm):
    """a11: the decoder applied (in float64, on the host) to the REFERENCE's last_norm rows: what the logits must be if LN is right."""
    return (act_last_norm.astype(np.float64) @ sd["decoder.weight"].astype(np.float64).T + sd["decoder.bias"].astype(np.float64)).astype(np.float32)


@pytest.mark.parametrize("kind", ["ab", "nb"])
def test_micro_components_vs_reference_activations(hip, kind):
    """micro_{kind}_forward.npz carries act_aa_encoder, act_pos, act_att_out, act_last_norm recorded from the reference's modules."""
    z = load_golden(f"micro_{kind}_forward.npz")
    cfg, sd = load_cfg(kind), load_weights(kind)
    cls = hip.AntiTFNet if kind == "ab" else hip.NanoAntiTFNet
    B, d, n_att = z["tokens"].shape[0], int(cfg["d_model"]), int(cfg["cs_layers"])
    m = cls(**cfg); m.load_state_dict(sd)
    try:
        got = _stages(m, kind, z["tokens"], z["region"], chain_or_none(z), B, d, n_att)
    finally:
        m.close()
    _close("a3/a4 aa_encoder", got["aa_encoder"], z["act_aa_encoder"], 2e-5)
    _close("a5 pos", got["pos"], z["act_pos"], 2e-5)
    _close("a9/a10 self_at output", got[f"att{n_att - 1}"], z["act_att_out"], 2e-5)
    _close("a11 decoder on the reference's last_norm rows", got["logits"], _decoder_on(sd, z["act_last_norm"]), 2e-5)
    _close("a2 logits", got["logits"], z["logits"], 2e-5)


@pytest.mark.parametrize("kind", ["ab", "nb"])
def test_deep_components_vs_reference_activations(hip, kind):
    """deep_{kind}_acts.npz: production depth and heads at small width; every ByteNet stack, every SelfAttBlock, both attention
    residual sums of a block and last_norm against the reference's own module outputs."""
    z, cfg, sd = load_deep(kind)
    a = np.load(os.path.join(GOLDEN, f"deep_{kind}_acts.npz"))
    assert str(a["weight_sha256"]) == str(z["weight_sha256"])
    cls = hip.AntiTFNet if kind == "ab" else hip.NanoAntiTFNet
    B, d, n_att = z["tokens"].shape[0], int(cfg["d_model"]), int(cfg["cs_layers"])
    m = cls(**cfg); m.load_state_dict(sd)
    try:
        got = _stages(m, kind, z["tokens"], z["region"], chain_or_none(z), B, d, n_att)
    finally:
        m.close()
    _close("a3/a4 aa_encoder", got["aa_encoder"], a["act_aa_encoder"], 2e-5)
    _close("a5 pos", got["pos"], a["act_pos"], 2e-5)
    if kind == "ab":
        _close("a5 side", got["chn"], a["act_side"], 2e-5)
    _close("a7 conv stack", got["conv"], a["act_conv"], 2e-5)
    for n in range(n_att):
        x_in = a["act_conv"] if n == 0 else a[f"act_att{n - 1}"]
        _close(f"a8/a9 block {n}: x + AttLayer(x)", got[f"att{n}_at1"], x_in + a[f"act_att{n}_a1"], 2e-5)
        _close(f"a10 block {n} output", got[f"att{n}"], a[f"act_att{n}"], 2e-5)
    _close("a11 decoder on the reference's last_norm rows", got["logits"], _decoder_on(sd, a["act_last_norm"]), 2e-5)
    _close("a2 logits", got["logits"], z["logits"], 2e-5)


@pytest.mark.parametrize("kind,route", [("ab", "split"), ("ab", "f32_gemm"), ("ab", "f32_all"), ("nb", "split"), ("nb", "f32_all")])
def test_production_width_components_vs_oracle(hip, kind, route):
    """Production width, launches large enough for the big-tile / split-precision kernels, stage by stage against the oracle's
    PyTorch-CPU twin (float32 CPU arithmetic, the reference's own).  The bound per stage is 5e-5 of the tensor's largest magnitude
    (observed ~1e-6 .. 5e-6: two float32 evaluations of a 768-wide network)."""
    torch = pytest.importorskip("torch")
    import hudiff_oracle_torch as hot
    from hudiff_amd import evalsets as E
    from hudiff_amd import synthetic as S
    cfg = dict(S.AB_CONFIG if kind == "ab" else S.NB_CONFIG, dropout=0.0)
    sd = S.random_state_dict(kind, cfg, seed=3)
    B = 32 if kind == "ab" else 64
    b = E.eval_batch("huab348" if kind == "ab" else "vhh", B, row0=17)
    torch.set_num_threads(max(1, min(32, os.cpu_count() or 8)))
    net = hot.TorchOracleNet(kind, cfg, sd)
    net.trace = {}
    ref_logits = net.forward(b["tokens"], b["region"], b["chain"])
    want = net.trace
    cls = hip.AntiTFNet if kind == "ab" else hip.NanoAntiTFNet
    d, n_att = int(cfg["d_model"]), int(cfg["cs_layers"])
    m = cls(**cfg, precision=route); m.load_state_dict(sd)
    try:
        got = _stages(m, kind, b["tokens"], b["region"], b["chain"], B, d, n_att)
        info = m.precision_info()
    finally:
        m.close()
    assert info["precision"] == route and info["range_fallbacks"] == 0 and info["lnsync_fallbacks"] == 0, info
    assert info["split_in_use"] == (route != "f32_all"), info
    worst = {}
    for key in ["aa_encoder", "pos"] + (["chn"] if kind == "ab" else []) + ["conv"]:
        worst[key] = _close(key, got[key], want[key], 5e-5)
    for n in range(n_att):
        worst[f"att{n}_at1"] = _close(f"block {n}: x + A1(x)", got[f"att{n}_at1"], want[f"att{n}_at1"], 5e-5)
        worst[f"att{n}"] = _close(f"block {n} output", got[f"att{n}"], want[f"att{n}"], 5e-5)
    last = n_att - 1
    worst["at2"] = _close("last block: at + A2(LN1(at))", got[f"att{last}_at2"], want[f"att{last}_at2"], 5e-5)
    assert float(np.abs(got["logits"] - ref_logits).max()) < 1e-4
    print(kind, route, "worst relative error per stage:", {k: "%.1e" % v for k, v in worst.items()})
