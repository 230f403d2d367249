"""The row-owner ByteNet chain kernel (hudiff_amd/csrc/hd_chain.hip.h; option bn_chain, off by default) against the three gemm_x3_k
launches per block it replaces and against the reference's production-width traces.  Reference: ByteNetBlock inside DualConv / NanoConv /
ByteNetTime (model/encoder/model.py:118-180, 249-304; sequence_models restated in oracle/ref_import.py:124-147)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu
CHAIN = {"bn_chain": 3, "bn_chain_min_tiles": 1}


@pytest.fixture(scope="module")
def hip():
    import hudiff_amd
    if hudiff_amd.device_count() < 1:
        pytest.fail("no MI355X visible: GPU tests must run on the GPU box (there is no CPU fallback)")
    return hudiff_amd


def _models(hip, kind, dropout=None):
    from hudiff_amd import synthetic as S
    cfg = dict(S.AB_CONFIG if kind == "ab" else S.NB_CONFIG)
    if dropout is not None:
        cfg["dropout"] = dropout
    sd = S.random_state_dict(kind, cfg, seed=2)
    cls = hip.AntiTFNet if kind == "ab" else hip.NanoAntiTFNet
    old = cls(**cfg, precision="split", options={"bn_chain": 0}); old.load_state_dict(sd)
    new = cls(**cfg, precision="split", options=CHAIN); new.load_state_dict(sd)
    return cfg, old, new


@pytest.mark.parametrize("kind,B", [("ab", 48), ("nb", 96)])
def test_chain_kernel_matches_the_per_gemm_launches(hip, kind, B):
    """Token encoder, Dual / NanoConv stack and the X16 copy handed to the attention: stage by stage and at the logits; generated
    dropout (the same keep decisions: the hash is keyed by (row, slot, column)); tokens of a short sample."""
    from hudiff_amd import evalsets as E
    cfg, old, new = _models(hip, kind)
    try:
        b = E.eval_batch("huab348" if kind == "ab" else "vhh", B, row0=9)
        args = (b["tokens"], b["region"], b["chain"])
        for stage, name in ((1, "FEAT"), (2, "Y"), (2, "YX")):
            got = []
            for m in (old, new):
                m.debug_stop_after(stage); m(*args, dropout="off"); got.append(m.debug_read(name, B)); m.debug_stop_after(0)
            assert np.isfinite(got[1]).all() and np.abs(got[0] - got[1]).max() <= 2e-5 * max(1.0, np.abs(got[0]).max()), (stage, name)
        a, c = old(*args, dropout="off"), new(*args, dropout="off")
        assert 0.0 < float(np.abs(a - c).max()) < 1e-5                       # the other kernels really ran, and agree
        a, c = old(*args, dropout="faithful", seed=4, row0=7, step=5), new(*args, dropout="faithful", seed=4, row0=7, step=5)
        assert float(np.abs(a - c).max()) < 3e-4                             # (p = 0.5 dropout at 12 sites amplifies last-ulp differences: NOTES.md B)
        rng = np.random.default_rng(1)
        L, d, D = cfg["max_len"], cfg["d_model"], cfg["sum_d_model"]
        em = (rng.random((cfg["n_encoder_layers"], B, L, d)) >= cfg["dropout"]).astype(np.uint8)
        cm = (rng.random((cfg["dual_layers"], B, L, D)) >= 0.5).astype(np.uint8)
        # (injected keep-masks keep the gemm_x3_k launches on both handles -- the chain kernel is instantiated for generated dropout only: the same bits)
        a, c = old(*args, dropout="inject", enc_masks=em, conv_masks=cm), new(*args, dropout="inject", enc_masks=em, conv_masks=cm)
        assert np.array_equal(a, c)
        T6 = np.minimum(b["T"], 6)
        assert np.array_equal(old.sample(*args, b["order"], T6, seed=3, row0=0), new.sample(*args, b["order"], T6, seed=3, row0=0))
        info = new.precision_info()
        assert info["split_in_use"] and info["range_fallbacks"] == 0 and info["lnsync_fallbacks"] == 0, info
    finally:
        old.close(); new.close()


@pytest.mark.parametrize("kind,suffix,mode", [("ab", "", "finetune"), ("nb", "_b", "plain")])
def test_chain_kernel_reproduces_the_reference_trace(hip, kind, suffix, mode):
    """Complete production-width samples of the REFERENCE (tests/golden/prod_*_trace*.npz) inside a launch large enough for the chain
    kernel: final tokens bit for bit."""
    from hudiff_amd import evalsets as E
    from test_prod_trace import _load
    z, cfg, sd = _load(kind, suffix)
    cls = hip.AntiTFNet if kind == "ab" else hip.NanoAntiTFNet
    chain = z["chain"] if z["chain"].size else None
    B, Tmax = z["tokens"].shape[0], z["order"].shape[1]
    n = 40 if kind == "ab" else 72
    fill = E.eval_batch("huab348" if kind == "ab" else "vhh", n - B, row0=100, mode=mode)
    Tm = max(Tmax, fill["order"].shape[1])
    tok = np.concatenate([z["tokens"], fill["tokens"]]); reg = np.concatenate([z["region"], fill["region"]])
    order = np.zeros((n, Tm), np.int64)
    order[:B, :Tmax] = z["order"]; order[B:, :fill["order"].shape[1]] = fill["order"]
    T = np.concatenate([z["T"], fill["T"]])
    q = np.ones((Tm, n, 22), np.float32)
    q[:Tmax, :B] = z["q"]
    q[:, B:] = np.random.default_rng(3).exponential(size=(Tm, n - B, 22)).astype(np.float32)
    ch = None if chain is None else np.concatenate([chain[:B], fill["chain"][:n - B], chain[B:], fill["chain"][n - B:]])
    m = cls(**cfg, precision="split", options=CHAIN); m.load_state_dict(sd)
    try:
        for lanes in (1, 2):
            out = m.sample(tok, reg, ch, order, T, q_noise=q, lanes=lanes)
            assert np.array_equal(out[:B], z["final"]), lanes
    finally:
        m.close()


def test_chain_option_is_off_by_default_and_fixed_at_finalize(hip):
    from hudiff_amd import synthetic as S
    from hudiff_amd._lib import HudiffError
    cfg = dict(S.NB_CONFIG)
    m = hip.NanoAntiTFNet(**cfg)
    try:
        assert m.get_option("bn_chain") == int(os.environ.get("HUDIFF_BN_CHAIN", "0"))
        m.load_state_dict(S.random_state_dict("nb", cfg, seed=0))
        with pytest.raises(HudiffError):
            m.set_option("bn_chain", 1 - min(m.get_option("bn_chain"), 1))      # the k-permuted weight images are built at hd_finalize
    finally:
        m.close()
