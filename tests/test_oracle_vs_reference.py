"""Oracle vs the REFERENCE's own classes at PRODUCTION width (configs/antibody_train.yml, heavy_train.yml), live.

Runs only where /root/reference exists (the build container); skipped on the GPU box.  This is the committed form
of the check DESIGN.md §3 quotes: with the same seeded weights loaded into the reference's ``AntiTFNet`` /
``NanoAntiTFNet`` (through oracle/ref_import.py) and into the numpy oracle, the logits agree to float32 round-off
(tolerance 2e-5; observed 2e-6 .. 4e-6), with dropout off and under the reference's own recorded dropout masks.
"""
import os

import numpy as np
import pytest

import hudiff_oracle as ho

pytestmark = pytest.mark.skipif(not os.path.isdir("/root/reference/model"), reason="/root/reference is not present on this machine")

TOL = 2e-5


@pytest.mark.parametrize("kind", ["ab", "nb"])
def test_production_width_oracle_matches_reference(kind):
    import torch
    import make_golden as mg
    import make_golden_deep as deep
    from hudiff_amd import synthetic as S
    cfg = dict(S.AB_CONFIG if kind == "ab" else S.NB_CONFIG)
    sd = S.random_state_dict(kind, cfg, seed=5)
    B = 2
    batch = S.synthetic_batch(kind, B, seed=17)
    tokens = batch["tokens"].astype(np.int64)
    tokens[1] = np.where(tokens[1] == 22, batch["truth"][1], tokens[1])
    region = batch["region"].astype(np.int64)
    chain = None if batch["chain"] is None else batch["chain"].astype(np.int64)
    torch.set_num_threads(8)
    ref0 = deep.build(kind, dict(cfg, dropout=0.0), sd)
    want = mg.ref_forward(ref0, tokens, region, chain)
    got = ho.OracleNet(kind, dict(cfg, dropout=0.0), sd)(tokens, region, chain)
    err = float(np.abs(got - want).max())
    assert err < TOL, err
    # dropout ON (the reference's inference behaviour): replay the masks the reference drew
    ref1 = deep.build(kind, cfg, sd)
    torch.manual_seed(7)
    with mg.Recorder() as rec:
        want_d = mg.ref_forward(ref1, tokens, region, chain)
    enc, conv = mg.canonical_masks(kind, rec.masks, cfg, B)
    got_d = ho.OracleNet(kind, cfg, sd)(tokens, region, chain, dropout=ho.Dropout("inject", enc_masks=enc, conv_masks=conv))
    err_d = float(np.abs(got_d - want_d).max())
    assert err_d < 2 * TOL, err_d
    assert np.abs(want_d - want).max() > 1e-2
    print(f"{kind}: max|dlogit| oracle vs reference = {err:.2e} (dropout off), {err_d:.2e} (reference's masks)")
