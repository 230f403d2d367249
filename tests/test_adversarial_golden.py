"""The oracle against the reference-generated "adversarial statistics" vectors (oracle/make_golden_adversarial.py):
production architecture, weights bent so that the residual stream has row mean >> row std / massive channels / |x| beyond
the fp16 range / |x| << 2^-3.  CPU only; the HIP paths are held to the same vectors in tests/test_gpu_adversarial.py."""
import hashlib
import os

import numpy as np
import pytest

import hudiff_oracle as ho
from conftest import GOLDEN

KINDS = ("ab", "nb")
VARIANTS = ("dc", "massive", "huge", "tiny", "plain")
LOGIT_TOL = 1e-4


def load_adv(kind, variant):
    """-> (fixture, config (dropout 0), state_dict regenerated from the recorded seed and checked against its SHA-256)."""
    from hudiff_amd import synthetic as S
    z = np.load(os.path.join(GOLDEN, f"adv_{kind}_{variant}.npz"))
    cfg = dict(S.AB_CONFIG if kind == "ab" else S.NB_CONFIG, dropout=0.0)
    sd = S.adversarial_state_dict(kind, cfg, int(z["weight_seed"]), variant)
    h = hashlib.sha256()
    for k in sorted(sd):
        h.update(k.encode())
        h.update(np.ascontiguousarray(sd[k], dtype=np.float32).tobytes())
    assert h.hexdigest() == str(z["weight_sha256"]), "regenerated weights differ from the ones the reference ran with"
    return z, cfg, sd


@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("kind", KINDS)
def test_fixture_statistics_are_what_they_claim(kind, variant):
    z = np.load(os.path.join(GOLDEN, f"adv_{kind}_{variant}.npz"))
    st = dict(zip([str(n) for n in z["stat_names"]], z["stats"]))          # name -> (max |mean| / std, max |x|, min std)
    assert float(z["reference_f32_vs_f64"]) < 5e-5                        # the function is well conditioned: 1e-4 means something
    if variant == "dc":
        assert st["norm1_in"][0] > 50 and st["norm2_in"][0] > 50 and st["attn1_in"][0] > 50
    elif variant == "massive":
        assert st["norm2_in"][1] > 1e4 and st["norm2_in"][1] < 32768
    elif variant == "huge":
        assert min(st[k][1] for k in st) > 65504
    elif variant == "tiny":
        assert max(st[k][1] for k in st) < 2.0 ** -8 and max(st[k][2] for k in st) < 2.0 ** -10
    else:                                                                 # plain: the control -- ordinary statistics
        assert max(st[k][1] for k in st) < 100 and max(st[k][0] for k in st) < 10


@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("kind", KINDS)
def test_oracle_reproduces_reference(kind, variant):
    z, cfg, sd = load_adv(kind, variant)
    chain = z["chain"] if z["chain"].size else None
    net = ho.OracleNet(kind, cfg, sd)
    logits = net(z["tokens"], z["region"], chain)
    assert np.abs(logits - z["logits"]).max() < LOGIT_TOL
    assert np.abs(logits - z["logits_f64"]).max() < LOGIT_TOL
    B, T = z["tokens"].shape[0], z["order"].shape[1]
    final = ho.sample(net, z["tokens"], z["region"], chain, z["order"], np.full(B, T), q_noise=z["q"])
    assert np.array_equal(final, z["final"])
