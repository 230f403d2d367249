"""Complete sampling traces of the REFERENCE at production width (tests/golden/prod_{ab,nb}_trace{,_b}.npz, written by
oracle/make_golden_prod_trace.py from the reference's own AntiTFNet / NanoAntiTFNet on two real rows, dropout 0, recorded
torch.multinomial noise) replayed through the CPU oracle, the fp32 HIP kernels and the split-precision HIP kernels.
Integer work: the final tokens and every per-step draw must match bit for bit."""
import hashlib
import os

import numpy as np
import pytest

from conftest import GOLDEN


# (kind, fixture suffix, masking mode of the filler rows): the "_b" set has other weights and rows, and samples the nanobody model
# without a mask (BASELINE configs[3]: --inpaint_sample False)
SETS = [("ab", "", "finetune"), ("nb", "", "inpaint"), ("ab", "_b", "finetune"), ("nb", "_b", "plain")]
IDS = [k + s for k, s, _ in SETS]


def _load(kind, suffix=""):
    from hudiff_amd import synthetic as S
    z = np.load(os.path.join(GOLDEN, f"prod_{kind}_trace{suffix}.npz"))
    cfg = dict(S.AB_CONFIG if kind == "ab" else S.NB_CONFIG, dropout=0.0)
    sd = S.random_state_dict(kind, cfg, seed=int(z["weight_seed"]))
    h = hashlib.sha256()
    for k in sorted(sd):
        h.update(k.encode())
        h.update(np.ascontiguousarray(sd[k], dtype=np.float32).tobytes())
    assert h.hexdigest() == str(z["weight_sha256"]), "regenerated weights differ from the ones the reference ran with"
    return z, cfg, sd


@pytest.mark.parametrize("kind,suffix,mode", SETS, ids=IDS)
def test_oracle_reproduces_the_reference_trace(kind, suffix, mode):
    torch = pytest.importorskip("torch")
    import hudiff_oracle as ho
    import hudiff_oracle_torch as hot
    z, cfg, sd = _load(kind, suffix)
    torch.set_num_threads(8)
    net = hot.TorchOracleNet(kind, cfg, sd)
    chain = z["chain"] if z["chain"].size else None
    trace = []
    final = ho.sample(net, z["tokens"], z["region"], chain, z["order"], z["T"], q_noise=z["q"], trace=trace)
    assert np.array_equal(final, z["final"])
    assert (z["T"] >= 80).all() and not (final == 22).any()


@pytest.mark.gpu
@pytest.mark.parametrize("kind,suffix,mode", SETS, ids=IDS)
def test_hip_paths_reproduce_the_reference_trace(kind, suffix, mode):
    import hudiff_amd
    from hudiff_amd import evalsets as E
    if hudiff_amd.device_count() < 1:
        pytest.fail("no MI355X visible: GPU tests must run on the GPU box (there is no CPU fallback)")
    z, cfg, sd = _load(kind, suffix)
    cls = hudiff_amd.AntiTFNet if kind == "ab" else hudiff_amd.NanoAntiTFNet
    chain = z["chain"] if z["chain"].size else None
    B, Tmax = z["tokens"].shape[0], z["order"].shape[1]
    for x3 in ("f32_all", "f32_gemm", "split"):               # the three precision routes, chosen through the interface
        m = cls(**cfg, precision=x3); m.load_state_dict(sd)
        try:
            # the reference's two rows alone (small-launch fp32 kernels in both modes) ...
            out = m.sample(z["tokens"], z["region"], chain, z["order"], z["T"], q_noise=z["q"])
            assert np.array_equal(out, z["final"]), ("rows alone", x3)
            # ... and inside a launch large enough for the 128-row-tile kernels (fp32 big kernels / split-precision
            # kernels): filler rows are other real rows with arbitrary noise; rows are independent
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
            for lanes in (1, 2):
                out = m.sample(tok, reg, ch, order, T, q_noise=q, lanes=lanes)
                assert np.array_equal(out[:B], z["final"]), ("in a big launch", x3, lanes)
        finally:
            m.close()

