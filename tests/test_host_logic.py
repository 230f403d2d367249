"""Host-side logic that needs no GPU: input preparation, checkpoint envelopes, CSV / FASTA writers, sharding."""
import io
import os
import re

import numpy as np
import pytest

from hudiff_amd import inputs as I
from hudiff_amd import tables as T
from hudiff_amd.cli import nanosample as nano_cli
from hudiff_amd.cli import sample as ab_cli
from hudiff_amd.sampler import Job, sample_jobs


def fake_numbering(seq, chain="H"):
    """Contiguous IMGT numbering 1..len with the 111/112 insertion block filled symmetrically -- enough to
    exercise the slot logic without ANARCI."""
    names = T.HEAVY_POSITIONS if chain == "H" else T.LIGHT_POSITIONS
    plain = [n for n in names if n.isdigit() and n != "10"]
    ins = [n for n in names if not n.isdigit()]
    extra = max(0, len(seq) - len(plain))
    lo = extra // 2 + extra % 2
    used = set(plain) | set(ins[:lo]) | set(ins[len(ins) - (extra - lo):] if extra - lo else [])
    keys = [n for n in names if n in used][: len(seq)]
    return dict(zip(keys, seq))


H_SEQ = "EVQLVESGGGLVQPGGSLRLSCAASGFTFSSYAMSWVRQAPGKGLEWVSAISGSGGSTYYADSVKGRFTISRDNSKNTLYLQMNSLRAEDTAVYYCAKDRLSITIRPRYYGLDVWGQGTTVTVSS"
L_SEQ = "DIQMTQSPSSLSASVGDRVTITCRASQSISSYLNWYQQKPGKAPKLLIYAASSLQSGVPSRFSGSGSGTDFTLTISSLQPEDFATYYCQQSYSTPLTFGGGTKVEIK"


def test_antibody_row_masks():
    h, l = fake_numbering(H_SEQ, "H"), fake_numbering(L_SEQ, "L")
    tok, reg, chain, loc = I.antibody_row(h, l, "K", finetune=True)
    assert tok.shape == (291,) and reg.tolist() == T.ab_region().tolist() and chain == (0, 2)
    table = np.array(T.HEAVY_CDR_KABAT_NO_VERNIER + T.LIGHT_CDR_KABAT_NO_VERNIER)
    slots = I.slot_residues(h, "H") + I.slot_residues(l, "L")
    for i in range(291):
        if table[i] == 0 and slots[i] != "-":
            assert tok[i] == 22 and i in loc
        else:
            assert tok[i] != 22 and i not in loc                   # CDRs and framework gaps are left alone
    assert len(loc) <= 157
    # pretrain mode masks every IMGT framework slot, gaps included (sample.py:148-151)
    tok2, _, _, loc2 = I.antibody_row(h, l, "L", finetune=False)
    assert len(loc2) == 93 + 92 == (np.array(T.HEAVY_CDR_INDEX + T.LIGHT_CDR_INDEX) == 0).sum()
    assert (tok2[loc2] == 22).all()
    # untokenize drops gaps and gives back the CDR residues untouched
    hs, ls = I.untokenize_antibody(np.where(tok == 22, 0, tok))
    assert len(hs) == len(H_SEQ) and len(ls) == len(L_SEQ)


def test_nanobody_row_masks():
    h = fake_numbering(H_SEQ, "H")
    for inpaint in (False, True):
        tok, reg, loc = I.nanobody_row(h, inpaint_sample=inpaint)
        table = np.array(T.INPAINT_HEAVY_CDR_INDEX if inpaint else T.HEAVY_CDR_INDEX)
        slots = I.slot_residues(h, "H")
        want = [i for i in range(152) if table[i] == 0 and slots[i] != "-"]
        assert loc.tolist() == want and (tok[loc] == 22).all() and (tok == 22).sum() == len(want)
        assert len(loc) <= (93 if not inpaint else 87)


def test_slot_residues_ignores_unknown_insertions(capsys):
    d = fake_numbering(H_SEQ, "H")
    d["111Z"] = "W"
    d["60A"] = "W"
    out = I.slot_residues(d, "H", quiet=False)
    assert len(out) == 152 and "W" not in [out[i] for i in range(152) if T.HEAVY_POSITIONS[i] in ("111Z", "60A")]
    msg = capsys.readouterr().out
    assert "CDR has problem" in msg


def test_slot_identity():
    a = np.array([0, 1, 21, 3, 21])
    assert I.slot_identity(a, a) == 1.0
    assert I.slot_identity(a, np.array([0, 2, 21, 3, 4])) == 2 / 4


class FakeModel:
    """CPU stand-in with the product model's sampling surface: fills visited slots with (global row + slot) % 20."""
    kind = "ab"
    max_len = 291

    def sample(self, tokens, region, chain, order, T, *, seed=0, row0=0, q_noise=None, dropout="faithful", **kw):
        out = np.array(tokens, copy=True)
        for b in range(out.shape[0]):
            for t in range(int(T[b])):
                s = order[b, t]
                out[b, s] = (row0 + b + s + seed) % 20
        return out


def _jobs(n):
    h, l = fake_numbering(H_SEQ, "H"), fake_numbering(L_SEQ, "L")
    jobs = []
    for i in range(n):
        tok, reg, chain, loc = I.antibody_row(h, l, "K")
        rng = np.random.default_rng(i)
        rng.shuffle(loc)
        jobs.append(Job(tokens=tok, region=reg, loc=loc, chain=chain, name=f"ab{i}"))
    return jobs


def test_sample_jobs_batches_rows_with_global_ids():
    jobs = _jobs(5)
    big = sample_jobs(FakeModel(), jobs, replicas=3, seed=7, device_batch=256)
    small = sample_jobs(FakeModel(), jobs, replicas=3, seed=7, device_batch=4)     # 15 rows in chunks of 4
    assert big.shape == (5, 1, 3, 291) and np.array_equal(big, small)
    assert not (big == 22).any()
    two = sample_jobs(FakeModel(), jobs, replicas=2, seed=7, passes=2)
    assert two.shape == (5, 2, 2, 291)


def _reference_retry(rows_per_sweep, want, tries, accept):
    """Literal walk of nanobody_scripts/nanosample.py:316-353 for one input over pre-drawn sweeps."""
    written, sweep = [], 0
    while want > 0 and tries > 0:
        for row in rows_per_sweep[sweep]:
            if want == 0:
                break
            if accept(row):
                written.append(row)
                want -= 1
            elif tries == 1:
                written.append(row)
            tries -= 1
        sweep += 1
    return written, sweep


def test_retry_loop_matches_reference_walk(monkeypatch):
    """sample_jobs_with_retry == the reference's accept / re-sweep loop, for several (want, tries, replicas):
    record every sweep the batched driver runs, then walk each input's own sweeps as the reference does."""
    from hudiff_amd import sampler
    jobs = _jobs(5)
    probe = int(jobs[0].loc[0])
    accept = lambda row: int(row[probe]) % 3 == 0                   # varies with global row id and sweep seed
    real = sampler.sample_jobs
    for want, tries, replicas in ((1, 10, 1), (2, 3, 2), (3, 4, 2), (1, 1, 3), (2, 5, 3), (0, 5, 2), (2, 0, 1)):
        sweeps = {j.name: [] for j in jobs}

        def recording(model, sub, *a, **kw):
            res = real(model, sub, *a, **kw)
            for i, j in enumerate(sub):
                sweeps[j.name].append(res[i, 0])
            return res
        monkeypatch.setattr(sampler, "sample_jobs", recording)
        got = sampler.sample_jobs_with_retry(FakeModel(), jobs, replicas, 7, want=want, tries=tries, accept=accept,
                                             device_batch=3)
        monkeypatch.setattr(sampler, "sample_jobs", real)
        n_accept = 0
        for i, j in enumerate(jobs):
            want_rows, used = _reference_retry(sweeps[j.name] + [None], want, tries, accept)
            assert used == len(sweeps[j.name]), (want, tries, replicas, j.name)      # no sweep too many or too few
            assert len(got[i]) == len(want_rows) and all(np.array_equal(x, y) for x, y in zip(got[i], want_rows))
            n_accept += sum(accept(r) for r in got[i])
        if want and tries:
            assert n_accept > 0 and any(len(v) > 1 for v in sweeps.values()) or tries == 1


def test_retry_rows_are_keyed_by_original_job_index():
    """A sequence's noise ids must not depend on which other inputs are still active: every device launch
    of a re-sweep keys its rows as original_job * replicas + replica."""
    from hudiff_amd import sampler
    jobs = _jobs(6)
    replicas = 2
    for i, j in enumerate(jobs):                       # tag every job's rows so the fake model can recognise them
        j.tokens = j.tokens.copy()
        j.tokens[int(jobs[0].loc[0])] = i % 20

    class Recording(FakeModel):
        calls = []

        def sample(self, tokens, region, chain, order, T, *, seed=0, row0=0, **kw):
            tag = tokens[:, int(jobs[0].loc[0])]
            tag = np.where(tag >= 100, tag - 100, tag)
            self.calls.append((seed, [int(row0) + b for b in range(tokens.shape[0])], [int(t) for t in tag]))
            out = super().sample(tokens, region, chain, order, T, seed=seed, row0=row0, **kw)
            out[:, int(jobs[0].loc[0])] = tag + 100            # keep the tag through the sweeps
            return out

    accept = lambda row: int(row[int(jobs[0].loc[0])]) - 100 in (0, 2, 5)       # jobs 1, 3, 4 keep failing
    Recording.calls = []
    sampler.sample_jobs_with_retry(Recording(), jobs, replicas, 7, want=1, tries=6, accept=accept, device_batch=3)
    assert len({c[0] for c in Recording.calls}) == 3                             # three sweeps
    for seed, gids, tags in Recording.calls:
        assert len(gids) <= 3
        for g, t in zip(gids, tags):
            assert g // replicas == t, (seed, gids, tags)                        # global id = original job * replicas + r
    later = [t for seed, g, tg in Recording.calls if seed != 7 for t in tg]
    assert sorted(set(later)) == [1, 3, 4]


def test_checkpoint_loader_refuses_foreign_pickles(tmp_path):
    torch = pytest.importorskip("torch")
    from hudiff_amd import checkpoint as ck

    class Evil:
        def __reduce__(self):
            return (os.path.join, ("a", "b"))           # any callable outside the allow-list
    p = tmp_path / "evil.pt"
    torch.save({"model": {}, "x": Evil()}, p)
    with pytest.raises(RuntimeError, match="allow-list"):
        ck.load_checkpoint(str(p))
    assert ck.load_checkpoint(str(p), trust_pickle=True)["x"] == os.path.join("a", "b")


def test_checkpoint_loader_refuses_nested_unpickle_gadgets(tmp_path):
    """torch.storage._load_from_bytes(b) is torch.load(BytesIO(b), weights_only=False), an unrestricted
    nested unpickle; torch.serialization.load likewise.  The allow-list is exact (module, name) pairs: neither resolves, and
    the inner payload never runs."""
    import io
    import pickle
    torch = pytest.importorskip("torch")
    from hudiff_amd import checkpoint as ck
    marker = tmp_path / "ran"

    class Inner:
        def __reduce__(self):
            return (open, (str(marker), "w"))            # harmless side effect that proves execution
    inner = io.BytesIO()
    torch.save(Inner(), inner)

    class ViaLoadFromBytes:
        def __reduce__(self):
            return (torch.storage._load_from_bytes, (inner.getvalue(),))

    class ViaSerializationLoad:
        def __reduce__(self):
            return (torch.serialization.load, (io.BytesIO(inner.getvalue()),))
    for n, gadget in enumerate((ViaLoadFromBytes(), ViaSerializationLoad())):
        p = tmp_path / f"gadget{n}.pt"
        try:
            torch.save({"model": {}, "x": gadget}, p)
        except (TypeError, pickle.PicklingError):        # BytesIO argument not picklable on this torch: the first gadget suffices
            continue
        with pytest.raises(RuntimeError, match="allow-list"):
            ck.load_checkpoint(str(p))
        assert not marker.exists()
    # whole modules are never admitted
    for module, name in (("torch.storage", "_load_from_bytes"), ("torch.serialization", "load"), ("torch._utils", "_rebuild_wrapper_subclass"),
                         ("builtins", "eval"), ("builtins", "getattr"), ("os", "system"), ("torch", "load"), ("numpy", "load")):
        assert not ck._allowed(module, name), (module, name)
    for module, name in (("torch._utils", "_rebuild_tensor_v2"), ("torch", "FloatStorage"), ("torch", "ComplexFloatStorage"),
                         ("torch", "float32"), ("collections", "OrderedDict"), ("easydict", "EasyDict")):
        assert ck._allowed(module, name), (module, name)


def test_checkpoint_envelopes(tmp_path):
    torch = pytest.importorskip("torch")
    from hudiff_amd import checkpoint as ck
    from hudiff_amd import synthetic as S
    cfg = ck.EasyDict({"name": "trans_oadm", "model": dict(S.AB_CONFIG)})
    sd = {"module.decoder.bias": torch.zeros(23), "self_at.layers.0.attn_hl.rope": torch.zeros(291, 32, dtype=torch.complex64)}
    p = tmp_path / "ab.pt"
    torch.save({"fineconfig": cfg, "pretrain_config": cfg, "model": sd, "iteration": 3}, p)
    ckpt = ck.load_checkpoint(str(p))
    config, state, finetune = ck.antibody_model_from_checkpoint(ckpt, "finetune")
    assert finetune and config.model.max_len == 291 and config["name"] == "trans_oadm"
    assert set(state) == {"decoder.bias", "self_at.layers.0.attn_hl.rope"}         # 'module.' stripped
    with pytest.raises(KeyError):
        ck.antibody_model_from_checkpoint(ckpt, "pretrain")
    # nanobody fine-tune envelope: prefixed framework state, only infilling_pretrain.* is used
    nb = {"config": ck.EasyDict({"name": "infilling", "model": {}}), "infilling_params": ck.EasyDict(dict(S.NB_CONFIG)),
          "abnativ_params": {}, "model": {"eval_abnativ_model.x": torch.zeros(1), "infilling_pretrain.decoder.bias": torch.ones(23),
                                          "target_infilling_pretrain.decoder.bias": torch.zeros(23)}}
    p2 = tmp_path / "nb.pt"
    torch.save(nb, p2)
    name, params, state = ck.nanobody_model_from_checkpoint(ck.load_checkpoint(str(p2)), "finetune_vh")
    assert name == "nano" and params["max_len"] == 152 and list(state) == ["decoder.bias"]
    assert float(state["decoder.bias"][0]) == 1.0


def test_cli_flags_and_tags():
    a = ab_cli.build_parser().parse_args([])
    assert (a.ckpt, a.ckpt_version, a.batch_size, a.sample_number, a.try_number, a.seed, a.sample_order, a.sample_method,
            a.similarity_search, a.fa_version) == ("checkpoints/antibody/hudiffab.pt", "finetune", 1, 1, 1, 2023, "shuffle",
                                                   "FR", True, "v007")
    assert ab_cli.sample_tag(a) == "2023_shuffle_lab_finetune_search_simi_True"
    a2 = ab_cli.build_parser().parse_args(["--data_fpath", "x/humab25/parental_mouse.csv", "--similarity_search", ""])
    assert a2.similarity_search is False and ab_cli.sample_tag(a2).startswith("2023_shuffle_humab_finetune_search_simi_False")
    n = nano_cli.build_parser().parse_args(["--inpaint_sample", "True", "--model", "pretrain"])
    assert n.inpaint_sample is True and n.try_number == 10 and n.fa_version == "v_nano"
    assert nano_cli.sample_tag(n) == "2023_shuffle_nanobert_gen_not_equal_pretrain"


def test_traditional_method_layout(tmp_path, monkeypatch):
    """--traditional_method (sample.py:539-576): grafting itself is abnumber's; the CSV / FASTA / log-dir layout is ours."""
    csv = tmp_path / "humab_x.csv"
    csv.write_text("type,name,h_seq,l_seq\nmouse,a1," + H_SEQ + "," + L_SEQ + "\nhuman,a1,AAA,CCC\nmouse,b2," + H_SEQ[1:] + "," + L_SEQ + "\n")
    with pytest.raises(RuntimeError, match="abnumber"):
        ab_cli.main(["--data_fpath", str(csv), "--traditional_method", "1"])
    calls = []
    monkeypatch.setattr(I, "cdr_pair_grafting", lambda h, l, back_mutation=False, scheme="kabat": (calls.append(back_mutation), ("G" + h, "G" + l))[1])
    out = ab_cli.main(["--data_fpath", str(csv), "--traditional_method", "1"])
    assert os.path.dirname(os.path.dirname(out)) == str(tmp_path)
    assert re.match(r"humab_cdr_graft_back_mutation_True_\d{4}_", os.path.basename(os.path.dirname(out))) and calls == [True, True]
    assert open(out).read().splitlines() == ["Specific,name,hseq,lseq,", f"humanization,a1human_sample,G{H_SEQ},G{L_SEQ}",
                                             f"humanization,b2human_sample,G{H_SEQ[1:]},G{L_SEQ}"]
    fa = open(os.path.join(os.path.dirname(out), "sample_identity.fa")).read().splitlines()
    assert fa[0] == ">v007human0 VH" and fa[1] == "G" + H_SEQ and len(fa) == 8


def test_evalset_fixture_rows():
    """hudiff_amd/data/real_rows.npz (scripts/make_real_rows.py): the reference's evaluation rows as slot tokens."""
    from hudiff_amd import evalsets as E
    z = E.load_rows()
    assert z["huab348_tokens"].shape == (348, 291) and z["humab25_tokens"].shape == (25, 291) and z["vhh_tokens"].shape == (300, 152)
    assert set(np.unique(z["huab348_lchain"])) <= {1, 2} and ((z["huab348_tokens"] >= 0) & (z["huab348_tokens"] <= 21)).all()
    b = E.eval_batch("huab348", 400, row0=0)
    assert (b["T"].min(), b["T"].max()) == (141, 154) and abs(float(b["T"][:348].mean()) - 152.86) < 0.01
    assert np.array_equal(b["truth"][348:400], b["truth"][:52]) and not np.array_equal(b["order"][348], b["order"][0])
    for r in (0, 17, 399):                                   # masks follow the reference's finetune rule (tests/test_input_prep_golden.py)
        maskable = np.array(T.HEAVY_CDR_KABAT_NO_VERNIER + T.LIGHT_CDR_KABAT_NO_VERNIER) == 0
        want = maskable & (b["truth"][r] != 21)
        assert np.array_equal(b["tokens"][r] == 22, want) and sorted(b["order"][r, :b["T"][r]]) == np.nonzero(want)[0].tolist()
    # sharding: rows of a later block are the same rows (global ids key everything)
    c = E.eval_batch("huab348", 16, row0=384)
    assert np.array_equal(c["tokens"], b["tokens"][384:400]) and np.array_equal(c["order"][:, :c["T"].max()], b["order"][384:400, :c["T"].max()])
    v = E.eval_batch("vhh", 300, mode="inpaint")
    assert v["T"].max() <= 87 and v["chain"] is None
    h, l = E.sequences("humab25")[0]
    assert h.startswith("EVKLQQSGPGLV") and l.endswith("GGGTKLEIK")


def test_fasta_writers(tmp_path):
    from hudiff_amd.cli.common import write_fasta_2line, write_fasta_wrapped
    p = tmp_path / "a.fa"
    write_fasta_2line([("v007human0", "VH", "EVQ"), ("v007human0", "VL", "DIQ")], p)
    assert p.read_text() == ">v007human0 VH\nEVQ\n>v007human0 VL\nDIQ\n"
    write_fasta_wrapped([("VHv_nano_0", "<unknown description>", "A" * 70)], p)
    assert p.read_text() == ">VHv_nano_0 <unknown description>\n" + "A" * 60 + "\n" + "A" * 10 + "\n"


def test_fasta_reader_and_single_molecule_helpers(tmp_path):
    """PDB-style FASTA -> chains (sample_for_anti_cdr.py:53-70, sample_for_nano_cdr.py:30-44); flanks are dropped by the
    domain cut (``Chain(seq).seq``); per-sample FASTA files of --structure."""
    from hudiff_amd.cli import common, sample_for_anti_cdr as ab, sample_for_nano_cdr as nb
    from hudiff_amd import numbering as N
    from test_numbering import KNOWN
    vh, vl, vhh = KNOWN["trastuzumab_VH"][0], KNOWN["trastuzumab_VK"][0], KNOWN["caplacizumab_VHH"][0]
    fa = tmp_path / "1abc.fasta"
    fa.write_text(">1ABC_1|Chain A|Spike protein S1|virus\nTNLCPFGEVFNATRF\nASVYAWN\n"
                  f">1ABC_2|Chain B[auth H]|2B04 heavy chain|Mus musculus (10090)\n{vh[:60]}\n{vh[60:]}ASTKGPSVFPLAP\n"
                  f">1ABC_3|Chain C[auth L]|2B04 light chain|Mus musculus (10090)\n{vl}RTVAAPSVFIFPPS\n"
                  f">1ABC_4|Chain D|Nanobody 3-2A2-4|Vicugna pacos (30538)\n{vhh}HHHHHH\n")
    recs = common.read_fasta(str(fa))
    assert len(recs) == 4 and recs[0][1] == "TNLCPFGEVFNATRFASVYAWN" and recs[1][0].startswith("1ABC_2|Chain B")
    h, l = ab.get_h_l_seq_from_fasta(str(fa))
    assert h == vh + "ASTKGPSVFPLAP" and l == vl + "RTVAAPSVFIFPPS"
    assert N.domain_sequence(h) == vh and N.domain_sequence(l) == vl            # constant-region starts are cut off
    nano = nb.get_nano_seq_from_fasta(str(fa))
    assert nano == vhh + "HHHHHH" and N.domain_sequence(nano) == vhh
    a, n = ab.build_parser().parse_args([]), nb.build_parser().parse_args([])
    assert (a.batch_size, a.sample_number, a.seed, a.log_dirpath, a.anti_complex_fasta) == (10, 10, 42, "antibody_sample_log/", "fasta_file/7k9i.fasta")
    assert (n.batch_size, n.sample_number, n.seed, n.inpaint_sample, n.model, n.fa_version) == (10, 100, 42, True, "finetune_vh", "v_nano")
    csv = tmp_path / "log" / "sample_humanization_result.csv"
    csv.parent.mkdir()
    csv.write_text("Specific,name,hseq,\n")
    d = common.split_fasta_for_save(str(csv), [vhh, (vh, vl)])
    assert open(os.path.join(d, "0_human.fasta")).read().splitlines()[0] == ">0_human_H <unknown description>"
    assert open(os.path.join(d, "1_human.fasta")).read().splitlines() == [">1_human_H VH", vh, ">1_human_L VL", vl]
    assert os.path.isdir(tmp_path / "log" / "sample_human_pdb")


def test_bench_clock_power_sampler_is_optional():
    """bench.py samples shader clock / package power from the amdgpu hwmon files of the GPU it runs on; where there is no
    such device (this container) the sampler must be inert: no thread, no exception, `None` in the JSON line."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    s = mod.ClockPowerSampler(0)
    if s.dir is None:                       # no GPU / no hwmon: inert
        assert s.start() is s and s.stop() is None
    else:                                   # on a GPU box: a short window yields a dict with the documented keys
        import time
        s.start(); time.sleep(0.35)
        out = s.stop()
        assert out is None or {"sclk_mhz_median", "power_w_median", "samples"} <= set(out)
    phys, logical = mod.physical_cores()
    assert 1 <= phys <= logical
