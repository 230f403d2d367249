"""End-to-end drop-in samplers on the GPU: synthetic checkpoint in the reference's envelope + pre-numbered input
-> CSV / FASTA with the reference's exact layout (antibody_scripts/sample.py:468-538, nanosample.py:294-353)."""
import json
import os
import re

import numpy as np
import pytest

from conftest import load_cfg, load_weights

pytestmark = pytest.mark.gpu


def _write_inputs(tmp_path, kind, n):
    from test_host_logic import H_SEQ, L_SEQ, fake_numbering
    rng = np.random.default_rng(0)
    aa = "ACDEFGHIKLMNPQRSTVWY"
    rows, numbered = [], []
    for i in range(n):
        h = "".join(rng.choice(list(aa)) if rng.random() < 0.1 else c for c in H_SEQ)
        l = "".join(rng.choice(list(aa)) if rng.random() < 0.1 else c for c in L_SEQ)
        if kind == "ab":
            rows.append(("mouse", f"m{i}", h, l))
            rows.append(("human", f"m{i}", h, l))          # non-mouse rows are ignored (get_mouse_line)
            numbered.append({"name": f"m{i}", "h": fake_numbering(h, "H"), "l": fake_numbering(l, "L"), "l_chain": "K"})
        else:
            rows.append((h,))
            numbered.append({"h": fake_numbering(h, "H")})
    csv = tmp_path / ("pairs.csv" if kind == "ab" else "vhh_filter.csv")
    with open(csv, "w") as f:
        f.write("type,name,h_seq,l_seq\n" if kind == "ab" else "vhhseq\n")
        for r in rows:
            f.write(",".join(r) + "\n")
    nb = tmp_path / "numbered.jsonl"
    with open(nb, "w") as f:
        for d in numbered:
            f.write(json.dumps(d) + "\n")
    return csv, nb


def test_antibody_cli_end_to_end(tmp_path):
    import torch
    from hudiff_amd import checkpoint as ck
    from hudiff_amd.cli import sample as cli
    cfg = dict(load_cfg("ab"), dropout=0.2)
    sd = {k: torch.from_numpy(v) for k, v in load_weights("ab").items()}
    ckdir = tmp_path / "run" / "checkpoints"
    ckdir.mkdir(parents=True)
    torch.save({"fineconfig": ck.EasyDict({}), "pretrain_config": ck.EasyDict({"name": "trans_oadm", "model": cfg}), "model": sd},
               ckdir / "hudiffab.pt")
    csv, nb = _write_inputs(tmp_path, "ab", 5)
    out = cli.main(["--ckpt", str(ckdir / "hudiffab.pt"), "--data_fpath", str(csv), "--numbered_fpath", str(nb),
                    "--batch_size", "3", "--seed", "5"])
    log_dir = os.path.dirname(out)
    assert os.path.dirname(log_dir) == str(tmp_path / "run")
    assert re.match(r"5_shuffle_lab_finetune_search_simi_True_\d{4}_\d\d_\d\d__\d\d_\d\d_\d\d$", os.path.basename(log_dir))
    lines = open(out).read().splitlines()
    assert lines[0] == "Specific,name,hseq,lseq,"
    assert len(lines) == 1 + 2 * 5
    for i in range(5):
        m, h = lines[1 + 2 * i].split(","), lines[2 + 2 * i].split(",")
        assert m[0] == "mouse" and m[1] == f"m{i}" and h[0] == "humanization" and h[1] == f"m{i}human_sample"
        assert set(h[2]) <= set("ACDEFGHIKLMNPQRSTVWYX") and set(h[3]) <= set("ACDEFGHIKLMNPQRSTVWYX")
        assert "<msk>" not in h[2] + h[3]
    fa = open(os.path.join(log_dir, "sample_identity.fa")).read().splitlines()
    assert fa[0] == ">v007human0 VH" and fa[2] == ">v007human0 VL" and len(fa) == 4 * 5
    # same seed -> same CSV body (noise is counter-based, independent of device batching)
    # (second run under another checkpoint parent: the log-dir name has one-second resolution, sample.py:434-443)
    ckdir2 = tmp_path / "run2" / "checkpoints"
    ckdir2.mkdir(parents=True)
    (ckdir2 / "hudiffab.pt").write_bytes((ckdir / "hudiffab.pt").read_bytes())
    out2 = cli.main(["--ckpt", str(ckdir2 / "hudiffab.pt"), "--data_fpath", str(csv), "--numbered_fpath", str(nb),
                     "--batch_size", "3", "--seed", "5", "--device_batch", "4"])
    assert os.path.dirname(out2) != log_dir
    assert open(out2).read() == open(out).read()
    # --precision picks the library's route (micro shapes run the fp32 small-launch kernels on every route: the same bytes)
    ckdir3 = tmp_path / "run3" / "checkpoints"
    ckdir3.mkdir(parents=True)
    (ckdir3 / "hudiffab.pt").write_bytes((ckdir / "hudiffab.pt").read_bytes())
    out3 = cli.main(["--ckpt", str(ckdir3 / "hudiffab.pt"), "--data_fpath", str(csv), "--numbered_fpath", str(nb),
                     "--batch_size", "3", "--seed", "5", "--precision", "f32_all"])
    assert open(out3).read() == open(out).read()


def test_nanobody_cli_end_to_end(tmp_path):
    import torch
    from hudiff_amd import checkpoint as ck
    from hudiff_amd.cli import nanosample as cli
    cfg = dict(load_cfg("nb"), dropout=0.5)
    sd = {"infilling_pretrain." + k: torch.from_numpy(v) for k, v in load_weights("nb").items()}
    sd["eval_abnativ_model.dummy"] = torch.zeros(3)
    ckdir = tmp_path / "run" / "checkpoints"
    ckdir.mkdir(parents=True)
    torch.save({"config": ck.EasyDict({"name": "infilling", "model": {}}), "infilling_params": ck.EasyDict(cfg),
                "abnativ_params": {}, "model": sd}, ckdir / "hudiffnb.pt")
    csv, nb = _write_inputs(tmp_path, "nb", 4)
    out = cli.main(["--ckpt", str(ckdir / "hudiffnb.pt"), "--data_fpath", str(csv), "--numbered_fpath", str(nb),
                    "--model", "finetune_vh", "--inpaint_sample", "True"])
    lines = open(out).read().splitlines()
    assert lines[0] == "Specific,name,hseq," and len(lines) == 1 + 2 * 4
    assert "abnativ_select_gen_not_equal_finetune_vh" in os.path.basename(os.path.dirname(out))
    for i in range(4):
        assert lines[1 + 2 * i].startswith(f"nano,{i},") and lines[2 + 2 * i].startswith(f"humanization,{i}human_sample,")
    fa = open(os.path.join(os.path.dirname(out), "sample_identity.fa")).read().splitlines()
    assert fa[0] == ">VHv_nano_0 <unknown description>"


def test_antibody_cli_from_raw_sequences(tmp_path):
    """No pre-numbered input: raw VH / VL strings go through the built-in IMGT slotter.  The
    finetune mask samples framework slots only, so every Kabat CDR (hence every shorter CDR-IMGT core) must come
    out unchanged, in order."""
    import torch
    from hudiff_amd import checkpoint as ck
    from hudiff_amd.cli import sample as cli
    from test_numbering import KNOWN
    cfg = dict(load_cfg("ab"), dropout=0.2)
    sd = {k: torch.from_numpy(v) for k, v in load_weights("ab").items()}
    ckdir = tmp_path / "run" / "checkpoints"
    ckdir.mkdir(parents=True)
    torch.save({"fineconfig": ck.EasyDict({}), "pretrain_config": ck.EasyDict({"name": "trans_oadm", "model": cfg}), "model": sd},
               ckdir / "hudiffab.pt")
    pairs = [("trastuzumab", "trastuzumab_VH", "trastuzumab_VK"), ("adalimumab", "adalimumab_VH", "adalimumab_VK"),
             ("chimera", "pembrolizumab_VH", "avelumab_VL")]
    csv = tmp_path / "humab_pairs.csv"
    with open(csv, "w") as f:
        f.write(",type,name,h_seq,l_seq\n")
        for i, (name, h, l) in enumerate(pairs):
            f.write(f"{i},mouse,{name},{KNOWN[h][0]},{KNOWN[l][0]}\n")
    out = cli.main(["--ckpt", str(ckdir / "hudiffab.pt"), "--data_fpath", str(csv), "--numbering", "builtin",
                    "--batch_size", "2", "--seed", "11"])
    assert "_humab_" in os.path.basename(os.path.dirname(out))
    lines = open(out).read().splitlines()
    assert len(lines) == 1 + 2 * len(pairs)
    for i, (name, h, l) in enumerate(pairs):
        m, s = lines[1 + 2 * i].split(","), lines[2 + 2 * i].split(",")
        assert m == ["mouse", name, KNOWN[h][0], KNOWN[l][0]] and s[:2] == ["humanization", name + "human_sample"]
        for chain_seq, key in ((s[2], h), (s[3], l)):
            pos = 0
            for cdr in KNOWN[key][2:]:
                core = cdr[2:] if cdr is KNOWN[key][4] else cdr            # CDR3-IMGT starts 2 before Kabat's
                pos = chain_seq.index(core[:max(3, len(core) - 3)], pos) + 1
            assert chain_seq != KNOWN[key][0]                               # frameworks were resampled


def test_nanobody_cli_from_raw_sequences(tmp_path):
    """Raw VHH strings -> built-in slotter -> sampler; with random weights no sample parses as a heavy domain, so
    every input exhausts its tries and the last sweep is written (nanosample.py:346-349)."""
    import torch
    from hudiff_amd import checkpoint as ck
    from hudiff_amd.cli import nanosample as cli
    from test_numbering import KNOWN
    cfg = dict(load_cfg("nb"), dropout=0.5)
    sd = {"infilling_pretrain." + k: torch.from_numpy(v) for k, v in load_weights("nb").items()}
    ckdir = tmp_path / "run" / "checkpoints"
    ckdir.mkdir(parents=True)
    torch.save({"config": ck.EasyDict({"name": "infilling", "model": {}}), "infilling_params": ck.EasyDict(cfg),
                "abnativ_params": {}, "model": sd}, ckdir / "hudiffnb.pt")
    vhh = [KNOWN["caplacizumab_VHH"][0], KNOWN["trastuzumab_VH"][0]]
    csv = tmp_path / "nanobert.csv"
    with open(csv, "w") as f:
        f.write(",vhhseq\n" + "".join(f"{i},{s}\n" for i, s in enumerate(vhh)))
    out = cli.main(["--ckpt", str(ckdir / "hudiffnb.pt"), "--data_fpath", str(csv), "--numbering", "builtin",
                    "--try_number", "3", "--seed", "3"])
    lines = open(out).read().splitlines()
    assert lines[0] == "Specific,name,hseq," and len(lines) == 1 + 2 * len(vhh)
    for i, s in enumerate(vhh):
        assert lines[1 + 2 * i] == f"nano,{i},{s}"
        got = lines[2 + 2 * i].split(",")
        assert got[:2] == ["humanization", f"{i}human_sample"]
        key = "caplacizumab_VHH" if i == 0 else "trastuzumab_VH"
        pos = 0
        for cdr in KNOWN[key][2:]:                                   # CDR-IMGT 1-3 are not sampled (HEAVY_CDR_INDEX)
            pos = got[2].index(cdr, pos) + 1
    log = open(os.path.join(os.path.dirname(out), "log.txt")).read()
    assert log.count("Need to re sample again.") >= 2 * 2            # tries 3 -> two rejected sweeps, third written


def _pdb_fasta(path):
    from test_numbering import KNOWN
    vh, vl, vhh = KNOWN["adalimumab_VH"][0], KNOWN["adalimumab_VK"][0], KNOWN["caplacizumab_VHH"][0]
    path.write_text(">7ZZZ_1|Chain A|Spike protein S1|virus\nTNLCPFGEVFNATRFASVYAWN\n"
                    f">7ZZZ_2|Chain B[auth H]|mAb heavy chain|Mus musculus\n{vh}ASTKGPSVFPLAP\n"
                    f">7ZZZ_3|Chain C[auth L]|mAb light chain|Mus musculus\n{vl}RTVAAPSVFIFPPS\n"
                    f">7ZZZ_4|Chain D|Nanobody X|Vicugna pacos\n{vhh}HHHHHH\n")
    return vh, vl, vhh


def test_single_antibody_cli(tmp_path):
    """sample_for_anti_cdr: PDB-style FASTA -> domain cut -> sample_number rows (duplicates dropped but counted)."""
    import torch
    from hudiff_amd import checkpoint as ck
    from hudiff_amd.cli import sample_for_anti_cdr as cli
    cfg = dict(load_cfg("ab"), dropout=0.2)
    sd = {k: torch.from_numpy(v) for k, v in load_weights("ab").items()}
    ckpt = tmp_path / "hudiffab.pt"
    torch.save({"fineconfig": ck.EasyDict({}), "pretrain_config": ck.EasyDict({"name": "trans_oadm", "model": cfg}), "model": sd}, ckpt)
    fa = tmp_path / "7zzz.fasta"
    vh, vl, _ = _pdb_fasta(fa)
    out = cli.main(["--ckpt", str(ckpt), "--anti_complex_fasta", str(fa), "--log_dirpath", str(tmp_path / "logs"),
                    "--batch_size", "3", "--sample_number", "7", "--numbering", "builtin"])
    assert re.match(r"7zzz_shuffle_pair_\d{4}_", os.path.basename(os.path.dirname(out)))
    lines = open(out).read().splitlines()
    assert lines[0] == "Specific,name,hseq,lseq," and lines[1] == f"mouse,7zzz,{vh},{vl}"
    human = [ln.split(",") for ln in lines[2:]]
    assert 1 <= len(human) <= 7 and all(h[:2] == ["humanization", "7zzzhuman_sample"] for h in human)
    assert len({(h[2], h[3]) for h in human}) == len(human)                 # no duplicates written
    assert open(os.path.join(os.path.dirname(out), "log.txt")).read().count("Already Sample number") == 7
    same = cli.main(["--ckpt", str(ckpt), "--heavy_seq", vh, "--light_seq", vl, "--log_dirpath", str(tmp_path / "logs2"),
                     "--batch_size", "3", "--sample_number", "7", "--numbering", "builtin"])
    assert [ln.split(",")[2:] for ln in open(same).read().splitlines()[2:]] == [h[2:] for h in human]   # same seed, same draw
    assert "Unkown_shuffle_pair" in os.path.basename(os.path.dirname(same))


def test_single_nanobody_cli(tmp_path):
    import torch
    from hudiff_amd import checkpoint as ck
    from hudiff_amd.cli import sample_for_nano_cdr as cli
    cfg = dict(load_cfg("nb"), dropout=0.5)
    sd = {"infilling_pretrain." + k: torch.from_numpy(v) for k, v in load_weights("nb").items()}
    ckpt = tmp_path / "hudiffnb.pt"
    torch.save({"config": ck.EasyDict({"name": "infilling", "model": {}}), "infilling_params": ck.EasyDict(cfg),
                "abnativ_params": {}, "model": sd}, ckpt)
    fa = tmp_path / "7zzz.fasta"
    _, _, vhh = _pdb_fasta(fa)
    out = cli.main(["--ckpt", str(ckpt), "--nano_complex_fasta", str(fa), "--batch_size", "2", "--sample_number", "5",
                    "--numbering", "builtin", "--structure", "True"])
    log_dir = os.path.dirname(out)
    assert os.path.dirname(log_dir) == str(tmp_path) and re.match(r"7zzz_finetune_vh_vhh_\d{4}_", os.path.basename(log_dir))
    lines = open(out).read().splitlines()
    assert lines[0] == "Specific,name,hseq," and lines[1] == f"Nano,7zzz,{vhh}HHHHHH"
    human = [ln.split(",") for ln in lines[2:]]
    assert all(h[:2] == ["humanization", "7zzz"] for h in human) and len(human) <= 5
    log = open(os.path.join(log_dir, "log.txt")).read()
    # random weights: samples do not number as heavy domains and are dropped (the reference would raise here)
    assert log.count("Already Sample number") + log.count("dropped") >= 1
    assert os.path.exists(os.path.join(log_dir, "sample_identity.fa")) and os.path.isdir(os.path.join(log_dir, "sample_human_pdb"))


def _recording_sample_jobs(monkeypatch, cli):
    """Wrap the CLI's sample_jobs so that the test sees the jobs it built and the token rows the HIP path returned."""
    seen = {}
    real = cli.sample_jobs

    def rec(model, jobs, *a, **kw):
        res = real(model, jobs, *a, **kw)
        seen["jobs"], seen["result"] = list(jobs), res
        return res
    monkeypatch.setattr(cli, "sample_jobs", rec)
    return seen


def test_antibody_cli_inpaint_with_grafted_chains(tmp_path, monkeypatch):
    """--sample_method inpaint --grafted_fpath through the HIP path (sample.py:283-310, 486-489; VERDICT r2 missing #3).  The
    graft here is synthetic (abnumber's germline database is not available offline): mouse CDRs on a framework that differs
    from the mouse framework at every third position.  Rows, order and names as the reference writes them; every identity
    position of the graft (CDR-IMGT + agreeing framework) untouched; every other CDR-IMGT framework slot -- mismatches and
    empty slots alike -- sampled to an id in [0, 21]; the similarity search compares with the MOUSE chains."""
    import torch
    from hudiff_amd import checkpoint as ck
    from hudiff_amd import inputs as I
    from hudiff_amd import tables as T
    from hudiff_amd.cli import sample as cli
    from test_host_logic import fake_numbering
    cfg = dict(load_cfg("ab"), dropout=0.2)
    sd = {k: torch.from_numpy(v) for k, v in load_weights("ab").items()}
    ckdir = tmp_path / "run" / "checkpoints"
    ckdir.mkdir(parents=True)
    torch.save({"fineconfig": ck.EasyDict({}), "pretrain_config": ck.EasyDict({"name": "trans_oadm", "model": cfg}), "model": sd},
               ckdir / "hudiffab.pt")
    csv, nb = _write_inputs(tmp_path, "ab", 4)
    mouse = [ln.split(",") for ln in open(csv).read().splitlines()[1:] if ln.startswith("mouse")]
    cdr = {"H": np.array(T.HEAVY_CDR_INDEX) != 0, "L": np.array(T.LIGHT_CDR_INDEX) != 0}
    names = {"H": T.HEAVY_POSITIONS, "L": T.LIGHT_POSITIONS}
    grafted, kept_slots = [], []
    for _, name, h, l in mouse:
        g, kept = {}, []
        for key, seq, chain in (("h", h, "H"), ("l", l, "L")):
            numbered = fake_numbering(seq, chain)
            graft, ident = {}, []
            for n, (pos, aa) in enumerate(numbered.items()):
                slot = names[chain].index(pos)
                if cdr[chain][slot] or n % 3:                   # CDR residues and two thirds of the framework agree with the mouse
                    graft[pos] = aa
                    ident.append(pos)
                else:
                    graft[pos] = "A" if aa != "A" else "G"      # germline differs here: not an identity position
            g[key], g["identity_" + key] = graft, ident
            kept += [(0 if chain == "H" else T.H_LEN) + names[chain].index(p) for p in ident]
        g["l_chain"] = "K"
        grafted.append(g)
        kept_slots.append(np.array(sorted(kept)))
    gpath = tmp_path / "grafted.jsonl"
    gpath.write_text("".join(json.dumps(g) + "\n" for g in grafted))
    seen = _recording_sample_jobs(monkeypatch, cli)
    out = cli.main(["--ckpt", str(ckdir / "hudiffab.pt"), "--data_fpath", str(csv), "--numbered_fpath", str(nb),
                    "--sample_method", "inpaint", "--grafted_fpath", str(gpath), "--batch_size", "3", "--seed", "13"])
    lines = open(out).read().splitlines()
    assert lines[0] == "Specific,name,hseq,lseq," and len(lines) == 1 + 2 * 4
    fr = np.array(T.HEAVY_CDR_INDEX + T.LIGHT_CDR_INDEX) == 0
    for j, (_, name, h, l) in enumerate(mouse):
        assert lines[1 + 2 * j] == f"mouse,{name},{h},{l}"
        job, rows = seen["jobs"][j], seen["result"][j, 0]
        kept = kept_slots[j]
        want_loc = np.nonzero(fr & ~np.isin(np.arange(T.AB_LEN), kept))[0]          # every non-identity framework slot, empty ones too
        assert np.array_equal(np.sort(job.loc), want_loc) and (job.tokens[want_loc] == 22).all()
        for r in range(3):
            assert np.array_equal(rows[r][kept], job.tokens[kept])                # identity positions untouched
            assert ((rows[r][want_loc] >= 0) & (rows[r][want_loc] <= 21)).all()   # everything else sampled
            untouched = np.setdiff1d(np.arange(T.AB_LEN), want_loc)
            assert np.array_equal(rows[r][untouched], job.tokens[untouched])
        best = cli.select_most_similar(job.parent["tokens"], rows)                 # sample.py:524: similarity to the mouse chains
        g_h, g_l = I.untokenize_antibody(rows[best])
        assert lines[2 + 2 * j] == f"humanization,{name}human_sample,{g_h},{g_l}"
    assert not (seen["result"] == 22).any()


def test_antibody_cli_pretrain_checkpoint(tmp_path, monkeypatch):
    """--ckpt_version pretrain (sample.py:148-151, 446-449): the {'config', 'model'} envelope, the CDR-IMGT mask over ALL
    framework slots, empty ones included -> T = 185 denoiser steps per row, the longest schedule of the path."""
    import torch
    from hudiff_amd import checkpoint as ck
    from hudiff_amd import tables as T
    from hudiff_amd.cli import sample as cli
    cfg = dict(load_cfg("ab"), dropout=0.2)
    sd = {"module." + k: torch.from_numpy(v) for k, v in load_weights("ab").items()}       # DataParallel prefix (antibody_train.py:23-30)
    ckdir = tmp_path / "run" / "checkpoints"
    ckdir.mkdir(parents=True)
    torch.save({"config": ck.EasyDict({"name": "trans_oadm", "model": cfg}), "model": sd, "iteration": 7}, ckdir / "pre.pt")
    csv, nb = _write_inputs(tmp_path, "ab", 3)
    seen = _recording_sample_jobs(monkeypatch, cli)
    out = cli.main(["--ckpt", str(ckdir / "pre.pt"), "--ckpt_version", "pretrain", "--data_fpath", str(csv), "--numbered_fpath", str(nb),
                    "--batch_size", "2", "--seed", "3"])
    assert "_pretrain_search_simi_True_" in os.path.basename(os.path.dirname(out))
    lines = open(out).read().splitlines()
    assert len(lines) == 1 + 2 * 3 and all(lines[2 + 2 * j].startswith(f"humanization,m{j}human_sample,") for j in range(3))
    fr = np.nonzero(np.array(T.HEAVY_CDR_INDEX + T.LIGHT_CDR_INDEX) == 0)[0]
    assert len(fr) == 185
    for j in range(3):
        job, rows = seen["jobs"][j], seen["result"][j, 0]
        assert len(job.loc) == 185 and np.array_equal(np.sort(job.loc), fr)
        cdr = np.setdiff1d(np.arange(T.AB_LEN), fr)
        for r in range(2):
            assert ((rows[r][fr] >= 0) & (rows[r][fr] <= 21)).all() and np.array_equal(rows[r][cdr], job.tokens[cdr])
