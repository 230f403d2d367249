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
    out2 = cli.main(["--ckpt", str(ckdir / "hudiffab.pt"), "--data_fpath", str(csv), "--numbered_fpath", str(nb),
                     "--batch_size", "3", "--seed", "5", "--device_batch", "4"])
    assert open(out2).read() == open(out).read() or os.path.dirname(out2) == log_dir


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
