"""The drop-in command-line programs against CSV text written by the REFERENCE's own ``__main__`` bodies (VERDICT r5 "Next" #5,
SURVEY.md section 8c last row).  tests/golden/cli_*.json were produced by oracle/make_golden_cli.py, which ran
antibody_scripts/sample.py (:379-588) and nanobody_scripts/nanosample.py (:195-368) unmodified on a micro checkpoint with a numbering
table in place of ANARCI and the ``torch.multinomial`` noise recorded.  Here the same argv, input CSV, numbering table
(``--numbered_fpath``) and noise (``--q_noise_fpath``) go to ``hudiff_amd.cli.sample`` / ``hudiff_amd.cli.nanosample``:

* ``sample_humanization_result.csv`` must equal the reference's text BYTE FOR BYTE (header sample.py:470 / nanosample.py:296, parental
  rows :495 / :312, humanization rows :528, :536 / :344, row order, one visiting order per input row from the seeded numpy stream);
* the log directory name must carry the reference's prefix (sample.py:426-443, nanosample.py:237-249);
* ``sample_identity.fa`` must hold the records (id, description, sequence) the reference handed to abnumber's / Biopython's writers,
  in order, in those writers' text formats (fasta-2line: sample.py:43-54; 60-column fasta: nanosample.py:40-51)."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, load_cfg, load_weights

CASES = ["cli_ab_default", "cli_ab_three_rows", "cli_nb_default", "cli_nb_inpaint_two_rows"]


def _setup(tmp_path, fx, kind):
    import torch
    from hudiff_amd import checkpoint as ck
    cfg = dict(load_cfg(kind), dropout=0.0)
    sd = {k: torch.from_numpy(v) for k, v in load_weights(kind).items()}
    ckdir = tmp_path / "run" / "checkpoints"
    ckdir.mkdir(parents=True)
    ckpt = ckdir / fx["ckpt_name"]
    if kind == "ab":
        torch.save({"fineconfig": ck.EasyDict({}), "pretrain_config": ck.EasyDict({"name": "trans_oadm", "model": cfg}), "model": sd}, ckpt)
    else:
        torch.save({"config": ck.EasyDict({"name": "nano", "model": cfg}), "model": sd}, ckpt)
    csv = tmp_path / fx["input_csv_name"]
    csv.write_text(fx["input_csv"])
    nb = tmp_path / "numbered.jsonl"
    with open(nb, "w") as f:
        for d in fx["numbered"]:
            f.write(json.dumps(d) + "\n")
    return ckpt, csv, nb


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES)
def test_cli_reproduces_the_reference_programs_output(case, tmp_path, monkeypatch):
    fx = json.load(open(os.path.join(GOLDEN, case + ".json")))
    kind = "ab" if fx["program"].endswith("sample.py") and "nano" not in fx["program"] else "nb"
    ckpt, csv, nb = _setup(tmp_path, fx, kind)
    noise = os.path.join(GOLDEN, fx["noise_file"])
    argv = ["--ckpt", str(ckpt), "--data_fpath", str(csv)] + fx["argv"] + ["--numbered_fpath", str(nb), "--q_noise_fpath", noise]
    if kind == "ab":
        from hudiff_amd.cli import sample as cli
    else:
        from hudiff_amd.cli import nanosample as cli
        # the reference's validity gate is abnumber.Chain(seq) (nanosample.py:342); in the recorded run it accepted every sample
        monkeypatch.setattr(cli, "chain_is_valid", lambda seq: True)
    out = cli.main(argv)
    log_dir = os.path.dirname(out)
    assert os.path.basename(out) == "sample_humanization_result.csv" and os.path.dirname(log_dir) == str(tmp_path / "run")
    assert os.path.basename(log_dir).startswith(fx["log_dir_prefix"]) and len(os.path.basename(log_dir)) == len(fx["log_dir_prefix"]) + len("2026_01_01__00_00_00")
    got = open(out, encoding="UTF-8").read()
    assert got == fx["output_csv"], "the CSV differs from the text the reference program wrote"
    # FASTA: the records the reference handed to the third-party writer
    from hudiff_amd.cli.common import read_fasta
    want = fx["fasta_records"]["sample_identity.fa"]
    recs = read_fasta(os.path.join(log_dir, "sample_identity.fa"))
    assert [(d, s) for d, s in recs] == [((f"{i} {d}" if d else i), s) for i, d, s in want]
    text = open(os.path.join(log_dir, "sample_identity.fa")).read().splitlines()
    if kind == "ab":                                   # fasta-2line: header, whole sequence, header, ...
        assert len(text) == 2 * len(want) and all(l.startswith(">") for l in text[0::2])
    else:                                              # Bio.SeqIO 'fasta': sequence wrapped at 60 columns
        assert all(len(l) <= 60 for l in text if not l.startswith(">"))


def test_noise_in_reference_order_rejects_a_wrong_count(tmp_path):
    """The parity aid refuses noise that does not cover exactly one draw per visited slot (an input-preparation mismatch would
    otherwise shift every later draw silently)."""
    from hudiff_amd.sampler import Job, noise_in_reference_order
    jobs = [Job(tokens=np.zeros(5, np.int32), region=np.zeros(5, np.int32), loc=np.array([1, 3])),
            Job(tokens=np.zeros(5, np.int32), region=np.zeros(5, np.int32), loc=np.array([0, 2, 4]))]
    q = np.arange(5 * 2 * 22, dtype=np.float32).reshape(5, 2, 22)
    out = noise_in_reference_order(q, jobs, 2)
    assert out.shape == (1, 3, 4, 22)
    assert np.array_equal(out[0, :2, 0:2], q[0:2]) and np.array_equal(out[0, :3, 2:4], q[2:5]) and (out[0, 2, 0:2] == 1).all()
    with pytest.raises(ValueError):
        noise_in_reference_order(q[:4], jobs, 2)
