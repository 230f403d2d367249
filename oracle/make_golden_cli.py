"""TEST INFRASTRUCTURE ONLY -- CSV text and FASTA records written by the REFERENCE's own command-line programs.

    python oracle/make_golden_cli.py            (build container: needs /root/reference; ~2 min of CPU)

SURVEY.md section 8(c), last row: "(argv, input CSV slice, pre-numbered slots, injected noise) -> output CSV text".  This script runs the
``if __name__ == '__main__'`` bodies of antibody_scripts/sample.py (:379-588) and nanobody_scripts/nanosample.py (:195-368) -- the
reference's argparse, log-dir naming, checkpoint loading, input preparation, sampling loop, CSV rows and FASTA hand-over, unmodified --
with ``runpy`` on a micro checkpoint, and commits what they produced: tests/golden/cli_*.json.  tests/test_gpu_cli_golden.py feeds the
same argv, input CSV, numbering and noise to ``hudiff_amd.cli.*`` and compares the CSV byte for byte.

What is NOT the reference in such a run, and why (all third-party, absent offline):
* ``anarci.number`` / ``abnumber.Chain(seq).chain_type`` answer from a table of pre-numbered residues (hudiff_amd.numbering's slotter on
  the input sequences) -- the table is stored in the fixture and handed to the drop-in CLI as ``--numbered_fpath``;
* ``abnumber.Chain.to_fasta`` / ``Bio.SeqIO.write`` are RECORDING stubs: the fixture stores the records (id, description, sequence) the
  reference handed to those writers, in order; the text format of the file is the third-party writers' (fasta-2line / 60-column fasta);
* ``torch.multinomial`` is recorded (oracle/make_golden.py Recorder: argmax(p / Exp(1)), asserted equal to torch's own draw);
* ``torch.load`` is called with ``weights_only=False`` (torch >= 2.6 refuses the EasyDict the reference's checkpoints hold);
* ``easydict.EasyDict`` is hudiff_amd.checkpoint's attribute-dictionary class (the package is not installed).
No reference source text is stored: inputs, numbering table, noise, outputs.
"""
from __future__ import annotations

import contextlib
import io
import json
import os
import runpy
import shutil
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
from hudiff_amd import checkpoint as _ck  # noqa: E402
_ck._ensure_easydict()          # BEFORE ref_import's dummy finder exists: sys.modules['easydict'].EasyDict = the attribute-dictionary class
import ref_import  # noqa: E402
import make_golden as mg  # noqa: E402  (Recorder)

OUT = os.path.join(ROOT, "tests", "golden")
REF = ref_import.REFERENCE_ROOT


class _Stubs:
    """Third-party stand-ins for one run (see module docstring)."""

    def __init__(self, table):
        self.table = table                      # sequence -> (numbered dict, chain type)
        self.fasta = {}                         # file name -> [(id, description, sequence)]
        self.asked = []                         # sequences the reference asked to number, in order

    def install(self):
        stubs = self

        def number(seq, scheme="imgt"):
            assert scheme == "imgt"
            stubs.asked.append(seq)
            d, ct = stubs.table[seq]
            numbering = []
            for key, aa in d.items():
                digits = "".join(c for c in key if c.isdigit())
                numbering.append(((int(digits), key[len(digits):] or " "), aa))
            return numbering, ct

        class Chain:
            def __init__(self, seq, scheme="imgt", **kw):
                self.seq, self.scheme, self.name = seq, scheme, None
                ent = stubs.table.get(seq)
                self.chain_type = ent[1] if ent else "H"

            @staticmethod
            def to_fasta(chains, path_or_fd, keep_tail=False, description=""):
                chains = chains if isinstance(chains, (list, tuple)) else [chains]
                name = os.path.basename(getattr(path_or_fd, "name", str(path_or_fd)))
                for c in chains:
                    stubs.fasta.setdefault(name, []).append((str(c.name), description, c.seq))

        class Seq(str):
            pass

        class SeqRecord:
            def __init__(self, seq, id="<unknown id>", name="<unknown name>", description="<unknown description>", **kw):
                self.seq, self.id, self.name, self.description = seq, id, name, description

        def seqio_write(records, handle, fmt):
            records = records if isinstance(records, (list, tuple)) else [records]
            name = os.path.basename(getattr(handle, "name", str(handle)))
            for r in records:
                stubs.fasta.setdefault(name, []).append((r.id, r.description, str(r.seq)))
            return len(records)

        def cal_all_preservation(ref_chain, test_chain):          # patent_eval.py:150-159 (abnumber alignment): never decides with ONE candidate
            return 1.0
        mods = {
            "anarci": dict(number=number), "abnumber": dict(Chain=Chain),
            "Bio": dict(), "Bio.Seq": dict(Seq=Seq), "Bio.SeqRecord": dict(SeqRecord=SeqRecord),
            "Bio.SeqIO": dict(write=seqio_write, parse=lambda *a, **k: iter(())),
            "patent_eval": dict(cal_all_preservation=cal_all_preservation),
        }
        for name, attrs in mods.items():
            m = ref_import._DummyModule(name)      # anything else the reference imports from these packages stays a permissive dummy
            m.__dict__.update(attrs)
            sys.modules[name] = m
        sys.modules["Bio"].SeqIO = sys.modules["Bio.SeqIO"]
        sys.modules["Bio"].Seq = sys.modules["Bio.Seq"]
        sys.modules["Bio"].SeqRecord = sys.modules["Bio.SeqRecord"]


def _run_main(script, argv, stubs):
    """The reference program's __main__ body under the stubs; -> (recorded noise list, stdout text)."""
    ref_import.install()
    stubs.install()
    for extra in (os.path.join(REF, "antibody_scripts"), os.path.join(REF, "nanobody_scripts")):
        if extra not in sys.path:
            sys.path.append(extra)
    real_load = torch.load
    torch.load = lambda f, *a, **k: real_load(f, *a, **{**k, "weights_only": False})
    old_argv, sink = sys.argv, io.StringIO()
    sys.argv = [script] + argv
    try:
        with mg.Recorder() as rec, contextlib.redirect_stdout(sink), contextlib.redirect_stderr(io.StringIO()):
            runpy.run_path(os.path.join(REF, script), run_name="__main__")
    finally:
        sys.argv = old_argv
        torch.load = real_load
    return rec.q, sink.getvalue()


def _micro(kind):
    z = np.load(os.path.join(OUT, f"micro_{kind}_config.npz"))
    cfg = {}
    for k, v in z.items():
        v = v.item() if v.ndim == 0 else v
        cfg[k] = str(v) if isinstance(v, (str, np.str_)) else v
    cfg["dropout"] = 0.0                        # the only noise of the run is torch.multinomial's (recorded)
    sd = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(OUT, f"micro_{kind}_weights.npz")).items()}
    # the reference loads strictly: its recomputable buffers (complex '...rope', 'pos_embedding.pe') must be in the file, as in a checkpoint
    # its own training scripts wrote -- take them from an instance of the reference class
    AntiTFNet, NanoAntiTFNet = ref_import.reference_models()
    model = (AntiTFNet if kind == "ab" else NanoAntiTFNet)(**cfg)
    missing = model.load_state_dict(sd, strict=False)
    assert not missing.unexpected_keys and all(k.endswith(".rope") or k.endswith("pos_embedding.pe") for k in missing.missing_keys), missing
    return cfg, {k: v.clone() for k, v in model.state_dict().items()}


def _case_ab(name, extra_argv, n_rows, batch_size):
    import pandas as pd
    from hudiff_amd.checkpoint import EasyDict
    from hudiff_amd.numbering import number_sequence_builtin
    hu = pd.read_csv(os.path.join(REF, "data/antibody_eval_data/HuAb348_data/humanization_pair_data_filter.csv"))
    mouse = hu[hu["type"] == "mouse"].reset_index(drop=True)
    pick = [3, 118, 251, 340][:n_rows]
    lines, table, numbered = ["type,name,h_seq,l_seq"], {}, []
    for i in pick:
        h, l, nm = mouse.loc[i, "h_seq"], mouse.loc[i, "l_seq"], str(mouse.loc[i, "name"]).replace(",", "_")
        lines.append(f"mouse,{nm},{h},{l}")
        lines.append(f"human,{nm},{h},{l}")     # non-mouse rows are skipped by get_mouse_line (sample.py:314-317)
        hd, ht = number_sequence_builtin(h)
        ld, lt = number_sequence_builtin(l)
        table[h], table[l] = (hd, ht), (ld, lt)
        numbered.append({"name": nm, "h": hd, "l": ld, "h_chain": ht, "l_chain": lt})
    cfg, sd = _micro("ab")
    work = tempfile.mkdtemp(prefix="refcli_")
    try:
        ckdir = os.path.join(work, "run", "checkpoints")
        os.makedirs(ckdir)
        ckpt = os.path.join(ckdir, "hudiffab.pt")
        torch.save({"fineconfig": EasyDict({}), "pretrain_config": EasyDict({"name": "trans_oadm", "model": EasyDict(cfg)}), "model": sd}, ckpt)
        csv = os.path.join(work, "pairs.csv")
        open(csv, "w").write("\n".join(lines) + "\n")
        argv = ["--batch_size", str(batch_size), "--seed", "11"] + extra_argv
        stubs = _Stubs(table)
        q, _ = _run_main("antibody_scripts/sample.py", ["--ckpt", ckpt, "--data_fpath", csv] + argv, stubs)
        run_dirs = [d for d in os.listdir(os.path.join(work, "run")) if d != "checkpoints"]
        assert len(run_dirs) == 1
        log_dir = os.path.join(work, "run", run_dirs[0])
        out_csv = open(os.path.join(log_dir, "sample_humanization_result.csv"), encoding="UTF-8").read()
        fixture = {"program": "antibody_scripts/sample.py", "argv": argv, "log_dir_prefix": run_dirs[0][:-len("2026_01_01__00_00_00")],
                   "input_csv_name": "pairs.csv", "input_csv": "\n".join(lines) + "\n", "numbered": numbered, "ckpt_name": "hudiffab.pt",
                   "checkpoint": "micro_ab_{config,weights}.npz, dropout 0, envelope {'fineconfig', 'pretrain_config', 'model'}",
                   "output_csv": out_csv, "fasta_records": stubs.fasta, "numbering_calls": len(stubs.asked)}
        return fixture, q
    finally:
        shutil.rmtree(work, ignore_errors=True)


def _case_nb(name, extra_argv, n_rows, batch_size):
    import pandas as pd
    from hudiff_amd.checkpoint import EasyDict
    from hudiff_amd.numbering import number_sequence_builtin
    vhh = pd.read_csv(os.path.join(REF, "data/nanobody_eval_data/abnativ_select_vhh.csv"))
    pick = [2, 140, 299][:n_rows]
    lines, table, numbered = ["vhhseq"], {}, []
    for i in pick:
        s = vhh.loc[i, "vhhseq"]
        lines.append(s)
        d, t = number_sequence_builtin(s)
        table[s] = (d, t)
        numbered.append({"h": d})
    cfg, sd = _micro("nb")
    work = tempfile.mkdtemp(prefix="refcli_")
    try:
        ckdir = os.path.join(work, "run", "checkpoints")
        os.makedirs(ckdir)
        ckpt = os.path.join(ckdir, "nanopretrain.pt")
        torch.save({"config": EasyDict({"name": "nano", "model": EasyDict(cfg)}), "model": sd}, ckpt)
        csv = os.path.join(work, "vhh_filter.csv")
        open(csv, "w").write("\n".join(lines) + "\n")
        argv = ["--model", "pretrain", "--batch_size", str(batch_size), "--seed", "13"] + extra_argv
        stubs = _Stubs(table)
        q, _ = _run_main("nanobody_scripts/nanosample.py", ["--ckpt", ckpt, "--data_fpath", csv] + argv, stubs)
        run_dirs = [d for d in os.listdir(os.path.join(work, "run")) if d != "checkpoints"]
        assert len(run_dirs) == 1
        log_dir = os.path.join(work, "run", run_dirs[0])
        out_csv = open(os.path.join(log_dir, "sample_humanization_result.csv"), encoding="UTF-8").read()
        fixture = {"program": "nanobody_scripts/nanosample.py", "argv": argv, "log_dir_prefix": run_dirs[0][:-len("2026_01_01__00_00_00")],
                   "input_csv_name": "vhh_filter.csv", "input_csv": "\n".join(lines) + "\n", "numbered": numbered, "ckpt_name": "nanopretrain.pt",
                   "checkpoint": "micro_nb_{config,weights}.npz, dropout 0, envelope {'config', 'model'} (--model pretrain)",
                   "output_csv": out_csv, "fasta_records": stubs.fasta, "numbering_calls": len(stubs.asked)}
        return fixture, q
    finally:
        shutil.rmtree(work, ignore_errors=True)


CASES = [
    # the reference's defaults: one replica per antibody, similarity search on (one candidate: the search cannot decide anything)
    ("cli_ab_default", _case_ab, [], 3, 1),
    # three replicas written one by one: `--similarity_search ""` is how argparse's type=bool is switched off (sample.py:402, 525-538)
    ("cli_ab_three_rows", _case_ab, ["--similarity_search", "", "--sample_number", "3", "--ckpt_version", "finetune", "--fa_version", "v9"], 2, 3),
    ("cli_nb_default", _case_nb, [], 3, 1),
    ("cli_nb_inpaint_two_rows", _case_nb, ["--inpaint_sample", "True", "--sample_number", "2", "--fa_version", "vX"], 2, 2),
]


def main():
    only = sys.argv[1:]
    for name, fn, extra, n_rows, bs in CASES:
        if only and name not in only:
            continue
        fixture, q = fn(name, extra, n_rows, bs)
        # noise in loop order: input row by input row, step by step, [batch_size, 22] per step
        fixture["noise_file"] = name + "_noise.npz"
        np.savez_compressed(os.path.join(OUT, fixture["noise_file"]), q=np.stack(q).astype(np.float32) if q else np.zeros((0, bs, 22), np.float32))
        fixture["batch_size"] = bs
        with open(os.path.join(OUT, name + ".json"), "w") as f:
            json.dump(fixture, f, indent=0)
        print(name, "rows of CSV:", fixture["output_csv"].count("\n"), "draws:", len(q), "fasta:", {k: len(v) for k, v in fixture["fasta_records"].items()})


if __name__ == "__main__":
    main()
