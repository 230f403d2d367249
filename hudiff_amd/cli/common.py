"""Pieces shared by the two drop-in samplers: log-dir naming, logger, numbered-input readers, FASTA."""
from __future__ import annotations

import json
import logging
import os
import time


def get_new_log_dir(root="./logs", prefix="", tag=""):
    """utils/misc.py:10-24 -- {prefix}_{YYYY_mm_dd__HH_MM_SS}[_{tag}] under root."""
    fn = time.strftime("%Y_%m_%d__%H_%M_%S", time.localtime())
    if prefix != "":
        fn = prefix + "_" + fn
    if tag != "":
        fn = fn + "_" + tag
    log_dir = os.path.join(root, fn)
    os.makedirs(log_dir, exist_ok=True)
    return log_dir


def get_logger(name, log_dir=None):
    """utils/misc.py:34-53 -- same format string, stream + log.txt handlers."""
    logger = logging.getLogger(name)
    logger.setLevel(logging.DEBUG)
    logger.handlers.clear()
    fmt = logging.Formatter("[%(asctime)s::%(name)s::%(levelname)s] %(message)s")
    sh = logging.StreamHandler()
    sh.setLevel(logging.DEBUG)
    sh.setFormatter(fmt)
    logger.addHandler(sh)
    if log_dir is not None:
        fh = logging.FileHandler(os.path.join(log_dir, "log.txt"))
        fh.setLevel(logging.DEBUG)
        fh.setFormatter(fmt)
        logger.addHandler(fh)
    return logger


def add_runtime_args(p):
    """Flags every drop-in CLI adds beside the reference's own: the precision route of the device library and a self-launching
    multi-GPU mode (rows shard across GPUs, one RCCL gather)."""
    p.add_argument("--precision", choices=["default", "split", "f32_gemm", "f32_all"], default="default",
                   help="precision route of libhudiff_hip (include/hudiff_hip.h): default = split (fp32 products as three fp16 MFMAs "
                        "on (hi, lo) operand splits, fp32 accumulation; logits within 1e-4, the reference's traces bit for bit), "
                        "f32_gemm = fp32 MFMA GEMMs + split attention core, f32_all = every product on the fp32 MFMA pipe")
    p.add_argument("--gpus", type=int, default=None,
                   help="N > 1 without a launcher: start N ranks of this command under torch.distributed.run (one process per GPU, "
                        "127.0.0.1 rendezvous); under torchrun it must equal WORLD_SIZE")
    return p


def relaunch_if_asked(args, module, argv):
    """`python -m hudiff_amd.cli.X --gpus N` without a launcher around it: re-execute as N ranks under torch.distributed.run (the
    same shape bench.py uses) and return that job's exit code; None when this process should carry on (single rank, or already a
    rank of a launched job).  A mismatch between --gpus and the world size a launcher formed is an error, never a silent run."""
    import subprocess
    import sys
    if not args.gpus or args.gpus < 1:
        return None
    world = os.environ.get("WORLD_SIZE")
    if world is not None:
        if int(world) != args.gpus:
            raise SystemExit(f"{module}: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
        return None
    if args.gpus == 1:
        return None
    cmd = relaunch_command(module, args.gpus, list(sys.argv[1:] if argv is None else argv))
    return subprocess.call(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))


def relaunch_command(module, gpus, argv):
    import socket
    import sys
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), "-m", module] + argv


def str2bool_like_reference(v):
    """argparse ``type=bool`` of the reference: any non-empty string is True (sample.py:402, 411-414)."""
    return bool(v)


def load_numbered(path):
    """Pre-numbered sequences for machines without ANARCI: JSON lines, one object per input row, e.g.
    {"name": "ab1", "h": {"1": "E", "2": "V", ..., "111A": "G"}, "l": {...}, "l_chain": "K"}   (antibody)
    {"h": {...}}                                                                               (nanobody)
    keys are IMGT position strings as ``get_pad_seq`` builds them (sample.py:84-88)."""
    rows = []
    with open(path) as f:
        for line in f:
            line = line.strip()
            if line:
                rows.append(json.loads(line))
    return rows


def write_fasta_2line(records, path):
    """[(id, description, sequence)] -> '>id description\\nSEQ' (the 'fasta-2line' flavour)."""
    with open(path, "w") as f:
        for rid, desc, seq in records:
            f.write(f">{rid} {desc}\n{seq}\n" if desc else f">{rid}\n{seq}\n")


def write_fasta_wrapped(records, path, width=60):
    """Bio.SeqIO 'fasta' flavour: header '>id description', sequence wrapped at 60 columns."""
    with open(path, "w") as f:
        for rid, desc, seq in records:
            f.write(f">{rid} {desc}\n")
            for i in range(0, len(seq), width):
                f.write(seq[i:i + width] + "\n")


def read_fasta(path):
    """-> [(description line without '>', sequence)], multi-line records joined (stand-in for Bio.SeqIO.parse)."""
    records, desc, chunks = [], None, []
    with open(path) as f:
        for line in f:
            line = line.strip()
            if not line:
                continue
            if line.startswith(">"):
                if desc is not None:
                    records.append((desc, "".join(chunks)))
                desc, chunks = line[1:], []
            elif desc is not None:
                chunks.append(line)
    if desc is not None:
        records.append((desc, "".join(chunks)))
    return records


def split_fasta_for_save(csv_path, human_seqs):
    """One '{idx}_human.fasta' per humanized sample under sample_human_fa/ next to the CSV, plus the empty
    sample_human_pdb/ the structure predictor fills.  Nanobody (nanosample.py:163-182): record '{idx}_human_H', wrapped;
    antibody (sample.py:326-349, items are (VH, VL) pairs): '>{idx}_human_H VH' and '>{idx}_human_L VL', two-line records
    as ``Chain.to_fasta`` writes them."""
    base = os.path.dirname(csv_path)
    fa_dir, pdb_dir = os.path.join(base, "sample_human_fa"), os.path.join(base, "sample_human_pdb")
    os.makedirs(fa_dir, exist_ok=True)
    os.makedirs(pdb_dir, exist_ok=True)
    for idx, seq in enumerate(human_seqs):
        path = os.path.join(fa_dir, f"{idx}_human.fasta")
        if isinstance(seq, str):
            write_fasta_wrapped([(f"{idx}_human_H", "<unknown description>", seq)], path)
        else:
            write_fasta_2line([(f"{idx}_human_H", "VH", seq[0]), (f"{idx}_human_L", "VL", seq[1])], path)
    return fa_dir
