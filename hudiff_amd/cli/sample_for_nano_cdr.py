"""Drop-in for ``nanobody_scripts/sample_for_nano_cdr.py`` (reference lines 49-209): humanize ONE nanobody given as a
PDB-style FASTA (the record whose description holds 'Nanobody').

    python -m hudiff_amd.cli.sample_for_nano_cdr --ckpt hudiffnb.pt --nano_complex_fasta fasta_file/7x2l.fasta

Same flags, log-dir naming ('{pdb}_{model}_vhh_{time}' next to the FASTA), CSV ('Specific,name,hseq,' / 'Nano,{pdb},{seq}'
/ 'humanization,{pdb},{seq}'; duplicates are skipped but count, :173-186), ``sample_identity.fa`` and, with
``--structure True``, the per-sample FASTA files.  The sampler is always the fine-tuned checkpoint's
``infilling_pretrain`` network (:104-133).  A sample that does not number as a heavy domain makes the reference raise
(``Chain(g_h)`` outside any try, :180); here it is logged and dropped.  ``--seed`` keys order and noise (the reference
leaves the RNG unseeded, :88).
"""
from __future__ import annotations

import argparse
import os

import numpy as np

from .. import dist as D
from .. import inputs as I
from ..checkpoint import load_checkpoint, nanobody_model_from_checkpoint
from ..model import NanoAntiTFNet
from ..sampler import Job, sample_jobs, seed_all
from .common import add_runtime_args, relaunch_if_asked, get_logger, get_new_log_dir, read_fasta, split_fasta_for_save, write_fasta_wrapped
from .nanosample import chain_is_valid


def build_parser():
    p = argparse.ArgumentParser(description="This program is designed to humanize non-human nanobodies.")
    p.add_argument("--ckpt", type=str, default=None)
    p.add_argument("--nano_complex_fasta", type=str, default=None)
    p.add_argument("--batch_size", type=int, default=10)
    p.add_argument("--sample_number", type=int, default=100)
    p.add_argument("--seed", type=int, default=42)
    p.add_argument("--sample_order", type=str, default="shuffle")
    p.add_argument("--sample_method", type=str, default="gen", choices=["gen", "rl_gen"])
    p.add_argument("--length_limit", type=str, default="not_equal")
    p.add_argument("--model", type=str, default="finetune_vh", choices=["pretrain", "finetune_vh"])
    p.add_argument("--fa_version", type=str, default="v_nano")
    p.add_argument("--inpaint_sample", type=eval, default=True)
    p.add_argument("--structure", type=eval, default=False)
    # additions
    p.add_argument("--numbering", choices=["auto", "anarci", "builtin"], default="auto")
    p.add_argument("--dropout", choices=["faithful", "off"], default="faithful")
    p.add_argument("--device", type=int, default=None)
    add_runtime_args(p)
    return p


def get_nano_seq_from_fasta(fpath):
    """:30-44 -- last record whose description mentions 'Nanobody'."""
    nano = None
    for desc, seq in read_fasta(fpath):
        if "Nanobody" in desc:
            nano = seq
    assert nano is not None, "Reading the fasta has problem."
    return nano


def main(argv=None):
    args = build_parser().parse_args(argv)
    rc = relaunch_if_asked(args, "hudiff_amd.cli.sample_for_nano_cdr", argv)
    if rc is not None:
        return rc
    rank, world, local_rank = D.env_rank_world()
    D.init_process_group()
    seed_all(args.seed)
    pdb_name = os.path.basename(args.nano_complex_fasta).split(".")[0]
    log_dir = logger = None
    if rank == 0:
        log_dir = get_new_log_dir(root=os.path.dirname(args.nano_complex_fasta), prefix=f"{pdb_name}_{args.model}_vhh")
        logger = get_logger("test", log_dir)
    ckpt = load_checkpoint(args.ckpt)
    _, params, state = nanobody_model_from_checkpoint(ckpt, "finetune_vh")
    model = NanoAntiTFNet(**params, device=args.device if args.device is not None else local_rank, precision=args.precision)
    model.load_state_dict(state)
    model.eval()
    if rank == 0:
        logger.info(args.ckpt)
        logger.info(args.seed)

    nano_chain = get_nano_seq_from_fasta(args.nano_complex_fasta)
    h_dict, _ = I.number_sequence(nano_chain, args.numbering)
    tok, reg, loc = I.nanobody_row(h_dict, inpaint_sample=args.inpaint_sample)
    if args.sample_order == "shuffle":
        np.random.shuffle(loc)
    passes = max(1, -(-args.sample_number // args.batch_size))
    result = sample_jobs(model, [Job(tokens=tok, region=reg, loc=loc, name=pdb_name)], args.batch_size, args.seed,
                         passes=passes, dropout=args.dropout)
    if rank != 0:
        return None
    save_fpath = os.path.join(log_dir, "sample_humanization_result.csv")
    seen, human, left = set(), [], args.sample_number
    with open(save_fpath, "a", encoding="UTF-8") as f:
        f.write("Specific,name,hseq,\n")
        f.write(f"Nano,{pdb_name},{nano_chain}\n")
        for p in range(passes):
            for r in range(args.batch_size):
                if left == 0:
                    break
                g_h = I.untokenize_nanobody(result[0, p, r])
                if g_h not in seen:
                    if chain_is_valid(g_h):
                        f.write(f"humanization,{pdb_name},{g_h}\n")
                        human.append(g_h)
                        logger.info("Already Sample number {}".format(args.sample_number - left + 1))
                        logger.info("Sample Heavy Chain Seq: {}".format(g_h))
                    else:
                        logger.info("Sample does not number as a heavy domain, dropped: {}".format(g_h))
                    seen.add(g_h)
                left -= 1
    fasta = os.path.join(log_dir, "sample_identity.fa")
    logger.info("Save fasta fpath: {}".format(fasta))
    write_fasta_wrapped([(f"VH{args.fa_version}_{i}", "<unknown description>", s) for i, s in enumerate(human)], fasta)
    if args.structure:
        split_fasta_for_save(save_fpath, human)
    logger.info("Length did not equal list: {}".format([]))
    logger.info("Wrong idx: {}".format([]))
    return save_fpath


if __name__ == "__main__":
    _r = main()
    raise SystemExit(_r if isinstance(_r, int) else 0)      # an int is the exit code of a --gpus N relaunch
