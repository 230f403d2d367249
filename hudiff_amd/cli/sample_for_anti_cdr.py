"""Drop-in for ``antibody_scripts/sample_for_anti_cdr.py`` (reference lines 73-221): humanize ONE antibody given as a
PDB-style FASTA (records whose description holds 'heavy chain' / 'light chain') or as ``--heavy_seq`` / ``--light_seq``.

    python -m hudiff_amd.cli.sample_for_anti_cdr --ckpt hudiffab.pt --anti_complex_fasta fasta_file/7k9i.fasta

Same flags, log-dir naming ('{pdb}_{order}_{type}_{time}' under --log_dirpath) and CSV ('Specific,name,hseq,lseq,' /
'mouse,{pdb},{VH},{VL}' / 'humanization,{pdb}human_sample,{h},{l}'; duplicates are skipped but still count towards
--sample_number, :199-213).  The variable domains are cut out with the reference's stack (``abnumber.Chain(seq).seq``)
when it is importable, else with the built-in slotter (``--numbering``).  The reference leaves the RNG unseeded
(``seed_all`` is commented out, :118); here ``--seed`` keys both the visiting order and the noise, so runs repeat.
"""
from __future__ import annotations

import argparse
import os

import numpy as np

from .. import dist as D
from .. import inputs as I
from ..checkpoint import antibody_model_from_checkpoint, load_checkpoint
from ..model import model_selected
from ..sampler import Job, sample_jobs, seed_all
from .common import add_runtime_args, relaunch_if_asked, get_logger, get_new_log_dir, read_fasta


def build_parser():
    p = argparse.ArgumentParser(description="This program is designed to humanize non-human antibodies.")
    p.add_argument("--ckpt", type=str, default="checkpoints/antibody/hudiffab.pt")
    p.add_argument("--anti_complex_fasta", type=str, default="fasta_file/7k9i.fasta")
    p.add_argument("--heavy_seq", type=str)
    p.add_argument("--light_seq", type=str)
    p.add_argument("--log_dirpath", type=str, default="antibody_sample_log/")
    p.add_argument("--batch_size", type=int, default=10)
    p.add_argument("--seed", type=int, default=42)
    p.add_argument("--sample_number", type=int, default=10)
    p.add_argument("--sample_order", type=str, default="shuffle")
    p.add_argument("--sample_type", type=str, default="pair")
    p.add_argument("--finetune", type=str, default=True)          # any non-empty string is truthy, as in the reference
    # additions
    p.add_argument("--numbering", choices=["auto", "anarci", "builtin"], default="auto")
    p.add_argument("--dropout", choices=["faithful", "off"], default="faithful")
    p.add_argument("--device", type=int, default=None)
    add_runtime_args(p)
    return p


def get_h_l_seq_from_fasta(fpath):
    """:53-70 -- last record whose description mentions 'heavy chain' / 'light chain'."""
    heavy = light = None
    for desc, seq in read_fasta(fpath):
        if "heavy chain" in desc:
            heavy = seq
        elif "light chain" in desc:
            light = seq
    assert heavy is not None and light is not None, "Reading the fasta has problem."
    return heavy, light


def variable_domain(seq, numbering):
    """``Chain(seq, scheme='imgt').seq`` (:154-155)."""
    if I.numbering_backend(numbering) == "anarci":
        from abnumber import Chain
        return Chain(seq, scheme="imgt").seq
    from ..numbering import domain_sequence
    return domain_sequence(seq)


def main(argv=None):
    args = build_parser().parse_args(argv)
    rc = relaunch_if_asked(args, "hudiff_amd.cli.sample_for_anti_cdr", argv)
    if rc is not None:
        return rc
    rank, world, local_rank = D.env_rank_world()
    D.init_process_group()
    seed_all(args.seed)
    if args.anti_complex_fasta is not None and not (args.heavy_seq and args.light_seq):
        mouse_heavy, mouse_light = get_h_l_seq_from_fasta(args.anti_complex_fasta)
        pdb_name = os.path.basename(args.anti_complex_fasta).split(".")[0]
    else:
        mouse_heavy, mouse_light, pdb_name = args.heavy_seq, args.light_seq, "Unkown"
    log_dir = logger = None
    if rank == 0:
        root = args.log_dirpath if args.log_dirpath is not None else os.path.dirname(args.anti_complex_fasta)
        log_dir = get_new_log_dir(root=root, prefix=f"{pdb_name}_{args.sample_order}_{args.sample_type}")
        logger = get_logger("test", log_dir)
    ckpt = load_checkpoint(args.ckpt)
    config, state, _ = antibody_model_from_checkpoint(ckpt, "finetune")        # always ckpt['pretrain_config'] (:135)
    model = model_selected(config, device=args.device if args.device is not None else local_rank, precision=args.precision)
    model.load_state_dict(state)
    model.eval()
    if rank == 0:
        logger.info(args.ckpt)
        logger.info(args.seed)

    mouse_aa_h = variable_domain(mouse_heavy, args.numbering)
    mouse_aa_l = variable_domain(mouse_light, args.numbering)
    h_dict, _ = I.number_sequence(mouse_aa_h, args.numbering)
    l_dict, l_type = I.number_sequence(mouse_aa_l, args.numbering)
    tok, reg, chain, loc = I.antibody_row(h_dict, l_dict, l_type, finetune=bool(args.finetune), pad_region=0)
    if args.sample_order == "shuffle":
        np.random.shuffle(loc)
    job = Job(tokens=tok, region=reg, loc=loc, chain=chain, name=pdb_name)
    passes = max(0, -(-args.sample_number // args.batch_size))                 # every replica looked at counts (:212)
    result = sample_jobs(model, [job], args.batch_size, args.seed, passes=max(passes, 1), dropout=args.dropout)
    if rank != 0:
        return None
    save_fpath = os.path.join(log_dir, "sample_humanization_result.csv")
    seen, left = set(), args.sample_number
    with open(save_fpath, "a", encoding="UTF-8") as f:
        f.write("Specific,name,hseq,lseq,\n")
        f.write(f"mouse,{pdb_name},{mouse_aa_h},{mouse_aa_l}\n")
        for p in range(passes):
            for r in range(args.batch_size):
                if left == 0:
                    break
                g_h, g_l = I.untokenize_antibody(result[0, p, r])
                if (g_h, g_l) not in seen:
                    seen.add((g_h, g_l))
                    f.write(f"humanization,{pdb_name}human_sample,{g_h},{g_l}\n")
                left -= 1
                logger.info("Already Sample number {}".format(args.sample_number - left))
                logger.info("Sample Heavy Chain Seq: {}".format(g_h))
                logger.info("Sample Light Chain Seq: {}".format(g_l))
    logger.info("Length did not equal list: {}".format([]))
    logger.info("Wrong idx: {}".format([]))
    return save_fpath


if __name__ == "__main__":
    _r = main()
    raise SystemExit(_r if isinstance(_r, int) else 0)      # an int is the exit code of a --gpus N relaunch
