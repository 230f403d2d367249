"""Drop-in for ``nanobody_scripts/nanosample.py`` (reference lines 195-368): same flags and defaults, same
log-dir naming, ``sample_humanization_result.csv`` with header 'Specific,name,hseq,' and per input row
'nano,{idx},{vhh}' then 'humanization,{idx}human_sample,{seq}', and ``sample_identity.fa``.

    python -m hudiff_amd.cli.nanosample --ckpt hudiffnb.pt --model finetune_vh --inpaint_sample True ...

See hudiff_amd/cli/sample.py for the (forced) differences; additionally the reference re-samples when
``abnumber.Chain(seq)`` fails to parse (nanosample.py:331-353) -- without abnumber the built-in slotter's
domain check (hudiff_amd/numbering.py ``is_variable_domain``) takes its place.
"""
from __future__ import annotations

import argparse
import os

import numpy as np

from .. import dist as D
from .. import inputs as I
from ..checkpoint import load_checkpoint, nanobody_model_from_checkpoint
from ..model import NanoAntiTFNet
from ..sampler import Job, noise_in_reference_order, sample_jobs_with_retry, seed_all
from .common import add_runtime_args, relaunch_if_asked, get_logger, get_new_log_dir, load_numbered, split_fasta_for_save, write_fasta_wrapped


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument("--ckpt", type=str, default="nanofinetune.pt")
    p.add_argument("--data_fpath", type=str, default="abnativ_select_vhh.csv")
    p.add_argument("--batch_size", type=int, default=1)
    p.add_argument("--sample_number", type=int, default=1)
    p.add_argument("--try_number", type=int, default=10)
    p.add_argument("--seed", type=int, default=2023)
    p.add_argument("--sample_order", type=str, default="shuffle")
    p.add_argument("--sample_method", type=str, default="gen", choices=["gen", "rl_gen"])
    p.add_argument("--length_limit", type=str, default="not_equal")
    p.add_argument("--model", type=str, default="finetune_vh", choices=["pretrain", "finetune_vh"])
    p.add_argument("--fa_version", type=str, default="v_nano")
    p.add_argument("--inpaint_sample", type=eval, default=False)
    p.add_argument("--structure", type=eval, default=False)
    # additions
    p.add_argument("--numbered_fpath", type=str, default=None)
    p.add_argument("--numbering", choices=["auto", "anarci", "builtin"], default="auto")
    p.add_argument("--device_batch", type=int, default=256)
    p.add_argument("--dropout", choices=["faithful", "off"], default="faithful")
    p.add_argument("--device", type=int, default=None)
    p.add_argument("--q_noise_fpath", type=str, default=None,
                   help="parity aid: .npz whose array 'q' [draws, batch_size, 22] is the torch.multinomial noise a run of the reference "
                        "recorded (input row by input row, step by step); replaces the library's counter-based noise for the first sweep")
    add_runtime_args(p)
    return p


def sample_tag(args):
    """nanosample.py:237-243."""
    data_sample = "abnativ_select" if "filter" in args.data_fpath else "nanobert"
    return f"{args.seed}_{args.sample_order}_{data_sample}_{args.sample_method}_{args.length_limit}_{args.model}"


def chain_is_valid(seq):
    """nanosample.py:342 -- ``Chain(g_h, scheme='imgt')`` must parse; without abnumber the built-in slotter must
    find a complete heavy domain (both cysteines, Trp41, J motif)."""
    try:
        from abnumber import Chain
    except ImportError:
        from ..numbering import is_variable_domain
        return is_variable_domain(seq, "H")
    try:
        Chain(seq, scheme="imgt")
        return True
    except Exception:
        return False


def main(argv=None):
    args = build_parser().parse_args(argv)
    rc = relaunch_if_asked(args, "hudiff_amd.cli.nanosample", argv)
    if rc is not None:
        return rc
    print(args.inpaint_sample)
    rank, world, local_rank = D.env_rank_world()
    D.init_process_group()
    seed_all(args.seed)
    log_dir = logger = None
    if rank == 0:
        log_dir = get_new_log_dir(root=os.path.dirname(os.path.dirname(args.ckpt)), prefix=sample_tag(args))
        logger = get_logger("test", log_dir)
    ckpt = load_checkpoint(args.ckpt)
    _, params, state = nanobody_model_from_checkpoint(ckpt, args.model)
    model = NanoAntiTFNet(**params, device=args.device if args.device is not None else local_rank, precision=args.precision)
    model.load_state_dict(state)
    model.eval()
    if rank == 0:
        logger.info(args.ckpt)
        logger.info(args.seed)

    import pandas as pd
    nano_df = pd.read_csv(args.data_fpath)
    numbered = load_numbered(args.numbered_fpath) if args.numbered_fpath else None
    if numbered is not None and len(numbered) != len(nano_df.index):
        raise ValueError(f"{args.numbered_fpath}: {len(numbered)} rows for {len(nano_df.index)} input rows")
    jobs = []
    for idx, line in enumerate(nano_df.itertuples()):
        h_dict = numbered[idx]["h"] if numbered is not None else I.number_sequence(line.vhhseq, args.numbering)[0]
        tok, reg, loc = I.nanobody_row(h_dict, inpaint_sample=args.inpaint_sample)
        if args.sample_order == "shuffle":
            np.random.shuffle(loc)                                            # nanosample.py:314-315
        jobs.append(Job(tokens=tok, region=reg, loc=loc, name=str(idx), parent={"h": line.vhhseq}))

    # nanosample.py:316-353: accept / re-sweep loop (a sample must parse as a heavy domain; the last try is
    # written regardless), run for all sequences at once
    def rejected(job, row):
        if rank == 0:
            logger.info(I.untokenize_nanobody(row))
            logger.info("Need to re sample again.")
    q_noise = noise_in_reference_order(np.load(args.q_noise_fpath)["q"], jobs, args.batch_size) if args.q_noise_fpath else None
    written = sample_jobs_with_retry(model, jobs, args.batch_size, args.seed, want=args.sample_number,
                                     tries=args.try_number, accept=lambda row: chain_is_valid(I.untokenize_nanobody(row)),
                                     device_batch=args.device_batch, dropout=args.dropout, log=rejected, q_noise=q_noise)
    if rank != 0:
        return None
    save_fpath = os.path.join(log_dir, "sample_humanization_result.csv")
    human = []
    with open(save_fpath, "a", encoding="UTF-8") as f:
        f.write("Specific,name,hseq,\n")
        for j, job in enumerate(jobs):
            f.write(f"nano,{job.name},{job.parent['h']}\n")
            for row in written[j]:
                g_h = I.untokenize_nanobody(row)
                logger.info(g_h)
                f.write(f"humanization,{job.name}human_sample,{g_h}\n")
                human.append(g_h)
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
