"""Drop-in for ``antibody_scripts/sample.py`` (reference lines 379-588): same flags and defaults, same
log-dir naming, same ``sample_humanization_result.csv`` (header 'Specific,name,hseq,lseq,' then, per mouse
row, 'mouse,{name},{h},{l}' followed by 'humanization,{name}human_sample,{h},{l}') and ``sample_identity.fa``.

    python -m hudiff_amd.cli.sample --ckpt checkpoints/antibody/hudiffab.pt --data_fpath data.csv [...]
    torchrun --nproc-per-node 8 -m hudiff_amd.cli.sample ...          # rows sharded over the node's GPUs

Differences, all forced by what is absent offline: IMGT numbering uses anarci/abnumber when
importable and otherwise the built-in slotter (``--numbering``; ``--numbered_fpath`` accepts pre-numbered
residues); ``--sample_method inpaint`` grafts with abnumber when importable and otherwise takes pre-grafted chains
from ``--grafted_fpath`` (everything after the graft -- placement of the identity positions, mask, loc -- is
hudiff_amd.inputs.antibody_inpaint_row, pinned against the reference's batch_inpaint_input_element);
``--traditional_method`` (pure abnumber CDR grafting, no model: ``traditional_main``) needs abnumber and says so without it; the similarity
search scores identity over the aligned IMGT slots instead of an abnumber alignment; noise comes from the
library's counter-based generator keyed by (seed, global row, step), not torch's global mt19937 stream.
"""
from __future__ import annotations

import argparse
import os
import sys

import numpy as np

from .. import dist as D
from .. import inputs as I
from ..checkpoint import antibody_model_from_checkpoint, load_checkpoint
from ..model import model_selected
from ..sampler import Job, noise_in_reference_order, sample_jobs, seed_all
from .common import add_runtime_args, relaunch_if_asked, get_logger, get_new_log_dir, load_numbered, split_fasta_for_save, write_fasta_2line


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument("--ckpt", type=str, default="checkpoints/antibody/hudiffab.pt")
    p.add_argument("--ckpt_version", type=str, default="finetune", choices=["pretrain", "finetune"])
    p.add_argument("--data_fpath", type=str, default="humanization_pair_data_filter.csv")
    p.add_argument("--batch_size", type=int, default=1)
    p.add_argument("--sample_number", type=int, default=1)
    p.add_argument("--try_number", type=int, default=1)
    p.add_argument("--seed", type=int, default=2023)
    p.add_argument("--sample_order", type=str, default="shuffle")
    p.add_argument("--sample_method", type=str, default="FR", choices=["FR", "inpaint"])
    p.add_argument("--similarity_search", type=bool, default=True)     # type=bool as in the reference (:402)
    p.add_argument("--length_limit", type=str, default="not_equal")
    p.add_argument("--sample_type", type=str, default="pair")
    p.add_argument("--fa_version", type=str, default="v007")
    p.add_argument("--structure", type=eval, default=False)
    p.add_argument("--traditional_method", type=bool, default=False)
    p.add_argument("--back_mutation", type=bool, default=True)
    # additions
    p.add_argument("--numbered_fpath", type=str, default=None,
                   help="JSON-lines file with pre-numbered IMGT residues, one object per mouse row of --data_fpath")
    p.add_argument("--grafted_fpath", type=str, default=None,
                   help="--sample_method inpaint without abnumber: JSON-lines file, one object per mouse row, "
                        '{"h": {IMGT position: residue of the CDR-grafted VH}, "l": {...}, "identity_h": [positions kept], '
                        '"identity_l": [...], "l_chain": "K"|"L"} (what graft_chain, sample.py:209-226, returns)')
    p.add_argument("--numbering", choices=["auto", "anarci", "builtin"], default="auto",
                   help="IMGT numbering of raw sequences: anarci+abnumber as the reference (auto: when importable), "
                        "else the built-in slotter hudiff_amd/numbering.py")
    p.add_argument("--device_batch", type=int, default=256, help="rows per device launch")
    p.add_argument("--dropout", choices=["faithful", "off"], default="faithful",
                   help="faithful = the reference's inference-time dropout (active iff config.dropout > 0)")
    p.add_argument("--device", type=int, default=None)
    p.add_argument("--q_noise_fpath", type=str, default=None,
                   help="parity aid: .npz whose array 'q' [draws, batch_size, 22] is the torch.multinomial noise a run of the reference "
                        "recorded (input row by input row, step by step); replaces the library's counter-based noise for the first pass")
    add_runtime_args(p)
    return p


def sample_tag(args):
    """sample.py:426-432."""
    if "humab" in args.data_fpath:
        data_sample = "humab"
    elif "putative" in args.data_fpath:
        data_sample = "putative"
    else:
        data_sample = "lab"
    return f"{args.seed}_{args.sample_order}_{data_sample}_{args.ckpt_version}_search_simi_{args.similarity_search}"


def read_mouse_rows(fpath):
    """get_mouse_line (sample.py:314-317): rows with type == 'mouse', file order."""
    import pandas as pd
    df = pd.read_csv(fpath)
    return df[df["type"] == "mouse"]


def select_most_similar(parent_tokens, replica_tokens):
    """select_the_most_similarity_seq (sample.py:352-367): first replica with the highest mean of heavy and
    light identity to the parental sequence."""
    best, best_v = 0, -1.0
    H = I.T.H_LEN
    for r, row in enumerate(replica_tokens):
        v = 0.5 * (I.slot_identity(parent_tokens[:H], row[:H]) + I.slot_identity(parent_tokens[H:], row[H:]))
        if v > best_v:
            best, best_v = r, v
    return best


def traditional_main(args):
    """sample.py:539-576: CDR grafting only (abnumber), one 'humanization' row per mouse row and no 'mouse' rows; the log
    directory sits next to the data file."""
    if D.env_rank_world()[0] != 0:          # no model, nothing to shard: under torchrun only rank 0 grafts and writes
        return None
    data_sample = "humab" if "humab" in args.data_fpath else ("putative" if "putative" in args.data_fpath else "lab")
    # graft first: without abnumber this raises before any directory is created
    names, human_rows = [], []
    for line in read_mouse_rows(args.data_fpath).itertuples():
        human_rows.append(I.cdr_pair_grafting(line.h_seq, line.l_seq, back_mutation=bool(args.back_mutation)))
        names.append(line.name)
    log_dir = get_new_log_dir(root=os.path.dirname(args.data_fpath), prefix=f"{data_sample}_cdr_graft_back_mutation_{args.back_mutation}")
    save_fpath = os.path.join(log_dir, "sample_humanization_result.csv")
    get_logger("test", log_dir)
    with open(save_fpath, "a", encoding="UTF-8") as f:
        f.write("Specific,name,hseq,lseq,\n")
        for name, (g_h, g_l) in zip(names, human_rows):
            f.write(f"humanization,{name}human_sample,{g_h},{g_l}\n")
    records = []
    for i, (g_h, g_l) in enumerate(human_rows):
        records += [(args.fa_version + "human" + f"{i}", "VH", g_h), (args.fa_version + "human" + f"{i}", "VL", g_l)]
    write_fasta_2line(records, os.path.join(log_dir, "sample_identity.fa"))
    if args.structure:
        split_fasta_for_save(save_fpath, human_rows)
    return save_fpath


def main(argv=None):
    args = build_parser().parse_args(argv)
    rc = relaunch_if_asked(args, "hudiff_amd.cli.sample", argv)
    if rc is not None:
        return rc
    if args.traditional_method:
        return traditional_main(args)
    if args.sample_method == "inpaint" and not args.grafted_fpath and I.numbering_backend("auto") != "anarci":
        raise RuntimeError("--sample_method inpaint grafts the CDRs onto a human germline with abnumber (sample.py:209-226), "
                           "which is not installed here: supply the grafted chains with --grafted_fpath")
    rank, world, local_rank = D.env_rank_world()
    D.init_process_group()
    seed_all(args.seed)

    log_dir = logger = None
    if rank == 0:
        log_dir = get_new_log_dir(root=os.path.dirname(os.path.dirname(args.ckpt)), prefix=sample_tag(args))
        logger = get_logger("test", log_dir)

    ckpt = load_checkpoint(args.ckpt)
    config, state, finetune = antibody_model_from_checkpoint(ckpt, args.ckpt_version)
    model = model_selected(config, device=args.device if args.device is not None else local_rank, precision=args.precision)
    model.load_state_dict(state)
    model.eval()
    if rank == 0:
        logger.info(args.ckpt)
        logger.info(args.seed)
    n_region = config["model"]["n_region"] if "model" in config else config.model.n_region
    pad_region = 7 if n_region > 7 else 0                                   # sample.py:462-465

    mouse_df = read_mouse_rows(args.data_fpath)
    if rank == 0 and not args.numbered_fpath:
        logger.info("IMGT numbering backend: {}".format(I.numbering_backend(args.numbering)))
    numbered = load_numbered(args.numbered_fpath) if args.numbered_fpath else None
    if numbered is not None and len(numbered) != len(mouse_df.index):
        raise ValueError(f"{args.numbered_fpath}: {len(numbered)} rows for {len(mouse_df.index)} mouse rows")
    grafted = load_numbered(args.grafted_fpath) if (args.sample_method == "inpaint" and args.grafted_fpath) else None
    if grafted is not None and len(grafted) != len(mouse_df.index):
        raise ValueError(f"{args.grafted_fpath}: {len(grafted)} rows for {len(mouse_df.index)} mouse rows")
    jobs = []
    for idx, line in enumerate(mouse_df.itertuples()):
        if args.sample_method == "inpaint":
            # sample.py:486-489 batch_inpaint_input_element: the graft's identity positions stay, the rest of the
            # CDR-IMGT framework is sampled.  The similarity search still compares with the MOUSE chains (:524).
            if grafted is not None:
                g = grafted[idx]
                gh, ih, gl, il, l_type = g["h"], g["identity_h"], g["l"], g["identity_l"], g.get("l_chain", "K")
            else:
                gh, ih, _ = I.graft_chain(line.h_seq)
                gl, il, l_type = I.graft_chain(line.l_seq)
            tok, reg, chain, loc = I.antibody_inpaint_row(gh, gl, ih, il, l_type, pad_region=pad_region)
            if numbered is not None:
                h_dict, l_dict = numbered[idx]["h"], numbered[idx]["l"]
            else:
                h_dict, l_dict = I.number_sequence(line.h_seq, args.numbering)[0], I.number_sequence(line.l_seq, args.numbering)[0]
            parent = np.array(I._TK.seq2idx(I.slot_residues(h_dict, "H") + I.slot_residues(l_dict, "L")))
            if args.sample_order == "shuffle":
                np.random.shuffle(loc)
            jobs.append(Job(tokens=tok, region=reg, loc=loc, chain=chain, name=str(line.name),
                            parent={"h": line.h_seq, "l": line.l_seq, "tokens": parent}))
            continue
        if numbered is None:
            h_dict, h_type = I.number_sequence(line.h_seq, args.numbering)
            l_dict, l_type = I.number_sequence(line.l_seq, args.numbering)
        else:
            h_dict, l_dict, l_type = numbered[idx]["h"], numbered[idx]["l"], numbered[idx].get("l_chain", "K")
        tok, reg, chain, loc = I.antibody_row(h_dict, l_dict, l_type, finetune=finetune, pad_region=pad_region)
        parent = np.array(I._TK.seq2idx(I.slot_residues(h_dict, "H") + I.slot_residues(l_dict, "L")))
        if args.sample_order == "shuffle":
            np.random.shuffle(loc)                                           # sample.py:497-498
        jobs.append(Job(tokens=tok, region=reg, loc=loc, chain=chain, name=str(line.name),
                        parent={"h": line.h_seq, "l": line.l_seq, "tokens": parent}))

    # sample.py:499-538: with similarity search one pass gives the single output row; without it the loop
    # re-sweeps until sample_number rows have been written (batch_size rows per pass)
    passes = 1 if args.similarity_search else max(1, -(-args.sample_number // args.batch_size))
    if args.sample_number <= 0 or args.try_number <= 0:            # `while sample_number > 0 and try_num > 0` never runs
        args.sample_number = 0
        result = np.zeros((len(jobs), 0, args.batch_size, model.max_len), np.int32)
    else:
        q_noise = None
        if args.q_noise_fpath:
            if passes != 1:
                raise ValueError("--q_noise_fpath holds the noise of ONE pass over the input rows")
            q_noise = noise_in_reference_order(np.load(args.q_noise_fpath)["q"], jobs, args.batch_size)
        result = sample_jobs(model, jobs, args.batch_size, args.seed, passes=passes, device_batch=args.device_batch,
                             dropout=args.dropout, q_noise=q_noise)
    if rank != 0:
        return None

    save_fpath = os.path.join(log_dir, "sample_humanization_result.csv")
    human_rows = []
    with open(save_fpath, "a", encoding="UTF-8") as f:
        f.write("Specific,name,hseq,lseq,\n")
        for j, job in enumerate(jobs):
            f.write(f"mouse,{job.name},{job.parent['h']},{job.parent['l']}\n")
            sample_name = str(job.name) + "human_sample"
            if args.similarity_search:
                if args.sample_number <= 0:
                    continue
                r = select_most_similar(job.parent["tokens"], result[j, 0])
                g_h, g_l = I.untokenize_antibody(result[j, 0, r])
                f.write(f"humanization,{sample_name},{g_h},{g_l}\n")
                human_rows.append((g_h, g_l))
            else:
                left = args.sample_number
                for p in range(result.shape[1]):
                    for r in range(args.batch_size):
                        if left == 0:
                            break
                        g_h, g_l = I.untokenize_antibody(result[j, p, r])
                        f.write(f"humanization,{sample_name},{g_h},{g_l}\n")
                        human_rows.append((g_h, g_l))
                        left -= 1
    logger.info("Length did not equal list: {}".format([]))
    logger.info("Wrong idx: {}".format([]))
    # sample_identity.fa (trans_to_chain + save_pairs, sample.py:34-54): '{fa_version}human{i}' VH / VL pairs
    records = []
    for i, (g_h, g_l) in enumerate(human_rows):
        name = args.fa_version + "human" + f"{i}"
        records += [(name, "VH", g_h), (name, "VL", g_l)]
    write_fasta_2line(records, os.path.join(log_dir, "sample_identity.fa"))
    if args.structure:
        split_fasta_for_save(save_fpath, human_rows)
    return save_fpath


if __name__ == "__main__":
    _r = main()
    raise SystemExit(_r if isinstance(_r, int) else 0)      # an int is the exit code of a --gpus N relaunch
