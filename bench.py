#!/usr/bin/env python
"""bench.py -- humanized sequences / second (full T-step sample), the BASELINE.json metric.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one complete T-step humanization sample of one batch of B = 256 independent rows per GPU
(BASELINE.json configs[1]: HuDiff-Ab on HuAb348, batch 256, 1 x MI355X): up to 154 denoiser forwards of the
39.8 M-parameter AntiTFNet over 291 slots + the exponential-race resampling, with inference-time dropout as
the reference runs it.  Rows are the 348 HuAb348 mouse pairs (pre-slotted integer fixture hudiff_amd/data/real_rows.npz,
scripts/make_real_rows.py; global row g = pair g % 348, replica g // 348, ragged T = 141..154) -- `--data synthetic`
gives the HuAb348-shaped random rows of round 1 instead.  Weights are seeded random weights of the exact production
architecture (no released checkpoint offline).  Rows shard across GPUs with no data-path collective; one RCCL gather
of the final int32 tokens ends the job (weak scaling: 256 rows per GPU).

Timed region: inputs already resident in HBM (hd_sample_begin uploaded them); K x [restore tokens
device-side, re-key noise, run all T steps]; bracketed by barrier + device synchronise on both sides,
max over ranks.  One JSON line is printed by rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # RCCL needs dmabuf IPC on this driver stack

PEAK_F32_MATRIX_TFLOPS = 157.3     # /opt/skills/guides/MI355X_MICROARCH.md "Peak FP32 (matrix)"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=256, help="rows per GPU")
    ap.add_argument("--kind", choices=["ab", "nb"], default="ab")
    ap.add_argument("--mode", default=None, help="ab: finetune|pretrain, nb: plain|inpaint")
    ap.add_argument("--dropout", choices=["faithful", "off"], default="faithful")
    ap.add_argument("--max-t", type=int, default=0, help="truncate every row to this many denoiser steps "
                    "(profiling aid; the JSON line is then marked truncated and is NOT the metric)")
    ap.add_argument("--data", choices=["auto", "real", "synthetic"], default="auto",
                    help="real = rows of the reference's evaluation set (HuAb348 / VHH, hudiff_amd/data/real_rows.npz); "
                         "auto = real when the fixture is present")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-rows", type=int, default=16, help="rows of the 'best batch' CPU leg (B = 1 is always timed too)")
    ap.add_argument("--cpu-steps", type=int, default=16)
    ap.add_argument("--traffic", choices=["auto", "live", "file", "off"], default="auto",
                    help="roofline.traffic: live = two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over a 4-step run of "
                         "this script in a subprocess after the timed region; file = profiles/r02/pmc_traffic.json")
    ap.add_argument("--cpu-impl", choices=["auto", "torch", "numpy"], default="auto",
                    help="CPU baseline on the oracle's PyTorch-CPU variant (auto: when torch is importable) or on numpy")
    ap.add_argument("--no-split-line", action="store_true",
                    help="skip the extra `split_precision` leg (same workload on the HUDIFF_X3=1 kernels, reported beside the f32 metric)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--loop-graph", action="store_true", help="the whole T-step loop of a lane as ONE hipGraph (HD_LOOP_GRAPH) "
                    "instead of T replays of the step graph")
    ap.add_argument("--lanes", type=int, default=2, choices=[1, 2],
                    help="2 = the batch runs as two concurrent half-batches on two streams (library default)")
    return ap.parse_args()


def physical_cores():
    """Physical cores of the host ((physical id, core id) pairs of /proc/cpuinfo), logical CPUs."""
    logical = os.cpu_count() or 1
    try:
        seen, phys, core = set(), None, None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                phys = line.split(":")[1].strip()
            elif line.startswith("core id"):
                core = line.split(":")[1].strip()
            elif not line.strip():
                if phys is not None and core is not None:
                    seen.add((phys, core))
                phys = core = None
        return (len(seen) or logical), logical
    except OSError:
        return logical, logical


class ClockPowerSampler:
    """Shader clock and package power of the GPU this rank runs on, read from the amdgpu hwmon files (freq1_input in Hz,
    power1_input in uW) every 100 ms by a thread while the timed region runs.  Both paths sit on the package power limit
    (DESIGN.md section 9), so the sustained clock -- not the 2.4 GHz of the peak figure -- is what the matrix pipe ran at."""

    def __init__(self, device):
        import threading
        self.dir, self.f, self.p, self._stop, self._t = None, [], [], threading.Event(), None
        try:
            import ctypes, glob
            hip = ctypes.CDLL("libamdhip64.so")
            buf = ctypes.create_string_buffer(64)
            if hip.hipDeviceGetPCIBusId(buf, 64, int(device)) == 0:
                d = glob.glob(f"/sys/bus/pci/devices/{buf.value.decode().lower()}/hwmon/hwmon*")
                if d and os.path.exists(os.path.join(d[0], "freq1_input")) and os.path.exists(os.path.join(d[0], "power1_input")):
                    self.dir = d[0]
        except (OSError, AttributeError):
            pass
        if self.dir:
            self._t = threading.Thread(target=self._run, daemon=True)

    def _read(self, name):
        with open(os.path.join(self.dir, name)) as fh:
            return float(fh.read().strip())

    def _run(self):
        while not self._stop.wait(0.1):
            try:
                self.f.append(self._read("freq1_input") * 1e-6)
                self.p.append(self._read("power1_input") * 1e-6)
            except (OSError, ValueError):
                return

    def start(self):
        if self._t:
            self._t.start()
        return self

    def stop(self):
        if not self._t:
            return None
        self._stop.set()
        self._t.join()
        busy = [(f, p) for f, p in zip(self.f, self.p) if p > 500.0] or list(zip(self.f, self.p))
        if not busy:
            return None
        med = lambda v: float(sorted(v)[len(v) // 2])
        out = {"sclk_mhz_median": round(med([f for f, _ in busy]), 1), "power_w_median": round(med([p for _, p in busy]), 1),
               "samples": len(busy)}
        try:
            out["power_cap_w"] = round(self._read("power1_cap") * 1e-6, 1)
        except (OSError, ValueError):
            pass
        return out


def make_batch(kind, B, mode, row0, data):
    from hudiff_amd import evalsets as E
    from hudiff_amd import synthetic as S
    if data == "real" or (data == "auto" and E.available()):
        return E.eval_batch("huab348" if kind == "ab" else "vhh", B, mode=mode, row0=row0, seed=2023), True
    return S.synthetic_batch(kind, B, seed=2023, mode=mode, row0=row0), False


def cpu_baseline(kind, cfg, sd, mode, rows, steps, mean_T, impl="auto", data="auto"):
    """The oracle (CPU port of the reference algorithm) on the host cores, bounded sample.  SURVEY.md §8d: the build's
    CPU restatement, on PyTorch-CPU kernels where torch is present (oracle/hudiff_oracle_torch.py), else on numpy; timed
    at B = 1 (the reference CLI's default --batch_size) and at a larger batch; `value` is the better of the two."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import hudiff_oracle as ho
    batch, _ = make_batch(kind, rows, mode, 0, data)
    net, name, threads = None, "numpy+OpenBLAS", None
    if impl in ("auto", "torch"):
        try:
            import torch
            import hudiff_oracle_torch as hot
            net, name, threads = hot.TorchOracleNet(kind, cfg, sd), "PyTorch-CPU kernels", torch.get_num_threads()
        except ImportError:
            if impl == "torch":
                raise
    if net is None:
        net = ho.OracleNet(kind, cfg, sd)
    phys, logical = physical_cores()

    def run(n_rows, n_steps):
        ch = None if batch["chain"] is None else np.concatenate([batch["chain"][:n_rows], batch["chain"][rows:rows + n_rows]])
        t0 = time.perf_counter()
        ho.sample(net, batch["tokens"][:n_rows], batch["region"][:n_rows], ch, batch["order"][:n_rows],
                  np.minimum(batch["T"][:n_rows], n_steps), seed=1, dropout_mode="philox")
        return time.perf_counter() - t0

    run(rows, 1)                                                               # warm-up (threads, caches)
    if threads is not None:
        # many-core hosts oversubscribe these matrix sizes (256 logical CPUs: 0.035 sequences/s on 128 threads, 0.145 on 16):
        # one step per candidate thread count -- up to ALL physical cores -- the timed sample runs on the fastest
        import torch
        best = None
        for n in sorted({8, 16, 32, min(64, phys), phys}):
            torch.set_num_threads(n)
            dtn = run(rows, 1)
            if best is None or dtn < best[0]:
                best = (dtn, n)
        threads = best[1]
        torch.set_num_threads(threads)
        name += f" (fastest of 8/16/32/64/{phys} threads)"
    dt_b = run(rows, steps)
    steps1 = max(2, steps // 2)
    run(1, 1)
    dt_1 = run(1, steps1)
    if threads is None:
        try:
            from threadpoolctl import threadpool_info
            threads = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
        except Exception:
            threads = os.cpu_count() or 1
    v_b = rows / (dt_b / steps * mean_T)
    v_1 = 1.0 / (dt_1 / steps1 * mean_T)
    return {"value": max(v_b, v_1), "unit": "sequences/s", "cores": int(threads), "kind": "port",
            "physical_cores": phys, "logical_cpus": logical, "value_batch_1": v_1, f"value_batch_{rows}": v_b,
            "sample": f"oracle restatement of the reference loop on {name}; host has {phys} physical cores / {logical} logical "
                      f"CPUs, {threads} threads used; {rows} rows x {steps} denoiser steps in {dt_b:.1f} s and 1 row x {steps1} "
                      f"steps in {dt_1:.1f} s (B = 1 is the reference CLI's default), philox dropout, extrapolated to the "
                      f"mean T = {mean_T:.1f} steps per sequence"}


def split_precision_leg(args, kind, cfg, sd, batch, T, Tmax, rank, local_rank, flops_row, ref_logits, ref_rows, env_name="HUDIFF_X3"):
    """The same workload on the split-precision GEMM kernels (HUDIFF_X3=1: three fp16 MFMAs with fp32 accumulation per fp32
    product, hd_kernels.hip.h gemm_x3_k).  Reported BESIDE the metric, never as it: the f32 line above is the product path."""
    import hudiff_amd
    B = args.batch
    prev = os.environ.get(env_name)
    os.environ[env_name] = "1"
    try:
        model = (hudiff_amd.AntiTFNet if kind == "ab" else hudiff_amd.NanoAntiTFNet)(**cfg, device=local_rank)
        model.load_state_dict(sd)
    finally:
        if prev is None:
            os.environ.pop(env_name, None)
        else:
            os.environ[env_name] = prev
    ch = None if batch["chain"] is None else np.concatenate([batch["chain"][:ref_rows], batch["chain"][B:B + ref_rows]])
    logits = model(batch["tokens"][:ref_rows], batch["region"][:ref_rows], ch, dropout=args.dropout, seed=2023, row0=rank * B, step=0)
    dmax = float(np.abs(logits - ref_logits).max())
    model.sample_begin(batch["tokens"], batch["region"], batch["chain"], batch["order"], T, seed=2023,
                       row0=rank * B, dropout=args.dropout, graph=(False if args.no_graph else "loop" if args.loop_graph else True), lanes=args.lanes)
    gpu_ms = 0.0
    watch = None
    steps = min(args.steps, 3)         # beside the metric: bounded so that a long --steps run spends its time on the f32 line
    first = args.steps - steps         # the last timed sample carries the same noise key as the f32 line's last one
    for i in range(first - min(args.warmup, 1), args.steps):
        model.sample_restart(2023 + 7919 * i)
        if i == first:
            model.sync()
            watch = ClockPowerSampler(local_rank).start()
            t0 = time.perf_counter()
        model.sample_run(0, Tmax)
        if i >= first:
            model.sync()
            gpu_ms += model.last_run_ms()[0]
    elapsed = time.perf_counter() - t0
    clock_power = watch.stop() if watch else None
    tokens = model.sample_end()
    model.close()
    tf = float(T.sum()) * flops_row * steps / (gpu_ms * 1e-3) / 1e12
    return tokens, {"value": round(B * steps / elapsed, 4), "unit": "sequences/s", "steps": steps, "ms_per_step": round(1e3 * elapsed / steps, 3),
                    "dtype": "fp32 operands split as fp16 hi + fp16 lo, 3 x v_mfma_f32_32x32x16_f16, fp32 accumulate",
                    "algorithmic_tflops": round(tf, 3), "avg_launch_ms": round(gpu_ms / (steps * Tmax), 4),
                    # three fp16 MFMAs per fp32 product: matrix-pipe rate against the dense fp16 peak of the guide (2.4 GHz;
                    # the sample sustains ~2.0 GHz at the 1.4 kW package limit)
                    "roofline": {"bound": "mfma", "achieved": round(3 * tf, 2), "peak": 2500.0,
                                 "unit": "TFLOP/s (fp16 MFMA, 3 per product)", "frac": round(3 * tf / 2500.0, 4)},
                    "clock_power": clock_power, "max_abs_dlogit_vs_f32_path": dmax, "dlogit_rows": ref_rows,
                    "note": "HUDIFF_X3=1 prototype (DESIGN.md section 9): Q|K|V, out-projection, FF and tap GEMMs; the remaining "
                            "kernels are the f32 ones.  Not the metric."}


def live_traffic(args, kind, mode, n_steps=4):
    """HBM-side bytes per denoiser step from rocprofv3 PMC passes of THIS script (FETCH_SIZE and WRITE_SIZE in separate
    passes, MI355X_MICROARCH.md "HBM" / "rocprofv3 PMC slots"; FETCH_SIZE doubled: gfx950 tallies 128-B requests at 64 B;
    unit KB).  Counted from the first token gather on (the once-per-batch static branch is excluded)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if shutil.which("rocprofv3") is None:
        return None
    tot = {}
    tmp = tempfile.mkdtemp(prefix="hudiff_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, counter)
            cmd = ["rocprofv3", "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", out, "-o", "p", "--",
                   sys.executable, os.path.abspath(__file__), "--kind", kind, "--mode", mode, "--batch", str(args.batch),
                   "--dropout", args.dropout, "--data", args.data, "--steps", "1", "--warmup", "0", "--max-t", str(n_steps),
                   "--no-cpu-baseline", "--lanes", "1", "--traffic", "off"]
            subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=300, check=True)
            files = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)
            rows = [r for f in files for r in csv.DictReader(open(f))]
            rows.sort(key=lambda r: int(r["Dispatch_Id"]))
            started, acc = False, 0.0
            for r in rows:
                started = started or "embed_tokens_k" in r["Kernel_Name"]
                if started and r["Counter_Name"] == counter:
                    acc += float(r["Counter_Value"])
            tot[counter] = acc
    except Exception as e:                 # profiler absent / refused: the bench line must still appear
        sys.stderr.write(f"[bench] live PMC traffic pass failed: {e!r}\n")
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    rd = 2.0 * tot["FETCH_SIZE"] * 1024.0 / n_steps
    wr = tot["WRITE_SIZE"] * 1024.0 / n_steps
    return {"traffic": rd + wr, "read_bytes": rd, "write_bytes": wr,
            "note": f"bytes per denoiser step, live: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over a "
                    f"{n_steps}-step one-lane run of this command; FETCH_SIZE x 2 (gfx950), KB -> bytes; L2<->fabric side, "
                    "Infinity-Cache hits included"}


def relaunch_one_rank_per_gpu(args):
    """`python bench.py --gpus N` without a launcher around it (the driver's single-node command shape): start N ranks of
    this very command under torch.distributed.run -- one process per GPU, rendezvous on 127.0.0.1 -- and hand back its exit
    code.  Rank 0 of that job prints the JSON line on the inherited stdout."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(relaunch_one_rank_per_gpu(args))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        # never print a line whose n_gpus is not what was asked for
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    dist = None
    # Debug aid for single-GPU boxes: HUDIFF_BENCH_SHARE_GPU=1 runs every rank on device 0 with a gloo (CPU)
    # gather, so that the N > 1 control flow can be exercised without N GPUs.  Never set by the driver.
    share_gpu = os.environ.get("HUDIFF_BENCH_SHARE_GPU") == "1"
    if share_gpu:
        local_rank = 0
    # HUDIFF_BENCH_FORCE_PG=1: form the RCCL process group even for one rank (a world-size-1 nccl group is legal), so that
    # the communicator init, the device-tensor gather and the all-reduce below run on a one-GPU box (tests/test_gpu_dist.py)
    force_pg = os.environ.get("HUDIFF_BENCH_FORCE_PG") == "1"
    if world > 1 or force_pg:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        if force_pg and world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if "MASTER_PORT" not in os.environ:
                import socket
                with socket.socket() as s:
                    s.bind(("127.0.0.1", 0))
                    os.environ["MASTER_PORT"] = str(s.getsockname()[1])
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if share_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))   # nccl == RCCL on ROCm
    n_gpus = world
    coll_dev = "cpu" if share_gpu else "cuda"
    formed = {"world_size": 1, "backend": None}
    if dist is not None:
        # what the process group actually formed (not what the environment asked for)
        formed = {"world_size": int(dist.get_world_size()), "backend": str(dist.get_backend())}
        assert formed["world_size"] == world, (formed, world)

    import hudiff_amd
    from hudiff_amd import synthetic as S

    kind = args.kind
    cfg = dict(S.AB_CONFIG if kind == "ab" else S.NB_CONFIG)
    mode = args.mode or ("finetune" if kind == "ab" else "plain")
    sd = S.random_state_dict(kind, cfg, seed=0)
    B = args.batch
    batch, real = make_batch(kind, B, mode, rank * B, args.data)
    T = batch["T"].copy()
    if args.max_t > 0:
        T = np.minimum(T, args.max_t)
    Tmax = int(T.max())

    model = (hudiff_amd.AntiTFNet if kind == "ab" else hudiff_amd.NanoAntiTFNet)(**cfg, device=local_rank)
    model.load_state_dict(sd)
    flops_row = model.flops_per_row_forward()            # canonical / algorithmic
    flops_row_exec = model.flops_per_row_sample_step()   # executed: last attention block pruned to the visited row

    def barrier():
        model.sync()
        if dist is not None:
            import torch
            torch.cuda.synchronize()
            dist.barrier()

    t_up0 = time.perf_counter()
    model.sample_begin(batch["tokens"], batch["region"], batch["chain"], batch["order"], T, seed=2023,
                       row0=rank * B, dropout=args.dropout, graph=(False if args.no_graph else "loop" if args.loop_graph else True), lanes=args.lanes)
    upload_s = time.perf_counter() - t_up0
    gpu_ms = 0.0

    def one_sample(i, timed):
        nonlocal gpu_ms
        model.sample_restart(2023 + 7919 * i)
        model.sample_run(0, Tmax)
        if timed:
            model.sync()
            gpu_ms += model.last_run_ms()[0]

    for i in range(args.warmup):
        one_sample(-1 - i, False)
    barrier()
    watch = ClockPowerSampler(local_rank).start() if rank == 0 else None
    t0 = time.perf_counter()
    for i in range(args.steps):
        one_sample(i, True)
    barrier()
    elapsed = time.perf_counter() - t0
    clock_power = watch.stop() if watch else None
    tokens = model.sample_end()
    split = None
    if rank == 0 and world == 1 and args.max_t == 0 and not args.no_split_line and os.environ.get("HUDIFF_X3", "0") in ("", "0"):
        ref_rows = min(B, 64)          # enough activation rows (>= 8192) for the big-launch kernels on both models
        ch = None if batch["chain"] is None else np.concatenate([batch["chain"][:ref_rows], batch["chain"][B:B + ref_rows]])
        ref_logits = model(batch["tokens"][:ref_rows], batch["region"][:ref_rows], ch, dropout=args.dropout, seed=2023,
                           row0=rank * B, step=0)
        x3_tokens, split = split_precision_leg(args, kind, cfg, sd, batch, T, Tmax, rank, local_rank, flops_row, ref_logits, ref_rows)
        split["rows_with_identical_tokens"] = f"{int((x3_tokens == tokens).all(1).sum())} of {B} (last timed sample, same noise)"
        # for the record: the fp32 GEMMs with ONLY the attention kernel in split precision (HUDIFF_ATTN_X3=1)
        ao_tokens, ao = split_precision_leg(args, kind, cfg, sd, batch, T, Tmax, rank, local_rank, flops_row, ref_logits, ref_rows,
                                            env_name="HUDIFF_ATTN_X3")
        split["attention_kernel_only"] = {
            "value": ao["value"], "unit": "sequences/s", "steps": ao["steps"], "max_abs_dlogit_vs_f32_path": ao["max_abs_dlogit_vs_f32_path"],
            "rows_with_identical_tokens": f"{int((ao_tokens == tokens).all(1).sum())} of {B}",
            "note": "fp32 GEMMs + attn_x3_k (HUDIFF_ATTN_X3=1); not the metric"}

    # ---- the single collective of the job: gather the final tokens on rank 0 (RCCL over xGMI) ----------
    gathered = [tokens]
    if dist is not None:
        import torch
        t = torch.from_numpy(tokens).to(coll_dev)
        outs = [torch.empty_like(t) for _ in range(world)] if rank == 0 else None
        dist.gather(t, outs, dst=0)
        if rank == 0:
            gathered = [o.cpu().numpy() for o in outs]
        el = torch.tensor([elapsed, gpu_ms], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        elapsed, gpu_ms = float(el[0]), float(el[1])

    if rank == 0:
        all_tokens = np.concatenate(gathered)
        filled = bool(((all_tokens >= 0) & (all_tokens <= 22)).all())
        if args.max_t == 0:
            assert not (all_tokens == 22).any(), "masked slots left after a full sample"
        seqs = n_gpus * B * args.steps
        value = seqs / elapsed
        useful_flops = float(T.sum()) * flops_row * args.steps            # per GPU, algorithmic
        executed_flops = float(B * Tmax) * flops_row_exec * args.steps     # every row is computed every step
        achieved = useful_flops / (gpu_ms * 1e-3) / 1e12
        out = {
            "metric": "humanized sequences/sec (full T-step sample)" + (" on HuAb348" if (real and kind == "ab") else ""),
            "value": round(value, 4), "unit": "sequences/s", "n_gpus": n_gpus, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": (("HuAb348 mouse pairs" if kind == "ab" else "abnativ_select_vhh VHH") +
                     f" ({batch['n_sequences']} sequences of the reference's evaluation CSV, IMGT-slotted by hudiff_amd.numbering into "
                     "hudiff_amd/data/real_rows.npz, cycled with distinct replica noise); random-init weights of the production architecture")
            if real else "synthetic",
            "config": {"workload": (("HuDiff-Ab AntiTFNet (39.8M params, L=291) on HuAb348, " if real else
                                     "HuDiff-Ab AntiTFNet (39.8M params, L=291), HuAb348-shaped synthetic rows, ")
                                    if kind == "ab" else
                                    ("HuDiff-Nb NanoAntiTFNet (17.5M params, L=152) on abnativ_select_vhh, " if real else
                                     "HuDiff-Nb NanoAntiTFNet (17.5M params, L=152), VHH-shaped synthetic rows, "))
                       + f"{mode} mask, batch {B}/GPU, full T-step sample (T {int(batch['T'].min())}..{int(batch['T'].max())}, "
                         f"mean {float(batch['T'].mean()):.1f}), dropout {args.dropout}",
                       "rows_per_gpu": B, "global_rows": n_gpus * B, "denoiser_steps_per_sample": Tmax,
                       "parallelism": f"rows sharded x{n_gpus}, one RCCL gather of int32 tokens",
                       "process_group": formed},
            "roofline": {"bound": "mfma", "achieved": round(achieved, 3), "peak": PEAK_F32_MATRIX_TFLOPS,
                         "unit": "TFLOP/s", "frac": round(achieved / PEAK_F32_MATRIX_TFLOPS, 4), "traffic": None,
                         "launch": "one denoiser step = one replay of the captured hipGraph (all kernels of a forward "
                                   "+ sampling), HIP events on the library's stream",
                         "flops_per_launch": B * flops_row, "avg_launch_ms": round(gpu_ms / (args.steps * Tmax), 4),
                         "executed_tflops": round(executed_flops / (gpu_ms * 1e-3) / 1e12, 3),
                         "executed_frac": round(executed_flops / (gpu_ms * 1e-3) / 1e12 / PEAK_F32_MATRIX_TFLOPS, 4),
                         "executed_over_algorithmic": round(flops_row_exec / flops_row, 4)},
            "gpu_event_ms": round(gpu_ms, 2), "upload_ms": round(1e3 * upload_s, 2), "all_tokens_valid": filled,
        }
        if clock_power:
            # the 157.3 TFLOP/s peak is the 2.4 GHz figure; the package power limit decides the clock the kernels really ran at
            pk = PEAK_F32_MATRIX_TFLOPS * clock_power["sclk_mhz_median"] / 2400.0
            clock_power.update({"peak_at_sustained_clock": round(pk, 2), "frac_at_sustained_clock": round(achieved / pk, 4),
                                "source": "amdgpu hwmon freq1_input / power1_input of this GPU, 10 Hz over the timed region"})
            out["roofline"]["clock_power"] = clock_power
        # HBM-side traffic per launch: live PMC passes of this very command (subprocess, after the timed region), else the
        # committed passes of the round (profiles/r02/pmc_traffic.json, stamped with the commit they were taken at)
        if args.traffic in ("auto", "live") and n_gpus == 1 and args.max_t == 0:
            lt = live_traffic(args, kind, mode)
            if lt is not None:
                out["roofline"]["traffic"] = lt["traffic"]
                out["roofline"]["traffic_read_bytes"], out["roofline"]["traffic_write_bytes"] = lt["read_bytes"], lt["write_bytes"]
                out["roofline"]["traffic_note"] = lt["note"]
        if out["roofline"]["traffic"] is None and args.traffic in ("auto", "file"):
            try:
                tr = json.load(open(os.path.join(ROOT, "profiles", "r02", "pmc_traffic.json")))
                c = tr["config"]      # measured with one lane; the bytes do not depend on how the batch is split into lanes
                if (c["kind"], c["rows_per_gpu"], c["dropout"]) == (kind, B, args.dropout):
                    out["roofline"]["traffic"] = tr["traffic_bytes_per_launch"]
                    out["roofline"]["traffic_note"] = ("bytes per denoiser step, profiles/r02/pmc_traffic.json (FETCH_SIZE x2 + WRITE_SIZE), "
                                                       f"taken at commit {tr.get('git_head', '?')}")
            except Exception:
                pass
        if split is not None:
            out["split_precision"] = split
        if args.max_t > 0:
            out["truncated"] = f"--max-t {args.max_t}: NOT the metric (profiling run)"
        if not args.no_cpu_baseline and n_gpus == 1:
            out["cpu_baseline"] = cpu_baseline(kind, cfg, sd, mode, args.cpu_rows, args.cpu_steps, float(batch["T"].mean()),
                                               args.cpu_impl, args.data)
        print(json.dumps(out), flush=True)
    model.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
