#!/usr/bin/env python
"""Where a small-batch step's time is: sequences/s, WALL milliseconds per denoiser step and HIP-EVENT milliseconds per denoiser step
(device time between the first and the last replay of a sample), for the step graph (T replays of one captured step) and for the loop
graph (the whole T-step loop as one hipGraph).  wall - event = host-side launch time that the device waits for.
    python scripts/small_batch_times.py [ab|nb] "1 8 16" [out.txt]          (VERDICT r4 "Next" #3 i)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
kind = sys.argv[1] if len(sys.argv) > 1 else "ab"
batches = [int(b) for b in (sys.argv[2] if len(sys.argv) > 2 else "1 8").split()]
out = open(sys.argv[3], "a") if len(sys.argv) > 3 else None
for B in batches:
    for tag, extra in (("step graph", []), ("loop graph", ["--loop-graph"])):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--kind", kind, "--batch", str(B), "--steps", "3", "--warmup", "1",
                            "--only-main", "--no-cpu-baseline", "--pmc", "off", *extra], capture_output=True, text=True, cwd=ROOT)
        d = json.loads(r.stdout.strip().splitlines()[-1])
        tmax = d["config"]["denoiser_steps_per_sample"]
        wall = d["ms_per_step"] / tmax if tmax else float("nan")
        line = (f"{kind} B {B:3d} {tag}: {d['value']:8.3f} sequences/s | wall {wall:.4f} ms per denoiser step | HIP events "
                f"{d['roofline']['avg_launch_ms']:.4f} ms per denoiser step | host-bound share {max(0.0, 1 - d['roofline']['avg_launch_ms'] / wall):.2f}")
        print(line, flush=True)
        if out:
            out.write(line + "\n"); out.flush()
