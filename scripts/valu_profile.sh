#!/bin/bash
# Per-kernel vector-ALU vs MFMA instruction counts for a few denoiser steps (one lane, graph off): on gfx950 the two
# time-slice one issue port, so (non-MFMA VALU instructions x 4 cycles) is matrix time lost.  Counters only.
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/valu
mkdir -p $OUT
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace -d $OUT/p -o p --output-format csv -- \
    python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --max-t 3 --no-cpu-baseline --traffic off --lanes 1 --no-graph "$@" > $OUT/run.log 2>&1
tail -1 $OUT/run.log | cut -c1-200
python3 - <<PY
import csv, glob, collections, re
f = sorted(glob.glob("$OUT/p/**/*counter_collection.csv", recursive=True))[-1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0] + " g=" + r["Grid_Size"]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_INSTS_VALU": n[k] += 1
rows = []
for k, c in acc.items():
    mf = c["SQ_INSTS_MFMA"]; va = c["SQ_INSTS_VALU"] - mf
    cyc_m = mf * (32 if "attn_k" in k else 64)
    rows.append((va * 4 + cyc_m, k, n[k], va, mf, cyc_m))
tot = sum(r[0] for r in rows)
print(f"{'kernel':86s} {'calls':>5s} {'VALU(M)':>9s} {'MFMA(M)':>9s} {'valu_cyc/mfma_cyc':>18s} {'share':>6s}")
for t, k, nn, va, mf, cm in sorted(rows, reverse=True)[:24]:
    print(f"{k[:86]:86s} {nn:5d} {va/1e6:9.1f} {mf/1e6:9.1f} {(va*4/cm if cm else float('inf')):18.3f} {100*t/tot:6.2f}")
print("total VALU cycles / MFMA cycles:", sum(r[3] for r in rows) * 4 / max(sum(r[5] for r in rows), 1))
PY
