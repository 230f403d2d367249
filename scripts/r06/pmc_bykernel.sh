#!/bin/bash
# per (kernel, grid) PMC averages per launch of a short one-lane run, one rocprofv3 pass per counter group (--kernel-trace only beside --pmc):
#   scripts/r06/pmc_bykernel.sh OUT NAME [ENV=VAL ...]        FETCH_SIZE is doubled (gfx950 correction, MI355X_MICROARCH.md); sizes in KB
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/$1; NAME=$2; shift 2
mkdir -p $OUT
CMD="python $R/bench.py --steps 1 --warmup 0 --max-t 3 --no-cpu-baseline --lanes 1 --pmc off --only-main"
i=0
for CNT in "FETCH_SIZE" "WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  env "$@" HUDIFF_QUIET=1 timeout 300 rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d $OUT/pmc_${NAME}_$i -o p -- $CMD > $OUT/pmc_${NAME}_$i.log 2>&1
done
python - "$OUT" "$NAME" > $OUT/${NAME}_pmc_by_kernel.txt <<'PY'
import csv, glob, os, sys
d, name = sys.argv[1], sys.argv[2]
agg, calls = {}, {}
for f in glob.glob(os.path.join(d, f"pmc_{name}_*", "**", "*counter_collection.csv"), recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        kn = r["Kernel_Name"].split("(")[0].replace("void ", "")
        key = (kn, r.get("Grid_Size", "?"))
        c = r["Counter_Name"]
        agg.setdefault(key, {}).setdefault(c, 0.0)
        agg[key][c] += float(r["Counter_Value"])
        k2 = (key, c, r.get("Dispatch_Id"))
        if k2 not in seen:
            seen.add(k2)
            calls.setdefault(key, {}).setdefault(c, 0)
            calls[key][c] += 1
rows, tot_f, tot_w = [], 0.0, 0.0
for key, cs in agg.items():
    n = max(calls[key].values())
    g = {c: v / calls[key][c] for c, v in cs.items()}
    fetch = 2 * g.get("FETCH_SIZE", 0) * 1024 / 1e6
    write = g.get("WRITE_SIZE", 0) * 1024 / 1e6
    tot_f += fetch * n; tot_w += write * n
    gui = g.get("GRBM_GUI_ACTIVE", 0) / 8.0
    busy = g.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(gui * 1024, 1)
    wc = max(g.get("SQ_WAVE_CYCLES", 0), 1)
    rows.append((fetch + write, key, n, fetch, write, busy, g.get("SQ_WAIT_INST_ANY", 0) / wc, g.get("SQ_WAIT_ANY", 0) / wc, g.get("SQ_ACTIVE_INST_ANY", 0) / wc,
                 g.get("SQ_INSTS_VALU", 0), g.get("SQ_INSTS_MFMA", 0), g.get("SQ_INSTS_LDS", 0), g.get("SQ_INSTS_VMEM_RD", 0)))
print(f"# 3 denoiser steps, one lane; HBM-side bytes of ALL launches: fetch {tot_f / 3e3:.1f} GB + write {tot_w / 3e3:.1f} GB = {(tot_f + tot_w) / 3e3:.1f} GB per step")
print(f"{'kernel grid':80s} {'calls':>5s} {'fetch MB':>9s} {'write MB':>9s} {'mfma busy':>9s} {'wait_inst':>9s} {'wait_any':>8s} {'active':>7s} {'VALU':>9s} {'MFMA':>9s} {'LDS':>9s} {'VMEM_RD':>9s}   (per launch)")
for tot, key, n, fetch, write, busy, wi, wa, ac, iv, im, il, ivm in sorted(rows, reverse=True)[:22]:
    print(f"{(key[0][:62] + ' ' + key[1]):80s} {n:5d} {fetch:9.1f} {write:9.1f} {busy:9.3f} {wi:9.3f} {wa:8.3f} {ac:7.3f} {iv:9.3g} {im:9.3g} {il:9.3g} {ivm:9.3g}")
PY
rm -rf $OUT/pmc_${NAME}_[0-9] $OUT/pmc_${NAME}_[0-9].log
cat $OUT/${NAME}_pmc_by_kernel.txt
