#!/bin/bash
# per-kernel PMC sums of a short one-lane run: scripts/r05/pmc_kernel.sh OUT TAG "COUNTER1 COUNTER2 ..." KERNEL_SUBSTRING [ENV=VAL ...]
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$1; TAG=$2; CNT=$3; KSUB=$4; shift 4
mkdir -p $OUT
CMD="python $R/bench.py --steps 1 --warmup 0 --max-t 3 --no-cpu-baseline --lanes 1 --pmc off --only-main"
env "$@" timeout 300 rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d $OUT/pmc_$TAG -o p -- $CMD > $OUT/pmc_$TAG.log 2>&1
python - "$OUT/pmc_$TAG" "$KSUB" "$TAG" <<'PY'
import csv, glob, os, sys
d, ksub, tag = sys.argv[1:4]
tot, n = {}, 0
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if ksub in r["Kernel_Name"]:
            tot[r["Counter_Name"]] = tot.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
print(tag, {k: f"{v:.4g}" for k, v in tot.items()})
PY
rm -rf $OUT/pmc_$TAG
