#!/bin/bash
# per-kernel (by grid) table of a 6-step one-lane default-route run at a given batch: scripts/f32_bygrid_b.sh OUTNAME BATCH [ENV=VAL ...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$1; B=$2; shift 2
mkdir -p $OUT
CMD="python $R/bench.py --batch $B --steps 1 --warmup 0 --max-t 6 --no-cpu-baseline --lanes 1 --pmc off"
env HUDIFF_X3=0 "$@" timeout 400 rocprofv3 --kernel-trace -d $OUT/trace -o t -- $CMD > $OUT/trace.log 2>&1
python $R/scripts/rocpd_summary.py $(find $OUT/trace -name "*.db" | head -1) --by-grid > $OUT/by_grid.txt
rm -rf $OUT/trace
head -12 $OUT/by_grid.txt | cut -c1-120
