#!/bin/bash
# per-kernel times of a 6-step one-lane HuDiff-Nb run (fp32 path; X3=1 in the environment for the split-precision kernels)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/nb; mkdir -p $OUT
CMD="python $R/bench.py --kind nb --steps 1 --warmup 0 --max-t 6 --no-cpu-baseline --lanes 1 --traffic off --no-split-line"
rm -rf /tmp/nbt; timeout 400 rocprofv3 --kernel-trace -d /tmp/nbt -o t -- $CMD > $OUT/trace.log 2>&1
python $R/scripts/rocpd_summary.py $(find /tmp/nbt -name "*.db" | head -1) --by-grid > $OUT/nb256_maxt6_lanes1_by_grid${HUDIFF_X3:+_x3}.txt
head -30 $OUT/nb256_maxt6_lanes1_by_grid${HUDIFF_X3:+_x3}.txt | cut -c1-150
