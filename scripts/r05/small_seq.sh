#!/bin/bash
# One denoiser step of a small batch, dispatch by dispatch (start, duration, grid, kernel):  scripts/r05/small_seq.sh OUT "1 8" [ab|nb]
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; KIND=${3:-ab}; mkdir -p $OUT
cd /tmp
for B in $2; do
  CMD="python $GRAFT_REPO_ROOT/bench.py --kind $KIND --batch $B --steps 1 --warmup 0 --max-t 6 --no-cpu-baseline --lanes 1 --pmc off --only-main"
  timeout 400 rocprofv3 --kernel-trace -d $OUT/trace$B -o t -- $CMD > $OUT/trace$B.log 2>&1
  python $GRAFT_REPO_ROOT/scripts/rocpd_summary.py $(find $OUT/trace$B -name "*.db" | head -1) --sequence > $OUT/${KIND}_B${B}_sequence.txt
  python $GRAFT_REPO_ROOT/scripts/rocpd_summary.py $(find $OUT/trace$B -name "*.db" | head -1) --by-grid > $OUT/${KIND}_B${B}_by_grid.txt
  rm -rf $OUT/trace$B
  head -1 $OUT/${KIND}_B${B}_sequence.txt
done
