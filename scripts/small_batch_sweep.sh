#!/bin/bash
# sequences/s of full samples at small batches for HUDIFF_BIG_ROWS thresholds and precision routes:  scripts/small_batch_sweep.sh OUTNAME
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$1; mkdir -p $OUT
for route in ${ROUTES:-split f32_all}; do
 for thr in ${THRESHOLDS:-8192 4096 2048 1024 256}; do
  for B in ${SMALL_BATCHES:-1 4 8 16 24}; do
   v=$(HUDIFF_BIG_ROWS=$thr python $R/bench.py --batch $B --steps 2 --warmup 1 --no-cpu-baseline --pmc off --only-main --precision $route 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'])")
   echo "route $route big_rows $thr B $B : $v seq/s" | tee -a $OUT/sweep.txt
  done
 done
done
