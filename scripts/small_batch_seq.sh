#!/bin/bash
# dispatch sequence of one denoiser step at small batches (where does a B = 1 / 8 / 32 step spend its time): scripts/small_batch_seq.sh OUTNAME [ENV=VAL ...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$1; shift
mkdir -p $OUT
for B in ${SMALL_BATCHES:-1 8 32}; do
  CMD="python $R/bench.py --batch $B --steps 1 --warmup 0 --max-t 4 --no-cpu-baseline --lanes 1 --pmc off --only-main"
  env "$@" timeout 300 rocprofv3 --kernel-trace -d $OUT/sq_$B -o t -- $CMD > $OUT/sq_$B.log 2>&1
  python $R/scripts/rocpd_summary.py $(find $OUT/sq_$B -name "*.db" | head -1) --sequence > $OUT/b${B}_step_sequence.txt
  rm -rf $OUT/sq_$B
  head -1 $OUT/b${B}_step_sequence.txt
  # un-profiled rate of a full sample (graph replay)
  env "$@" python $R/bench.py --batch $B --steps 2 --warmup 1 --no-cpu-baseline --pmc off --only-main 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('B=$B', d['value'], 'seq/s', d['ms_per_step'], 'ms/sample')"
done
