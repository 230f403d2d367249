#!/bin/bash
# Round-5 A/B driver over several builds of the library (one gpurun call):
#   scripts/r05/ab_libs.sh OUT "libA.so libB.so ..." [repeats] [pytest -k expression | none] [extra bench args]
#   1. the -k subset of the split-route GPU tests on EVERY library (HUDIFF_LIB), 2. interleaved bench.py samples (two-lane metric
#   workload, --only-main), 3. the one-lane by-grid kernel table of every library (top rows).
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=gpurun_out/$1; LIBS=$2; REP=${3:-3}; KEXPR=${4:-none}; EXTRA=${5:-}
mkdir -p $OUT
if [ "$KEXPR" != "none" ]; then
  for l in $LIBS; do
    env HUDIFF_LIB=$GRAFT_REPO_ROOT/$l timeout 1500 python -m pytest tests/test_gpu_x3.py tests/test_prod_trace.py -x -q -m gpu -k "$KEXPR" -p no:cacheprovider > $OUT/tests_$(basename $l .so).log 2>&1
    echo "tests $l rc $? : $(tail -1 $OUT/tests_$(basename $l .so).log)"
  done
fi
for r in $(seq $REP); do for l in $LIBS; do
  v=$(env HUDIFF_LIB=$GRAFT_REPO_ROOT/$l python bench.py --steps 2 --warmup 1 --only-main --no-cpu-baseline --pmc off $EXTRA 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline']['avg_launch_ms'], d['roofline'].get('clock_power',{}).get('sclk_mhz_median'))")
  echo "$l : $v" | tee -a $OUT/ab.txt
done; done
for l in $LIBS; do
  tag=$(basename $l .so)
  cd /tmp
  CMD="python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --max-t 6 --no-cpu-baseline --lanes 1 --pmc off --only-main $EXTRA"
  env HUDIFF_LIB=$GRAFT_REPO_ROOT/$l timeout 400 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$OUT/trace_$tag -o t -- $CMD > $GRAFT_REPO_ROOT/$OUT/trace_$tag.log 2>&1
  python $GRAFT_REPO_ROOT/scripts/rocpd_summary.py $(find $GRAFT_REPO_ROOT/$OUT/trace_$tag -name "*.db" | head -1) --by-grid > $GRAFT_REPO_ROOT/$OUT/by_grid_$tag.txt
  python $GRAFT_REPO_ROOT/scripts/rocpd_summary.py $(find $GRAFT_REPO_ROOT/$OUT/trace_$tag -name "*.db" | head -1) --sequence > $GRAFT_REPO_ROOT/$OUT/sequence_$tag.txt
  rm -rf $GRAFT_REPO_ROOT/$OUT/trace_$tag
  echo "== $tag"; head -8 $GRAFT_REPO_ROOT/$OUT/by_grid_$tag.txt
  cd $GRAFT_REPO_ROOT
done
