#!/bin/bash
# Round-5 A/B driver (one gpurun call): scripts/r05/ab.sh OUT "libA.so libB.so ..." [repeats] [pytest -k expression | none] [profile-lib | none]
#   1. a subset of the -m gpu suite on the in-tree library (the candidate), 2. interleaved bench.py samples of every library
#   (two-lane metric workload, --only-main), 3. one-lane by-grid kernel table of `profile-lib`.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=gpurun_out/$1; LIBS=$2; REP=${3:-3}; KEXPR=${4:-none}; PROF=${5:-none}
mkdir -p $OUT
if [ "$KEXPR" != "none" ]; then
  timeout 2400 python -m pytest tests/test_gpu_x3.py tests/test_prod_trace.py -x -q -m gpu -k "$KEXPR" -p no:cacheprovider > $OUT/tests.log 2>&1
  echo "tests rc $? : $(tail -1 $OUT/tests.log)"
fi
for r in $(seq $REP); do for l in $LIBS; do
  v=$(env HUDIFF_LIB=$GRAFT_REPO_ROOT/$l python bench.py --steps 2 --warmup 1 --only-main --no-cpu-baseline --pmc off 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline']['avg_launch_ms'], d['roofline'].get('clock_power',{}).get('sclk_mhz_median'))")
  echo "$l : $v" | tee -a $OUT/ab.txt
done; done
if [ "$PROF" != "none" ]; then
  cd /tmp
  CMD="python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --max-t 6 --no-cpu-baseline --lanes 1 --pmc off --only-main"
  env HUDIFF_LIB=$GRAFT_REPO_ROOT/$PROF timeout 400 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$OUT/trace -o t -- $CMD > $GRAFT_REPO_ROOT/$OUT/trace.log 2>&1
  python $GRAFT_REPO_ROOT/scripts/rocpd_summary.py $(find $GRAFT_REPO_ROOT/$OUT/trace -name "*.db" | head -1) --by-grid > $GRAFT_REPO_ROOT/$OUT/by_grid.txt
  rm -rf $GRAFT_REPO_ROOT/$OUT/trace
  head -14 $GRAFT_REPO_ROOT/$OUT/by_grid.txt
fi
