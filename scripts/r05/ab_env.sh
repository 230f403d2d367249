#!/bin/bash
# Round-5 A/B driver over ENVIRONMENT variants of the in-tree library (one gpurun call):
#   scripts/r05/ab_env.sh OUT "tagA:VAR=1,VAR2=x tagB:VAR=0" [repeats] [pytest -k expression | none] [profile-tag | none] [extra bench args]
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=gpurun_out/$1; VARS=$2; REP=${3:-3}; KEXPR=${4:-none}; PROF=${5:-none}; EXTRA=${6:-}
mkdir -p $OUT
envof() { echo "$1" | cut -d: -f2- | tr ',' ' '; }
if [ "$KEXPR" != "none" ]; then
  timeout 2400 python -m pytest tests/test_gpu_x3.py tests/test_prod_trace.py -x -q -m gpu -k "$KEXPR" -p no:cacheprovider > $OUT/tests.log 2>&1
  echo "tests rc $? : $(tail -1 $OUT/tests.log)"
fi
for r in $(seq $REP); do for v in $VARS; do
  tag=${v%%:*}
  val=$(env $(envof $v) python bench.py --steps 2 --warmup 1 --only-main --no-cpu-baseline --pmc off $EXTRA 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline']['avg_launch_ms'], d['roofline'].get('clock_power',{}).get('sclk_mhz_median'))")
  echo "$tag : $val" | tee -a $OUT/ab.txt
done; done
for v in $VARS; do
  tag=${v%%:*}
  if [ "$PROF" == "$tag" ] || [ "$PROF" == "all" ]; then
    cd /tmp
    CMD="python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --max-t 6 --no-cpu-baseline --lanes 1 --pmc off --only-main $EXTRA"
    env $(envof $v) timeout 400 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$OUT/trace_$tag -o t -- $CMD > $GRAFT_REPO_ROOT/$OUT/trace_$tag.log 2>&1
    python $GRAFT_REPO_ROOT/scripts/rocpd_summary.py $(find $GRAFT_REPO_ROOT/$OUT/trace_$tag -name "*.db" | head -1) --by-grid > $GRAFT_REPO_ROOT/$OUT/by_grid_$tag.txt
    rm -rf $GRAFT_REPO_ROOT/$OUT/trace_$tag
    echo "== $tag"; head -12 $GRAFT_REPO_ROOT/$OUT/by_grid_$tag.txt
    cd $GRAFT_REPO_ROOT
  fi
done
