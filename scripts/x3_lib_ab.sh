#!/bin/bash
# A/B of alternative builds of the library on the split-precision sample: scripts/x3_lib_ab.sh "libA.so libB.so ..." [repeats]   (paths relative to the repo)
cd $GRAFT_REPO_ROOT
libs=$1; rep=${2:-2}
for r in $(seq $rep); do for l in $libs; do
  out=$(env HUDIFF_X3=1 HUDIFF_LIB=$GRAFT_REPO_ROOT/$l python bench.py --steps 2 --warmup 1 --only-main --no-cpu-baseline --pmc off 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline']['avg_launch_ms'])")
  echo "$l : $out"
done; done
