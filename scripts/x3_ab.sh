#!/bin/bash
# A/B of a split-precision knob on the metric's workload: scripts/x3_ab.sh ENVVAR "v1 v2 ..." [repeats]
cd $GRAFT_REPO_ROOT
var=$1; vals=$2; rep=${3:-2}
for r in $(seq $rep); do for v in $vals; do
  out=$(env HUDIFF_X3=1 $var=$v python bench.py --steps 2 --warmup 1 --only-main --no-cpu-baseline --pmc off 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline']['avg_launch_ms'], d['precision_info'])")
  echo "$var=$v : $out"
done; done
