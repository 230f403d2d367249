#!/bin/bash
# A/B of a knob on the metric's default path: scripts/f32_ab.sh ENVVAR "v1 v2 ..." [repeats] [extra bench args]
cd $GRAFT_REPO_ROOT
var=$1; vals=$2; rep=${3:-2}; shift 3
for r in $(seq $rep); do for v in $vals; do
  out=$(env $var=$v python bench.py --steps 2 --warmup 1 --only-main --no-cpu-baseline --pmc off "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline']['avg_launch_ms'])")
  echo "$var=$v : $out"
done; done
