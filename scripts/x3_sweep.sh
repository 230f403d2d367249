#!/bin/bash
# split-precision path: lanes x tile-shape sweep of the metric's workload (HUDIFF_X3=1 exported: the main leg runs the split kernels)
cd $GRAFT_REPO_ROOT
for lanes in 2 3 4; do for tile in 0 128 512; do
  v=$(HUDIFF_X3=1 HUDIFF_LANES=$lanes HUDIFF_X3_TILE=$tile python bench.py --steps 2 --warmup 1 --only-main --no-cpu-baseline --pmc off 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline']['avg_launch_ms'])")
  echo "lanes $lanes tile $tile : $v"
done; done
