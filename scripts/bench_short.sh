#!/bin/bash
# usage: scripts/bench_short.sh [bench.py args...]  -> value, TF/s, ms/step
python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline "$@" 2>&1 | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$*', '->', d['value'], 'seq/s', d['roofline']['achieved'], 'TF/s', d['roofline']['avg_launch_ms'], 'ms/step')"
