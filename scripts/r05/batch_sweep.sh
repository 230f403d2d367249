#!/bin/bash
# sequences/s over batch sizes for environment variants: scripts/r05/batch_sweep.sh OUT "tagA:VAR=1 tagB:VAR=0" "1 8 32" "ab nb" [extra bench args]
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=gpurun_out/$1; mkdir -p $OUT; VARS=$2; BS=$3; KINDS=${4:-ab}; EXTRA=${5:-}
for kind in $KINDS; do for B in $BS; do for v in $VARS; do
  tag=${v%%:*}; e=$(echo "$v" | cut -d: -f2- | tr ',' ' ')
  val=$(env $e python bench.py --kind $kind --batch $B --steps 2 --warmup 1 --only-main --no-cpu-baseline --pmc off $EXTRA 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'])")
  echo "$kind B $B $tag : $val" | tee -a $OUT/sweep.txt
done; done; done
