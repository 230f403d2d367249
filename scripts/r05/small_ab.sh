#!/bin/bash
# Small-batch A/B over library builds (one gpurun call): scripts/r05/small_ab.sh OUT "libA.so libB.so" "1 2 4 8" [kinds]
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=gpurun_out/$1; LIBS=$2; BS=$3; KINDS=${4:-"ab nb"}
mkdir -p $OUT
for kind in $KINDS; do for B in $BS; do for l in $LIBS; do
  v=$(env HUDIFF_LIB=$GRAFT_REPO_ROOT/$l python bench.py --kind $kind --batch $B --steps 3 --warmup 1 --no-cpu-baseline --pmc off --only-main 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'])")
  echo "$kind B $B $(basename $l .so) : $v" | tee -a $OUT/small_ab.txt
done; done; done
