#!/bin/bash
# fused-attention ablations (HUDIFF_QA_ABL bits) under rocprofv3: µs per launch of qkv_attn_x3_k.   scripts/r05/qa_abl.sh OUT "0 1 2 3 4 7 12" [kind]
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$1; mkdir -p $OUT; KIND=${3:-ab}
for a in $2; do
  CMD="python $R/bench.py --kind $KIND --steps 1 --warmup 0 --max-t 4 --no-cpu-baseline --lanes 1 --pmc off --only-main"
  env HUDIFF_QA_ABL=$a timeout 300 rocprofv3 --kernel-trace -d $OUT/tr_$a -o t -- $CMD > $OUT/tr_$a.log 2>&1
  python $R/scripts/rocpd_summary.py $(find $OUT/tr_$a -name "*.db" | head -1) --by-grid > $OUT/abl_$a.txt
  rm -rf $OUT/tr_$a
  echo "abl $a: $(grep qkv_attn $OUT/abl_$a.txt | head -1)"
done
