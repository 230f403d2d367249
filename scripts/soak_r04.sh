#!/bin/bash
# Token digests of complete 256-row samples under different lane counts (lanes of 256 ... 64 sequences: all tile shapes, pipeline depths and
# forms of the pruned tail), ln_sync on / off, the three precision routes and the tail forms: every line of a model must print the same digest.
#   gpurun -- bash scripts/soak_r04.sh      (result: gpurun_out/soak/lnsync_soak.txt)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/soak; mkdir -p $OUT; rm -f $OUT/lnsync_soak.txt
for kind in ab nb; do
  for lanes in 2 1 3 4; do echo "route split tail default $(HUDIFF_PRECISION=split HUDIFF_LANES=$lanes python $R/scripts/lnsync_soak.py $kind 2 2>&1 | tail -1)" >> $OUT/lnsync_soak.txt; done
  echo "route split tail 0 (separate launches) $(HUDIFF_PRECISION=split HUDIFF_LANES=4 HUDIFF_TAIL=0 python $R/scripts/lnsync_soak.py $kind 2 2>&1 | tail -1)" >> $OUT/lnsync_soak.txt
  echo "route split tail 1 (one kernel) $(HUDIFF_PRECISION=split HUDIFF_LANES=4 HUDIFF_TAIL=1 python $R/scripts/lnsync_soak.py $kind 2 2>&1 | tail -1)" >> $OUT/lnsync_soak.txt
  echo "route split lnsync off $(HUDIFF_PRECISION=split HUDIFF_X3_LNSYNC=0 python $R/scripts/lnsync_soak.py $kind 2 2>&1 | tail -1)" >> $OUT/lnsync_soak.txt
  echo "route f32_gemm $(HUDIFF_PRECISION=f32_gemm python $R/scripts/lnsync_soak.py $kind 2 2>&1 | tail -1)" >> $OUT/lnsync_soak.txt
  echo "route f32_all $(HUDIFF_PRECISION=f32_all HUDIFF_LANES=4 python $R/scripts/lnsync_soak.py $kind 2 2>&1 | tail -1)" >> $OUT/lnsync_soak.txt
done
cut -c1-140 $OUT/lnsync_soak.txt
