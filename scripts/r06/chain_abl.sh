#!/bin/bash
# per-launch times of the chain kernels in the probe libraries (scripts/r06/build_abl.sh):  scripts/r06/chain_abl.sh OUT "0 1 10 4 112 126"     (through gpurun)
OUT=$1
for a in $2; do
  if [ "$a" = "0" ]; then L=""; else L="HUDIFF_LIB=$GRAFT_REPO_ROOT/hudiff_amd/libhudiff_abl$a.so"; fi
  bash $GRAFT_REPO_ROOT/scripts/r06/prof_bygrid.sh $OUT abl$a HUDIFF_QUIET=1 $L
  echo "== HD_CHAIN_ABL=$a" >> $GRAFT_REPO_ROOT/$OUT/abl_summary.txt
  grep "bn_chain\|total kernel" $GRAFT_REPO_ROOT/$OUT/abl${a}_by_grid.txt >> $GRAFT_REPO_ROOT/$OUT/abl_summary.txt
done
cat $GRAFT_REPO_ROOT/$OUT/abl_summary.txt
