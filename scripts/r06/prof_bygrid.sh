#!/bin/bash
# Per-kernel times of a 6-step sample (one lane): scripts/r06/prof_bygrid.sh OUTDIR NAME [ENV=VAL ...]      (through gpurun)
OUT=$1; NAME=$2; shift 2
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/$OUT
KIND=${KIND:-ab}; LANES=${LANES:-1}
CMD="python $R/bench.py --kind $KIND --steps 1 --warmup 0 --max-t 6 --no-cpu-baseline --lanes $LANES --pmc off --only-main --precision split"
env "$@" timeout 400 rocprofv3 --kernel-trace -d $R/$OUT/tr_$NAME -o t -- $CMD > $R/$OUT/tr_$NAME.log 2>&1
python $R/scripts/rocpd_summary.py $(find $R/$OUT/tr_$NAME -name "*.db" | head -1) --by-grid > $R/$OUT/${NAME}_by_grid.txt
python $R/scripts/rocpd_summary.py $(find $R/$OUT/tr_$NAME -name "*.db" | head -1) --sequence > $R/$OUT/${NAME}_step_sequence.txt
rm -rf $R/$OUT/tr_$NAME
