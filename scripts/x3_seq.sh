#!/bin/bash
# dispatch sequence (launch order, duration each) of one denoiser step of a one-lane run: scripts/x3_seq.sh OUTNAME [ENV=VAL ...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$1; shift
mkdir -p $OUT
CMD="python $R/bench.py --steps 1 --warmup 0 --max-t 4 --no-cpu-baseline --lanes 1 --pmc off --only-main"
env "$@" timeout 400 rocprofv3 --kernel-trace -d $OUT/trace -o t -- $CMD > $OUT/trace.log 2>&1
python $R/scripts/rocpd_summary.py $(find $OUT/trace -name "*.db" | head -1) --sequence > $OUT/sequence.txt
rm -rf $OUT/trace
wc -l $OUT/sequence.txt
