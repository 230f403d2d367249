#!/bin/bash
# usage: scripts/ab_profile.sh NAME [ENV=VAL ...]   -- rocprofv3 kernel trace of 6 denoiser steps (one lane), summary by grid
name=$1; shift
cd /tmp && export TMPDIR=/tmp
env "$@" rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/ab_$name -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --max-t 6 --no-cpu-baseline --lanes 1 > $GRAFT_REPO_ROOT/gpurun_out/ab_$name.log 2>&1
python $GRAFT_REPO_ROOT/scripts/rocpd_summary.py $GRAFT_REPO_ROOT/gpurun_out/ab_$name/t_results.db --by-grid | head -16
