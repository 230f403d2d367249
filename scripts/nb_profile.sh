cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/nbp -o t -- python $GRAFT_REPO_ROOT/bench.py --kind nb --steps 1 --warmup 0 --max-t 6 --no-cpu-baseline --lanes 1 > $GRAFT_REPO_ROOT/gpurun_out/nbp.log 2>&1
python $GRAFT_REPO_ROOT/scripts/rocpd_summary.py $GRAFT_REPO_ROOT/gpurun_out/nbp/t_results.db --by-grid | head -24
