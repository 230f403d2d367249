cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for r in 1 2; do for k in ab nb; do echo "R=$r $k X3=1 $(HUDIFF_X3=1 timeout 200 python $R/bench.py --kind $k --steps 3 --warmup 1 --no-cpu-baseline --traffic off --no-split-line 2>&1 | tail -1 | cut -c40-100)"; done; done
CMD="python $R/bench.py --steps 1 --warmup 0 --max-t 6 --no-cpu-baseline --lanes 1 --traffic off"
rm -rf /tmp/st; HUDIFF_X3=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st -o t -- $CMD > /tmp/st.log 2>&1
grep attn_x3 $(find /tmp/st -name "*kernel_stats.csv" | head -1) | cut -c1-160
cd $R; timeout 900 python -m pytest tests/test_gpu_x3.py tests/test_prod_trace.py tests/test_gpu_evalsets.py -x -q 2>&1 | tail -2
