#!/bin/bash
# Regenerates everything under profiles/r06 from ONE GPU box (run through gpurun; results land in gpurun_out/r06):
#   bash scripts/refresh_profiles_r06.sh <git-head>
# Precision routes and tuning options are chosen on the command line / through the interface; counter passes (inside bench.py) carry
# --kernel-trace only.
HEAD=${1:-unknown}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r06
rm -rf $OUT; mkdir -p $OUT
python $R/bench.py --steps 3 --warmup 1 2>$OUT/bench_ab.err | tail -1 > $OUT/bench_ab256.json
grep "^\[bench\]" $OUT/bench_ab.err > $OUT/bench_ab256_phases.txt
python $R/bench.py --kind nb --steps 3 --warmup 1 2>$OUT/bench_nb.err | tail -1 > $OUT/bench_nb256.json
prof() {   # name, kind, route, lanes, [ENV=VAL ...]
  name=$1; kind=$2; route=$3; lanes=$4; shift 4
  CMD="python $R/bench.py --kind $kind --steps 1 --warmup 0 --max-t 6 --no-cpu-baseline --lanes $lanes --pmc off --precision $route"
  env "$@" HUDIFF_QUIET=0 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/st_$name -o t -- $CMD > $OUT/st_$name.log 2>&1
  cp $(find $OUT/st_$name -name "*kernel_stats.csv" | head -1) $OUT/${name}_kernel_stats.csv
  env "$@" HUDIFF_QUIET=0 timeout 400 rocprofv3 --kernel-trace -d $OUT/tr_$name -o t -- $CMD > $OUT/tr_$name.log 2>&1
  python $R/scripts/rocpd_summary.py $(find $OUT/tr_$name -name "*.db" | head -1) --by-grid > $OUT/${name}_by_grid.txt
  rm -rf $OUT/st_$name $OUT/tr_$name $OUT/st_$name.log $OUT/tr_$name.log
}
prof ab256_split_maxt6_lanes1 ab split 1
prof ab256_split_maxt6_lanes2 ab split 2
prof ab256_split_chain_maxt6_lanes1 ab split 1 HUDIFF_BN_CHAIN=3
prof ab256_f32all_maxt6_lanes1 ab f32_all 1
prof nb256_split_maxt6_lanes1 nb split 1
prof nb256_split_chain_maxt6_lanes1 nb split 1 HUDIFF_BN_CHAIN=3
seq() {   # name, budget route, precision route, [ENV=VAL ...]
  name=$1; broute=$2; route=$3; shift 3
  CMD="python $R/bench.py --steps 1 --warmup 0 --max-t 4 --no-cpu-baseline --lanes 1 --pmc off --only-main --precision $route"
  env "$@" HUDIFF_QUIET=0 timeout 400 rocprofv3 --kernel-trace -d $OUT/sq_$name -o t -- $CMD > $OUT/sq_$name.log 2>&1
  python $R/scripts/rocpd_summary.py $(find $OUT/sq_$name -name "*.db" | head -1) --sequence > $OUT/${name}_step_sequence.txt
  rm -rf $OUT/sq_$name $OUT/sq_$name.log
}
seq ab256_split x3 split
python $R/scripts/launch_budget.py $OUT/ab256_split_step_sequence.txt x3 > $OUT/ab256_split_launch_budget.txt
seq ab256_split_chain x3 split HUDIFF_BN_CHAIN=3
# chain kernel on / off on the metric itself (two lanes), interleaved
for rep in 1 2; do for c in 0 3; do
  v=$(HUDIFF_BN_CHAIN=$c python $R/bench.py --steps 3 --warmup 1 --only-main 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'])")
  echo "ab 256 rows, option bn_chain=$c : $v sequences/s" >> $OUT/chain_on_off.txt
  v=$(HUDIFF_BN_CHAIN=$c python $R/bench.py --kind nb --steps 3 --warmup 1 --only-main 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'])")
  echo "nb 256 rows, option bn_chain=$c : $v sequences/s" >> $OUT/chain_on_off.txt
done; done
# parity evidence that prints: flip-rate counts per route, per-stage errors at production width, chain kernel
(cd $R && timeout 900 python -m pytest tests/test_fliprate.py tests/test_gpu_components.py tests/test_gpu_chain.py tests/test_gpu_cli_golden.py -m gpu -q -s 2>&1 | grep "fliprate\|worst relative\|passed\|failed" > $OUT/parity_prints.txt)
python $R/scripts/adv_report.py $OUT/adversarial_errors.json > $OUT/adv.log 2>&1
(python $R/scripts/cli_e2e.py 1) 2>/dev/null | grep "end to end" > $OUT/cli_e2e.txt
for kind in ab nb; do
  for v in "HUDIFF_QUIET=0" "HUDIFF_FUSED_ATTN=0" "HUDIFF_LANES=1" "HUDIFF_X3_LNSYNC=0" "HUDIFF_BN_CHAIN=3" "HUDIFF_PRECISION=f32_all"; do
    echo "$v : $(env $v python $R/scripts/lnsync_soak.py $kind 2 2>&1 | tail -1)" >> $OUT/option_soak.txt
  done
done
# (kernel_resources.txt: python scripts/spill_report.py --all, run in the build container -- it only compiles)
echo "$HEAD" > $OUT/GIT_HEAD
ls -la $OUT
cut -c1-300 $OUT/bench_ab256.json
