#!/bin/bash
# Regenerates everything under profiles/r03 from ONE GPU box (run through gpurun; results land in gpurun_out/r03):
#   bash scripts/refresh_profiles_r03.sh <git-head>
# bench lines (Ab on HuAb348 with its all-fp32 / split-precision / HuDiff-Nb secondary objects and live PMC passes; Nb plain and
# inpaint), rocprofv3 --kernel-trace --stats of a 6-step one-lane run for the three routes (default, all fp32, HUDIFF_X3=1), the
# adversarial-statistics error table.  Counter passes (inside bench.py) carry --kernel-trace only.
HEAD=${1:-unknown}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03
rm -rf $OUT; mkdir -p $OUT
python $R/bench.py --steps 2 --warmup 1 2>$OUT/bench_ab.err | tail -1 > $OUT/bench_ab256.json
python $R/bench.py --kind nb --steps 2 --warmup 1 2>$OUT/bench_nb.err | tail -1 > $OUT/bench_nb256.json
python $R/bench.py --kind nb --mode inpaint --steps 2 --warmup 1 --no-cpu-baseline 2>>$OUT/bench_nb.err | tail -1 > $OUT/bench_nb256_inpaint.json
prof() {   # name, kind, env...
  name=$1; kind=$2; shift 2
  CMD="python $R/bench.py --kind $kind --steps 1 --warmup 0 --max-t 6 --no-cpu-baseline --lanes 1 --pmc off"
  env "$@" timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/st_$name -o t -- $CMD > $OUT/st_$name.log 2>&1
  cp $(find $OUT/st_$name -name "*kernel_stats.csv" | head -1) $OUT/${name}_maxt6_lanes1_kernel_stats.csv
  env "$@" timeout 400 rocprofv3 --kernel-trace -d $OUT/tr_$name -o t -- $CMD > $OUT/tr_$name.log 2>&1
  python $R/scripts/rocpd_summary.py $(find $OUT/tr_$name -name "*.db" | head -1) --by-grid > $OUT/${name}_maxt6_lanes1_by_grid.txt
  rm -rf $OUT/st_$name $OUT/tr_$name $OUT/st_$name.log $OUT/tr_$name.log
}
prof ab256 ab HUDIFF_X3=0
prof ab256_allfp32 ab HUDIFF_X3=0 HUDIFF_ATTN_X3=0
prof ab256_x3 ab HUDIFF_X3=1
prof nb256 nb HUDIFF_X3=0
prof nb256_x3 nb HUDIFF_X3=1
# dispatch sequence of one denoiser step + per-launch roofline budget (scripts/launch_budget.py) on the three routes
seq() {   # name, route, env...
  name=$1; route=$2; shift 2
  CMD="python $R/bench.py --steps 1 --warmup 0 --max-t 4 --no-cpu-baseline --lanes 1 --pmc off --only-main"
  env "$@" timeout 400 rocprofv3 --kernel-trace -d $OUT/sq_$name -o t -- $CMD > $OUT/sq_$name.log 2>&1
  python $R/scripts/rocpd_summary.py $(find $OUT/sq_$name -name "*.db" | head -1) --sequence > $OUT/${name}_step_sequence.txt
  python $R/scripts/launch_budget.py $OUT/${name}_step_sequence.txt $route > $OUT/${name}_launch_budget.txt
  rm -rf $OUT/sq_$name $OUT/sq_$name.log
}
seq ab256 default HUDIFF_X3=0
seq ab256_allfp32 allfp32 HUDIFF_X3=0 HUDIFF_ATTN_X3=0
seq ab256_x3 x3 HUDIFF_X3=1
python $R/scripts/adv_report.py $OUT/adversarial_errors.json > $OUT/adv.log 2>&1
bash $R/scripts/x3_ab.sh HUDIFF_X3_LNSYNC "0 2" 1 > $OUT/lnsync_ab.txt 2>&1
echo "$HEAD" > $OUT/GIT_HEAD
ls -la $OUT
cut -c1-400 $OUT/bench_ab256.json
