#!/bin/bash
# Regenerates everything under profiles/r05 from ONE GPU box (run through gpurun; results land in gpurun_out/r05):
#   bash scripts/refresh_profiles_r05.sh <git-head>
# Precision routes and tuning options are chosen on the command line / through the interface; counter passes (inside bench.py) carry
# --kernel-trace only.
HEAD=${1:-unknown}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r05
rm -rf $OUT; mkdir -p $OUT
python $R/bench.py --steps 3 --warmup 1 2>$OUT/bench_ab.err | tail -1 > $OUT/bench_ab256.json
grep "^\[bench\]" $OUT/bench_ab.err > $OUT/bench_ab256_phases.txt
python $R/bench.py --kind nb --steps 3 --warmup 1 2>$OUT/bench_nb.err | tail -1 > $OUT/bench_nb256.json
python $R/bench.py --kind nb --mode inpaint --steps 3 --warmup 1 --no-cpu-baseline --no-evidence 2>>$OUT/bench_nb.err | tail -1 > $OUT/bench_nb256_inpaint.json
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
prof ab256_split_twolaunch_maxt6_lanes1 ab split 1 HUDIFF_FUSED_ATTN=0
prof ab256_f32all_maxt6_lanes1 ab f32_all 1
prof nb256_split_maxt6_lanes1 nb split 1
prof nb256_f32all_maxt6_lanes1 nb f32_all 1
# dispatch sequence of one denoiser step + per-launch roofline budget (scripts/launch_budget.py, HBM priced at 6.29 TB/s)
seq() {   # name, budget route, precision route
  name=$1; broute=$2; route=$3
  CMD="python $R/bench.py --steps 1 --warmup 0 --max-t 4 --no-cpu-baseline --lanes 1 --pmc off --only-main --precision $route"
  timeout 400 rocprofv3 --kernel-trace -d $OUT/sq_$name -o t -- $CMD > $OUT/sq_$name.log 2>&1
  python $R/scripts/rocpd_summary.py $(find $OUT/sq_$name -name "*.db" | head -1) --sequence > $OUT/${name}_step_sequence.txt
  python $R/scripts/launch_budget.py $OUT/${name}_step_sequence.txt $broute > $OUT/${name}_launch_budget.txt
  rm -rf $OUT/sq_$name $OUT/sq_$name.log
}
seq ab256_split x3 split
seq ab256_f32all allfp32 f32_all
# fused attention kernel: ablations of its phases (HUDIFF_QA_ABL; us per launch)
for a in 0 1 2 3 4 8 12 15; do
  CMD="python $R/bench.py --steps 1 --warmup 0 --max-t 4 --no-cpu-baseline --lanes 1 --pmc off --only-main"
  env HUDIFF_QA_ABL=$a timeout 300 rocprofv3 --kernel-trace -d $OUT/qa_$a -o t -- $CMD > $OUT/qa_$a.log 2>&1
  echo "HUDIFF_QA_ABL=$a: $(python $R/scripts/rocpd_summary.py $(find $OUT/qa_$a -name "*.db" | head -1) --by-grid | grep qkv_attn | head -1)" >> $OUT/fused_attention_ablations.txt
  rm -rf $OUT/qa_$a $OUT/qa_$a.log
done
python $R/scripts/adv_report.py $OUT/adversarial_errors.json > $OUT/adv.log 2>&1
# small batches (the reference CLI's default batch_size is 1): full samples per second, split and f32_all routes, antibody and nanobody
for route in split f32_all; do for kind in ab nb; do for B in 1 2 4 8 16 32 64 128; do
  v=$(python $R/bench.py --kind $kind --batch $B --steps 2 --warmup 1 --no-cpu-baseline --pmc off --only-main --precision $route 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'])")
  echo "route $route kind $kind B $B : $v sequences/s" >> $OUT/small_batches.txt
done; done; done
python $R/scripts/small_batch_times.py ab "1 8" $OUT/small_batch_times.txt > /dev/null 2>&1
python $R/scripts/small_batch_times.py nb "1 8" $OUT/small_batch_times.txt > /dev/null 2>&1
(python $R/scripts/cli_e2e.py 1; python $R/scripts/cli_e2e.py 4) 2>/dev/null | grep "end to end" > $OUT/cli_e2e.txt
# soak: complete 256-row samples under kernel-selecting options -- one token digest per model is the claim (options change kernels, not tokens)
for kind in ab nb; do
  for v in "HUDIFF_QUIET=0" "HUDIFF_FUSED_ATTN=0" "HUDIFF_LANES=1" "HUDIFF_LANES=3" "HUDIFF_X3_LNSYNC=0" "HUDIFF_TAIL=0" "HUDIFF_PRECISION=f32_all"; do
    echo "$v : $(env $v python $R/scripts/lnsync_soak.py $kind 2 2>&1 | tail -1)" >> $OUT/option_soak.txt
  done
done
python $R/scripts/spill_report.py --all > $OUT/kernel_resources.txt 2>&1
echo "$HEAD" > $OUT/GIT_HEAD
ls -la $OUT
cut -c1-300 $OUT/bench_ab256.json
