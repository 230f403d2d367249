#!/bin/bash
# Regenerates everything under profiles/rNN from ONE GPU box (run through gpurun; results land in gpurun_out/final):
#   bash scripts/refresh_profiles.sh <git-head>          (the head is stamped into the files; the box has no .git)
#   bench lines (Ab on HuAb348, Nb on the VHH set; each with its own live PMC traffic passes), rocprofv3 kernel stats of a
#   6-step one-lane run, HBM-side traffic (FETCH_SIZE / WRITE_SIZE in separate --pmc passes), MFMA-busy, L2 hit rate, and
#   the VALU-vs-MFMA instruction table.  Counter passes carry --kernel-trace only (no other tracing domain).
HEAD=${1:-unknown}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/final
rm -rf $OUT; mkdir -p $OUT
python $R/bench.py --steps 2 --warmup 1 2>$OUT/bench_ab.err | tail -1 > $OUT/bench_ab256.json
python $R/bench.py --kind nb --steps 2 --warmup 1 2>$OUT/bench_nb.err | tail -1 > $OUT/bench_nb256.json
python $R/bench.py --kind nb --mode inpaint --steps 2 --warmup 1 --no-cpu-baseline --traffic off 2>>$OUT/bench_nb.err | tail -1 > $OUT/bench_nb256_inpaint.json
CMD="python $R/bench.py --steps 1 --warmup 0 --max-t 6 --no-cpu-baseline --lanes 1 --traffic off"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o t -- $CMD > $OUT/stats.log 2>&1
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/ab256_maxt6_lanes1_kernel_stats.csv
timeout 400 rocprofv3 --kernel-trace -d $OUT/trace -o t -- $CMD > $OUT/trace.log 2>&1
python $R/scripts/rocpd_summary.py $(find $OUT/trace -name "*.db" | head -1) --by-grid > $OUT/ab256_maxt6_lanes1_by_grid.txt
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $c | tr ' ' '_')
  timeout 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$n -o p -- $CMD > $OUT/pmc_$n.log 2>&1
done
python3 - <<PY
import csv, glob, json, collections
tot = collections.defaultdict(float)
for f in glob.glob("$OUT/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        tot[r["Counter_Name"]] += float(r["Counter_Value"])
json.dump(tot, open("$OUT/pmc_totals_6_steps.json", "w"), indent=1)
line = json.load(open("$OUT/bench_ab256.json"))
rd, wr = 2.0 * tot["FETCH_SIZE"] * 1024 / 6, tot["WRITE_SIZE"] * 1024 / 6
json.dump({"git_head": "$HEAD",
           "config": {"kind": "ab", "rows_per_gpu": 256, "dropout": "faithful", "lanes": 1},
           "method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (scripts/refresh_profiles.sh, same box "
                     "and run as bench_ab256.json) over 'python bench.py --steps 1 --warmup 0 --max-t 6 --no-cpu-baseline --lanes 1': counters "
                     "summed over every kernel of the process and divided by 6 steps; FETCH_SIZE doubled (gfx950 tallies 128-B requests at "
                     "64 B, MI355X_MICROARCH.md HBM section); KB -> bytes; L2<->fabric side, Infinity-Cache hits included",
           "read_bytes_per_launch": rd, "write_bytes_per_launch": wr, "traffic_bytes_per_launch": rd + wr,
           "l2_hit_rate": tot["TCC_HIT_sum"] / max(tot["TCC_HIT_sum"] + tot["TCC_MISS_sum"], 1.0),
           "mfma_busy_fraction_profiled_run": tot["SQ_VALU_MFMA_BUSY_CYCLES"] / max(tot["GRBM_GUI_ACTIVE"] / 8 * 1024, 1.0),
           "bench_line_live_traffic": line.get("roofline", {}).get("traffic"),
           "bench_line_value": line.get("value")}, open("$OUT/pmc_traffic.json", "w"), indent=1)
print(dict(tot))
PY
$R/scripts/valu_profile.sh > $OUT/valu_vs_mfma.txt 2>&1
# ---- split-precision prototype (HUDIFF_X3=1): per-kernel times of the same 6-step run, error / token agreement report
HUDIFF_X3=1 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/x3stats -o t -- $CMD > $OUT/x3stats.log 2>&1
cp $(find $OUT/x3stats -name "*kernel_stats.csv" | head -1) $OUT/x3_ab256_maxt6_lanes1_kernel_stats.csv
HUDIFF_X3=1 timeout 400 rocprofv3 --kernel-trace -d $OUT/x3trace -o t -- $CMD > $OUT/x3trace.log 2>&1
python $R/scripts/rocpd_summary.py $(find $OUT/x3trace -name "*.db" | head -1) --by-grid > $OUT/x3_ab256_maxt6_lanes1_by_grid.txt
python $R/scripts/x3_eval.py ab 2>/dev/null | tail -1 > $OUT/x3_eval_ab.json
python $R/scripts/x3_eval.py nb 2>/dev/null | tail -1 > $OUT/x3_eval_nb.json
rm -rf $OUT/x3stats $OUT/x3trace
# ---- clock / power while sampling (rocm-smi polled at ~10 Hz) and the isolated Q|K|V split-precision launch (scripts/x3_probe.hip)
B2="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --traffic off --no-split-line"
{ HUDIFF_X3=0 bash $R/scripts/smi_watch.sh f32_ab $B2 2>&1 | grep SMI
  HUDIFF_X3=1 bash $R/scripts/smi_watch.sh x3_ab $B2 2>&1 | grep SMI
  HUDIFF_X3=0 bash $R/scripts/smi_watch.sh f32_nb $B2 --kind nb --steps 8 2>&1 | grep SMI
  HUDIFF_X3=1 bash $R/scripts/smi_watch.sh x3_nb $B2 --kind nb --steps 8 2>&1 | grep SMI; } > $OUT/clock_power.txt
[ -x $R/scripts/x3_probe.bin ] && timeout 120 $R/scripts/x3_probe.bin 2>&1 | grep -E "END|MON" > $OUT/x3_probe_qkv.txt
echo "$HEAD" > $OUT/GIT_HEAD
rm -rf $OUT/stats $OUT/trace $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmc_SQ_* $OUT/pmc_TCC_*
ls -la $OUT
cut -c1-600 $OUT/bench_ab256.json
