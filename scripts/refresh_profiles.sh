#!/bin/bash
# Regenerates everything under profiles/rNN from one GPU box (run through gpurun; results land in gpurun_out/final).
#   bench lines (Ab, Nb), rocprofv3 kernel stats of a 6-step one-lane run, HBM-side traffic (FETCH_SIZE / WRITE_SIZE
#   in separate --pmc passes), MFMA-busy, and the VALU-vs-MFMA instruction table.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/final
rm -rf $OUT; mkdir -p $OUT
python $R/bench.py --steps 2 --warmup 1 2>$OUT/bench_ab.err | tail -1 > $OUT/bench_ab256.json
python $R/bench.py --kind nb --steps 2 --warmup 1 2>$OUT/bench_nb.err | tail -1 > $OUT/bench_nb256.json
CMD="python $R/bench.py --steps 1 --warmup 0 --max-t 6 --no-cpu-baseline --lanes 1"
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
print(dict(tot))
PY
$R/scripts/valu_profile.sh > $OUT/valu_vs_mfma.txt 2>&1
rm -rf $OUT/stats $OUT/trace $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmc_SQ_* $OUT/pmc_TCC_*
ls -la $OUT
cat $OUT/bench_ab256.json | cut -c1-400
