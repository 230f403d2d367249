#!/bin/bash
# After `gpurun -- 'bash scripts/refresh_profiles.sh <HEAD>; bash scripts/x3_epi_pmc.sh > gpurun_out/final/x3_epilogue_valu.txt; bash scripts/x3_pmc.sh >
# gpurun_out/final/x3_pmc.txt; HUDIFF_X3=1 bash scripts/nb_profile.sh; bash scripts/nb_profile.sh'`: copies the summaries that are cited into
# profiles/r02/ (gpurun_out/ is scratch) and regenerates the numbers quoted in profiles/README.md and DESIGN.md from them.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
F=$R/gpurun_out/final
for f in GIT_HEAD ab256_maxt6_lanes1_by_grid.txt ab256_maxt6_lanes1_kernel_stats.csv bench_ab256.json bench_nb256.json bench_nb256_inpaint.json \
         clock_power.txt pmc_totals_6_steps.json pmc_traffic.json valu_vs_mfma.txt x3_ab256_maxt6_lanes1_by_grid.txt \
         x3_ab256_maxt6_lanes1_kernel_stats.csv x3_eval_ab.json x3_eval_nb.json x3_probe_qkv.txt x3_epilogue_valu.txt x3_pmc.txt; do
  cp "$F/$f" "$R/profiles/r02/"
done
cp "$R/gpurun_out/nb/nb256_maxt6_lanes1_by_grid.txt" "$R/gpurun_out/nb/nb256_maxt6_lanes1_by_grid_x3.txt" "$R/profiles/r02/"
python "$R/scripts/update_profile_docs.py"
