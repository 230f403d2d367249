#!/bin/bash
# PMC passes over selected kernels of the default route (counters only + kernel trace): scripts/attn_pmc.sh OUTNAME [ENV=VAL ...]
# PMC_FILTER=regex selects the kernels (default: the attention core)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$1; shift
rm -rf $OUT; mkdir -p $OUT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" \
           "SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  env "$@" timeout 300 rocprofv3 --pmc $set --kernel-trace -d $OUT/p$i -o p --output-format csv -- \
      python $R/bench.py --steps 1 --warmup 0 --max-t 2 --no-cpu-baseline --lanes 1 --no-graph --pmc off --only-main > $OUT/p$i.log 2>&1
done
python3 - <<PY > $OUT/attn_pmc.txt
import csv, glob, collections, re, os
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
for f in sorted(glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if not re.search(os.environ.get("PMC_FILTER", "attn_x3_k|attn_k<"), r["Kernel_Name"]):
            continue
        k = re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0] + " g=" + r["Grid_Size"]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
for k in sorted(acc, key=lambda k: -acc[k].get("SQ_WAVE_CYCLES", 0)):
    print(k)
    for c, v in sorted(acc[k].items()):
        print(f"    {c:32s} {v / max(n[k][c], 1):16.0f} per dispatch ({n[k][c]} dispatches)")
PY
cat $OUT/attn_pmc.txt
rm -rf $OUT/p*/
