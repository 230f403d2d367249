#!/bin/bash
# PMC passes over the split-precision GEMM kernels (counters only + kernel trace): where do the cycles of gemm_x3_k go?
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/x3pmc
rm -rf $OUT; mkdir -p $OUT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL" \
           "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  HUDIFF_X3=1 timeout 300 rocprofv3 --pmc $set --kernel-trace -d $OUT/p$i -o p --output-format csv -- \
      python $R/bench.py --steps 1 --warmup 0 --max-t 2 --no-cpu-baseline --lanes 1 --no-graph --traffic off > $OUT/p$i.log 2>&1
done
python3 - <<PY
import csv, glob, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
for f in sorted(glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if not any(s in r["Kernel_Name"] for s in ("gemm_x3_k", "attn_k", "attn_x3_k", "ln_apply_k")):
            continue
        k = re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0] + " g=" + r["Grid_Size"]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
for k in sorted(acc, key=lambda k: -acc[k].get("SQ_WAVE_CYCLES", 0)):
    print(k)
    for c, v in sorted(acc[k].items()):
        print(f"    {c:32s} {v / max(n[k][c], 1):16.0f} per dispatch ({n[k][c]} dispatches)")
PY
rm -rf $OUT/p*/
