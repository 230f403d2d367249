#!/bin/bash
# PMC passes on the production QKV GEMM kernel (scripts/gemm_probe.bin pc); counters only, no tracing domains
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_probe
mkdir -p $OUT
i=0
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS" \
           "SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS" \
           "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAVE_CYCLES" \
           "SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_BUSY_CU_CYCLES SQ_CYCLES" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_CYCLES_VALU SQ_ACTIVE_INST_VALU" \
           "SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_IFETCH" \
           "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_IFETCH_LEVEL SQ_LEVEL_WAVES"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $set --kernel-trace -d $OUT/p$i -o p --output-format csv -- $GRAFT_REPO_ROOT/scripts/gemm_probe.bin pc > $OUT/p$i.log 2>&1
  tail -1 $OUT/p$i.log
done
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(float); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        if "gemm_k" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    print(f.split("/")[-3] if "/" in f else f, {k: round(v / n[k]) for k, v in acc.items()}, "dispatches", max(n.values()) if n else 0)
PY
