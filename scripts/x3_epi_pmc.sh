#!/bin/bash
# Vector-ALU instructions per epilogue feature of gemm_x3_k (one launch per configuration of scripts/x3_probe.bin pmc).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/epipmc; rm -rf $OUT; mkdir -p $OUT
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/p -o p -- $R/scripts/x3_probe.bin pmc > $OUT/run.log 2>&1
python3 - <<PY
import csv, glob, collections
names = [l[4:].strip() for l in open("$OUT/run.log") if l.startswith("PMC ")]
rows = collections.defaultdict(dict)
for f in glob.glob("$OUT/p/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm_x3_k" in r["Kernel_Name"]:
            rows[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
elems = 74496 * 768
for i, d in enumerate(sorted(rows)):
    c = rows[d]
    print(f"{names[i] if i < len(names) else d:75s} VALU {c.get('SQ_INSTS_VALU', 0) * 64 / elems:6.2f} / element  SALU {c.get('SQ_INSTS_SALU', 0) * 64 / elems:5.2f}  LDS {c.get('SQ_INSTS_LDS', 0) * 64 / elems:5.2f}  MFMA {c.get('SQ_INSTS_MFMA', 0):.0f}  wave cycles {c.get('SQ_WAVE_CYCLES', 0):.3g}  busy {c.get('SQ_BUSY_CYCLES', 0):.3g}")
PY
