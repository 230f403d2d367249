#!/bin/bash
# per-kernel (by grid) table of a 6-step TWO-lane run on the default route: scripts/x3_bygrid2.sh OUTNAME [ENV=VAL ...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$1; shift
mkdir -p $OUT
CMD="python $R/bench.py --steps 1 --warmup 0 --max-t 6 --no-cpu-baseline --lanes 2 --pmc off"
env "$@" timeout 400 rocprofv3 --kernel-trace -d $OUT/trace -o t -- $CMD > $OUT/trace.log 2>&1
python $R/scripts/rocpd_summary.py $(find $OUT/trace -name "*.db" | head -1) --by-grid > $OUT/by_grid.txt
python - <<PY
import sqlite3,glob
db=glob.glob("$OUT/trace/**/*.db",recursive=True)[0]
c=sqlite3.connect(db)
rows=c.execute("select start,end from kernels order by start").fetchall()
# wall time covered by at least one kernel, and total
t0=rows[0][0]; t1=max(r[1] for r in rows)
print("wall %.1f ms, sum of kernel durations %.1f ms" % ((t1-t0)/1e6, sum(e-s for s,e in rows)/1e6))
PY
rm -rf $OUT/trace
head -14 $OUT/by_grid.txt
