#!/bin/bash
# per (kernel, grid) PMC averages per launch of a short one-lane run, one rocprofv3 pass per counter group (FETCH_SIZE and WRITE_SIZE do
# not fit one pass):  scripts/r05/pmc_bygrid.sh OUT [bench args ...]       FETCH_SIZE is doubled (gfx950 correction, MI355X_MICROARCH.md)
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$1; shift
mkdir -p $OUT
CMD="python $R/bench.py --steps 1 --warmup 0 --max-t 3 --no-cpu-baseline --lanes 1 --pmc off --only-main $@"
i=0
for CNT in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d $OUT/pmc_$i -o p -- $CMD > $OUT/pmc_$i.log 2>&1
done
python - "$OUT" <<'PY'
import csv, glob, os, sys
d = sys.argv[1]
agg, calls = {}, {}
for f in glob.glob(os.path.join(d, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        key = (name, r.get("Grid_Size", "?"))
        c = r["Counter_Name"]
        agg.setdefault(key, {}).setdefault(c, 0.0)
        agg[key][c] += float(r["Counter_Value"])
        k2 = (key, c, r.get("Dispatch_Id"))
        if k2 not in seen:
            seen.add(k2)
            calls.setdefault(key, {}).setdefault(c, 0)
            calls[key][c] += 1
rows = []
for key, cs in agg.items():
    n = max(calls[key].values())
    g = {c: v / calls[key][c] for c, v in cs.items()}
    fetch = 2 * g.get("FETCH_SIZE", 0) * 1024 / 1e6      # KB -> MB, doubled
    write = g.get("WRITE_SIZE", 0) * 1024 / 1e6
    hit, miss = g.get("TCC_HIT_sum", 0), g.get("TCC_MISS_sum", 0)
    rows.append((fetch + write, key, n, fetch, write, hit / max(hit + miss, 1), g.get("TCC_EA0_RDREQ_sum", 0), g.get("TCC_EA0_RDREQ_32B_sum", 0)))
print(f"{'kernel grid':78s} {'calls':>5s} {'fetch MB':>9s} {'write MB':>9s} {'L2 hit':>7s} {'rdreq':>10s} {'rdreq32':>10s}   (averages per launch)")
for tot, key, n, fetch, write, hr, rq, rq32 in sorted(rows, reverse=True)[:24]:
    print(f"{(key[0][:60] + ' ' + key[1]):78s} {n:5d} {fetch:9.1f} {write:9.1f} {hr:7.3f} {rq:10.3g} {rq32:10.3g}")
PY
rm -rf $OUT/pmc_[0-9]
