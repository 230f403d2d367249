#!/bin/bash
# usage: smi_watch.sh <tag> <command...>: runs the command while polling rocm-smi; prints min / median / max of sclk and power
# over the samples taken while the GPU drew more than 500 W.
tag=$1; shift
( while true; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E 'sclk|Power \(W\)' | sed -E 's/.*\(([0-9]+)Mhz\).*/\1/; s/.*\(W\): ([0-9.]+).*/\1/' | tr '\n' ' '; echo; sleep 0.05; done ) > /tmp/smi_$tag.log &
P=$!
"$@"
kill $P 2>/dev/null
python3 - "$tag" <<'PY'
import sys, statistics as st
rows = [l.split() for l in open(f"/tmp/smi_{sys.argv[1]}.log") if len(l.split()) == 2]
rows = [(float(a), float(b)) for a, b in rows if float(b) > 500]
if rows:
    c = sorted(r[0] for r in rows); w = sorted(r[1] for r in rows)
    print(f"SMI {sys.argv[1]}: {len(rows)} busy samples; sclk MHz min/med/max {c[0]:.0f}/{st.median(c):.0f}/{c[-1]:.0f}; power W min/med/max {w[0]:.0f}/{st.median(w):.0f}/{w[-1]:.0f}")
PY
