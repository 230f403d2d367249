#!/bin/bash
# Runs scripts/x3_probe.bin; SMI=1 also polls rocm-smi (clock, power) every ~0.25 s.  Output: gpurun_out/x3_probe.log.
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
if [ -n "$SMI" ]; then
( for i in $(seq 1 200); do echo "SMI $(cut -d' ' -f1 /proc/uptime) $(rocm-smi --showclocks --showpower 2>/dev/null | grep -E 'sclk|Power \(W\)' | sed -E 's/.*\(([0-9]+)Mhz\).*/\1MHz/; s/.*\(W\): ([0-9.]+).*/\1W/' | tr '\n' ' ')"; sleep 0.1; done ) > /tmp/smi.log &
SMIPID=$!
fi
timeout 90 $R/scripts/x3_probe.bin > $R/gpurun_out/x3_probe.log 2>&1
[ -n "$SMI" ] && { kill $SMIPID 2>/dev/null; cat /tmp/smi.log >> $R/gpurun_out/x3_probe.log; }
grep -E "MON|END" $R/gpurun_out/x3_probe.log
