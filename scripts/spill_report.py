"""Per-kernel register / LDS / scratch usage of the library (hipcc -Rpass-analysis=kernel-resource-usage), listing every
kernel that spills.   python scripts/spill_report.py [--all]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "hudiff_amd", "csrc", "hd_api.hip")
r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-c", "-mllvm", "-pragma-unroll-threshold=200000", "-Rpass-analysis=kernel-resource-usage",
                    src, "-o", "/tmp/hd_api_spill.o"], capture_output=True, text=True)
txt = r.stderr + r.stdout
blocks = re.split(r"remark: [^\n]*Function Name: ", txt)[1:]
n_bad = 0
for b in blocks:
    name = subprocess.run(["c++filt", b.split()[0]], capture_output=True, text=True).stdout.strip()
    g = lambda k: int(re.search(k + r": (\d+)", b).group(1))
    vg, ag, sp, sc, lds, occ = g("VGPRs"), g("AGPRs"), g("VGPRs Spill"), g(r"ScratchSize \[bytes/lane\]"), g(r"LDS Size \[bytes/block\]"), g(r"Occupancy \[waves/SIMD\]")
    bad = sp > 0 or sc > 0
    n_bad += bad
    if bad or "--all" in sys.argv:
        print(f"{'SPILL ' if bad else '      '}{name[:120]:120s} vgpr {vg} agpr {ag} spill {sp} scratch {sc} lds {lds} occ {occ}")
print(f"{len(blocks)} kernels, {n_bad} with spills / scratch")
