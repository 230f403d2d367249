"""Maximum sizes: one call with more rows than the 32-bit-offset kernels address.

The hot GEMMs and attn_x3_k address their operands with 32-bit byte offsets (operands < 2 GiB per lane: up to ~1 200 antibody rows
per lane); beyond that the library takes the generic 64-bit-address GEMMs (`gemm_k<..., BK = 32>`) and the fp32 attention kernel.
6 000 antibody rows in ONE call (3 000 per lane: the A operand of the Q|K|V projection alone is 2.7 GB, its output 5.4 GB) must give
the tokens of the same rows sampled 256 at a time -- on the default route and with HUDIFF_X3=1 (whose launches then fall back to
the fp32 kernels by eligibility, not by the range guard).
"""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kind,rows,x3", [("ab", 6000, "0"), ("ab", 6000, "1"), ("nb", 12000, "0")])
def test_one_call_beyond_the_32_bit_offset_kernels(kind, rows, x3):
    env = dict(os.environ, HUDIFF_X3=x3)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "big_batch_probe.py"), kind, str(rows), "2"], env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "rows that differ: 0" in r.stdout, r.stdout[-1500:] + r.stderr[-2000:]
