"""world_size = 2 on CPU (gloo): rows shard by global id, one gather at the end, result equals one process."""
import os
import subprocess
import sys

import numpy as np

from conftest import ROOT
from hudiff_amd import dist as D

WORKER = r'''
import os, sys
import numpy as np
sys.path.insert(0, os.environ["HD_ROOT"]); sys.path.insert(0, os.path.join(os.environ["HD_ROOT"], "tests"))
from hudiff_amd import dist as D
from hudiff_amd.sampler import sample_jobs
from test_host_logic import FakeModel, _jobs
D.init_process_group("gloo")
rank, world, _ = D.env_rank_world()
res = sample_jobs(FakeModel(), _jobs(7), replicas=3, seed=11, device_batch=5)
if rank == 0:
    np.save(os.environ["HD_OUT"], res)
else:
    assert res is None
# accept / re-sweep loop: every rank takes the same decisions from all-gathered rows
from hudiff_amd.sampler import sample_jobs_with_retry
jobs = _jobs(5)
probe = int(jobs[0].loc[0])
rows = sample_jobs_with_retry(FakeModel(), jobs, 2, 7, want=2, tries=5, accept=lambda r: int(r[probe]) % 3 == 0, device_batch=3)
np.save(os.environ["HD_OUT"] + f".retry{rank}.npy", np.array([len(r) for r in rows] + [int(np.sum([x.sum() for x in r])) for r in rows]))
import torch.distributed as dist
dist.barrier(); dist.destroy_process_group()
'''


def test_shard_bounds_cover_rows_exactly():
    for n in (0, 1, 7, 21, 256, 2048):
        for world in (1, 2, 3, 8):
            b = [D.shard_bounds(n, r, world) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1


def test_two_ranks_equal_one_process(tmp_path):
    from hudiff_amd.sampler import sample_jobs
    from test_host_logic import FakeModel, _jobs
    want = sample_jobs(FakeModel(), _jobs(7), replicas=3, seed=11, device_batch=5)
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    out = tmp_path / "res.npy"
    env = dict(os.environ, HD_ROOT=ROOT, HD_OUT=str(out), OMP_NUM_THREADS="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", str(script)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    got = np.load(out)
    assert got.shape == want.shape == (7, 1, 3, 291) and np.array_equal(got, want)
    from hudiff_amd.sampler import sample_jobs_with_retry
    jobs = _jobs(5)
    probe = int(jobs[0].loc[0])
    rows = sample_jobs_with_retry(FakeModel(), jobs, 2, 7, want=2, tries=5, accept=lambda r: int(r[probe]) % 3 == 0, device_batch=3)
    single = np.array([len(r) for r in rows] + [int(np.sum([x.sum() for x in r])) for r in rows])
    for rank in (0, 1):
        assert np.array_equal(np.load(str(out) + f".retry{rank}.npy"), single)
