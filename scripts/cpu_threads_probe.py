"""How many host threads the CPU baseline should use on this box (scripts only)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import hudiff_oracle as ho, hudiff_oracle_torch as hot
from hudiff_amd import synthetic as S
cfg = dict(S.AB_CONFIG); sd = S.random_state_dict("ab", cfg, seed=0)
b = S.synthetic_batch("ab", 16, seed=2023, mode="finetune")
net = hot.TorchOracleNet("ab", cfg, sd)
print("cpus", os.cpu_count(), "torch threads default", torch.get_num_threads())
for n in (8, 16, 32, 64, 128):
    torch.set_num_threads(n)
    T = np.minimum(b["T"], 1)
    ho.sample(net, b["tokens"], b["region"], b["chain"], b["order"], T, seed=1, dropout_mode="philox")
    t0 = time.perf_counter()
    ho.sample(net, b["tokens"], b["region"], b["chain"], b["order"], np.minimum(b["T"], 3), seed=1, dropout_mode="philox")
    dt = (time.perf_counter() - t0) / 3
    t1 = time.perf_counter()
    ho.sample(net, b["tokens"], b["region"], b["chain"], b["order"], np.minimum(b["T"], 3), seed=1, dropout_mode="off")
    dt_off = (time.perf_counter() - t1) / 3
    print(f"threads {n:4d}: {dt:.2f} s per step with philox dropout ({16 / (dt * 155):.3f} seq/s), {dt_off:.2f} s without mask generation")
