"""End-to-end wall time of the drop-in CLI on the HuAb348 evaluation set: raw sequences in a CSV -> IMGT numbering (built-in
slotter) -> input preparation -> full T-step sampling on the GPU -> similarity search -> CSV / FASTA on disk.
    python scripts/cli_e2e.py [sample_number]
Random-init weights of the production architecture (no checkpoint offline); the 348 mouse pairs are the package's evaluation rows."""
import os, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import hudiff_amd
from hudiff_amd import checkpoint as ck, evalsets as E, synthetic as S, numbering as N
from hudiff_amd.cli import sample as cli

n_samples = int(sys.argv[1]) if len(sys.argv) > 1 else 1
tmp = tempfile.mkdtemp(prefix="hudiff_e2e_")
pairs = E.sequences("huab348")
csv = os.path.join(tmp, "huab348.csv")
with open(csv, "w") as f:
    f.write("type,name,order_name,h_seq,l_seq\n")
    for i, (vh, vl) in enumerate(pairs):
        f.write(f"mouse,ab{i:03d},ab{i:03d},{vh},{vl}\n")
cfg = dict(S.AB_CONFIG)
sd = {k: torch.from_numpy(v) for k, v in S.random_state_dict("ab", cfg, seed=0).items()}
ckpt = os.path.join(tmp, "hudiffab.pt")
torch.save({"fineconfig": ck.EasyDict({}), "pretrain_config": ck.EasyDict({"name": "trans_oadm", "model": cfg}), "model": sd}, ckpt)
t0 = time.time()
for vh, vl in pairs:
    N.number_imgt(vh); N.number_imgt(vl)
t_num = time.time() - t0
os.chdir(tmp)
t0 = time.time()
out = cli.main(["--ckpt", ckpt, "--data_fpath", csv, "--numbering", "builtin", "--sample_number", str(n_samples), "--batch_size", str(n_samples)])
t_all = time.time() - t0
n_rows = sum(1 for ln in open(out) if ln.startswith("humanization"))
print(f"HuAb348 CLI end to end: {len(pairs)} antibodies x {n_samples} sample(s) -> {n_rows} humanized rows in {t_all:.1f} s wall "
      f"({n_rows / t_all:.1f} rows/s; of it ~{t_num:.1f} s IMGT numbering of {2 * len(pairs)} chains on one host core, the rest model build + "
      f"weight upload + sampling + similarity search + CSV / FASTA)")
