import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["HUDIFF_ATTN_X3"] = sys.argv[1]
import hudiff_amd
from hudiff_amd import synthetic as S, evalsets as E
cfg = dict(S.AB_CONFIG); sd = S.random_state_dict("ab", cfg, seed=0)
m = hudiff_amd.AntiTFNet(**cfg); m.load_state_dict(sd)
batch = E.eval_batch("huab348", 32, row0=0)
lg = m(batch["tokens"], batch["region"], batch["chain"], dropout="off")
np.save(sys.argv[2], lg)
print(os.environ.get("HUDIFF_LIB"), sys.argv[1], float(np.abs(lg).max()), m.precision_info())
