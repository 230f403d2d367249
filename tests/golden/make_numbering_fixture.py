"""Regenerates tests/golden/numbering_vectors.json (run in the build container, where /root/reference exists).

Inputs: raw sequences of the reference's evaluation sets (data files, not code): all 25 Humab25 mouse pairs,
every 12th HuAb348 row, every 10th VHH of abnativ_select_vhh.csv.  Expected values: the IMGT-gapped strings
hudiff_amd/numbering.py produces for them TODAY (as 152 / 139-slot strings) -- regression vectors of this module, NOT ANARCI output
(ANARCI is not installable here; parity with it is unpinned).
"""
import json
import os
import sys

import pandas as pd

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hudiff_amd import numbering as N   # noqa: E402

R = "/root/reference/data/"


def gapped(seq):
    """-> chain type, the residues laid into the model's 152 / 139 IMGT slots, numbered residues outside them."""
    from hudiff_amd import inputs as I
    from hudiff_amd import tables as T
    d, cls = N.number_sequence_builtin(seq)
    table = T.HEAVY_POSITIONS_dict if cls == "H" else T.LIGHT_POSITIONS_dict
    slots = "".join(I.slot_residues(d, "H" if cls == "H" else "L"))
    extra = {k: v for k, v in d.items() if k not in table and v != "-"}
    return cls, slots, extra


rows = []
df = pd.read_csv(R + "antibody_eval_data/Humab25_data/parental_mouse.csv")
for r in df.itertuples():
    rows += [("humab25/" + r.name + "/H", r.h_seq), ("humab25/" + r.name + "/L", r.l_seq)]
df = pd.read_csv(R + "antibody_eval_data/HuAb348_data/humanization_pair_data_filter.csv")
for r in list(df.itertuples())[::12]:
    rows += [("huab348/" + str(r.name) + "/H", r.h_seq), ("huab348/" + str(r.name) + "/L", r.l_seq)]
df = pd.read_csv(R + "nanobody_eval_data/abnativ_select_vhh.csv")
for i, s in list(enumerate(df["vhhseq"]))[::10]:
    rows.append((f"vhh/{i}", s))
out = []
for name, seq in rows:
    cls, slots, extra = gapped(seq)
    out.append({"name": name, "seq": seq, "chain": cls, "slots": slots, "extra": extra})
json.dump(out, open(os.path.join(ROOT, "tests", "golden", "numbering_vectors.json"), "w"), indent=0)
print(len(out), "vectors")
