"""the similarity search that picks one replica per antibody (antibody_scripts/sample.py:352-367).

The reference scores each replica by ``cal_all_preservation`` (antibody_scripts/patent_eval.py:150-159): abnumber aligns
the two chains BY NUMBERED POSITION and counts equal positions over the union of positions.  The drop-in scores identity
over the shared IMGT slots (hudiff_amd.inputs.slot_identity) -- the same quantity whenever both chains carry the same
numbering, which holds by construction for a sample (it is written into the parent's slots).  abnumber is not
installable here, so the equivalence is pinned against an independent, numbering-free yard-stick: a plain
Needleman-Wunsch global alignment of the two residue strings (match +1, mismatch -1, gap -1; identity = matches /
alignment length).  On realistic replicas -- HuAb348 parents, framework substitutions at the finetune mask's slots,
occasional deletions ('-' drawn by the sampler) -- both scores must pick the same replica.
"""
import numpy as np

from hudiff_amd import evalsets as E
from hudiff_amd import inputs as I
from hudiff_amd.cli.sample import select_most_similar


def nw_identity(a: str, b: str) -> float:
    """Global alignment (match +1, mismatch -1, gap -1) -> matches / alignment columns."""
    n, m = len(a), len(b)
    A = np.frombuffer(a.encode(), np.uint8)
    B = np.frombuffer(b.encode(), np.uint8)
    S = np.zeros((n + 1, m + 1), np.int32)
    S[:, 0] = -np.arange(n + 1)
    S[0, :] = -np.arange(m + 1)
    for i in range(1, n + 1):
        sub = np.where(B == A[i - 1], 1, -1)
        diag = S[i - 1, :-1] + sub
        up = S[i - 1, 1:] - 1
        row = np.maximum(diag, up)
        for j in range(1, m + 1):                       # the left dependency is sequential
            v = row[j - 1]
            left = S[i, j - 1] - 1
            S[i, j] = v if v >= left else left
    i, j, match, cols = n, m, 0, 0
    while i > 0 or j > 0:
        if i > 0 and j > 0 and S[i, j] == S[i - 1, j - 1] + (1 if a[i - 1] == b[j - 1] else -1):
            match += a[i - 1] == b[j - 1]
            i, j = i - 1, j - 1
        elif i > 0 and S[i, j] == S[i - 1, j] - 1:
            i -= 1
        else:
            j -= 1
        cols += 1
    return match / max(cols, 1)


def test_slot_identity_picks_the_replica_a_global_alignment_picks():
    z = E.load_rows()
    rng = np.random.default_rng(7)
    n_cases, agree, exact_score = 0, 0, 0
    for a in range(0, 348, 3):                                           # 116 antibodies
        parent = z["huab348_tokens"][a].astype(np.int64)
        _, _, _, loc = I.antibody_row_from_tokens(parent, int(z["huab348_lchain"][a]), finetune=True)
        replicas = []
        for r in range(6):
            row = parent.copy()
            k = int(rng.integers(5, 60))                                 # framework slots that came out different
            hit = rng.choice(loc, size=k, replace=False)
            row[hit] = rng.integers(0, 20, size=k)
            if rng.random() < 0.3:                                       # the sampler may draw '-' (a deletion) or X
                gone = rng.choice(loc, size=int(rng.integers(1, 3)), replace=False)
                row[gone] = 21
            if rng.random() < 0.1:
                row[rng.choice(loc)] = 20
            replicas.append(row)
        pick = select_most_similar(parent, replicas)
        ph, pl = I.untokenize_antibody(parent)
        scores = []
        for row in replicas:
            h, l = I.untokenize_antibody(row)
            scores.append(0.5 * (nw_identity(ph, h) + nw_identity(pl, l)))
        want = int(np.argmax(scores))                                    # first maximum, as list.index(max(...)) in the reference
        n_cases += 1
        agree += pick == want
        slot = 0.5 * (I.slot_identity(parent[:152], replicas[pick][:152]) + I.slot_identity(parent[152:], replicas[pick][152:]))
        exact_score += abs(slot - scores[pick]) < 1e-12
    assert n_cases >= 100
    assert agree == n_cases, (agree, n_cases)
    assert exact_score >= 0.8 * n_cases        # without indels the two identities are the same number, not just the same ranking
