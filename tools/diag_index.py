import random, sys, collections
import numpy as np, torch
sys.path.insert(0,'.'); sys.path.insert(0,'napkon-string-matching_amd')
from napkon_string_matching_amd import grid, tables
from oracle import native
def _rand_padded(rng, n, width, vocab, kmax, allow_empty):
    ids = np.full((n, width), -1, dtype=np.int32)
    for r in range(n):
        k = rng.randint(0 if allow_empty else 1, kmax)
        ids[r, :k] = rng.sample(range(vocab), k)
    return ids
dev=torch.device('cuda:0')
rng = random.Random(16 * 1000 + 16)
left = _rand_padded(rng, 333, 16, 60, 16, allow_empty=False)
right = _rand_padded(rng, 517, 16, 60, 16, allow_empty=True)
lt = tables.SetTable.from_padded(left, "left", dev, width=16); rt = tables.SetTable.from_padded(right, "right", dev, width=16)
want = native.jaccard_raw(native.csr_from_padded(left), native.csr_from_padded(right), 0.05, cap=1 << 18)
got = grid.jaccard_raw_grid(lt, rt, 0.05, index=True, capacity=1<<18).as_tuples()
cnt=(right>=0).sum(1); order=np.argsort(-cnt,kind='stable'); pos={int(j):k for k,j in enumerate(order)}
lcnt=(left>=0).sum(1); lorder=np.argsort(-lcnt,kind='stable'); lpos={int(i):k for k,i in enumerate(lorder)}
cg=collections.Counter((i,j) for s,i,j in got); cw=set((i,j) for s,i,j in want)
dups=[p for p,c in cg.items() if c>1]; extra=[p for p in cg if p not in cw]; missing=[p for p in cw if p not in cg]
print('got',len(got),'want',len(want),'dups',len(dups),'extra',len(extra),'missing',len(missing))
print('dup tiles',collections.Counter(pos[j]//64 for i,j in dups))
print('dup left rows (sorted pos) sample',sorted(set(lpos[i] for i,j in dups))[:40])
print('extra tiles',collections.Counter(pos[j]//64 for i,j in extra))
