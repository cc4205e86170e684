import sys; sys.path.insert(0,'/root/repo')
import numpy as np
from dust_amd import api
ctx=api.Context(device=0)
rng=np.random.default_rng(5)
for n in (9, 100, 32400, 129600, 300001):
    cost=rng.integers(1000,500000,n).astype(np.uint32)
    cost[n//3:n//2]//=10
    out=ctx.device_eval(14, np.ascontiguousarray(cost.reshape(-1,1)), 2)
    order=out[:,0]; cuts=out[:9,1]
    assert sorted(order.tolist())==list(range(n)), n
    cs=np.concatenate([[0],np.cumsum(cost.astype(np.int64))])
    sums=[cs[cuts[b+1]]-cs[cuts[b]] for b in range(8)]
    print(n, cuts.tolist(), [round(x/ (cs[-1]/8),3) for x in sums])
    for b in range(8):
        seg=order[cuts[b]:cuts[b+1]]
        assert ((seg>=cuts[b])&(seg<cuts[b+1])).all()
