import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from dust_amd import api
ctx = api.Context(device=0)
rng = np.random.default_rng(1)
cost = rng.integers(20000, 400000, 32400).astype(np.uint32).reshape(-1, 1)
for _ in range(10): ctx.device_eval(13, cost, 1)
for _ in range(10): ctx.device_eval(14, cost, 2)
