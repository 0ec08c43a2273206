import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import aswstereomatch_amd as asw
from oracle import asw_oracle
asw_oracle.build()
ctx = asw.Context(0)
H, W, win = 600, 4000, 15
L = np.zeros((H, W, 3), np.uint8); R = np.zeros((H, W, 3), np.uint8)
L[:] = (np.arange(H) % 50 * 5)[:, None, None]
rc, want = asw_oracle.cost_sad(L, R, 0, win, 0, 20)
got = np.stack(ctx.getCostSAD(L, R, 0, win, 0, 20))
print(((got[0, :140, 20] - want[0, :140, 20]) * 225 / 15 / 5).round(2).tolist())
