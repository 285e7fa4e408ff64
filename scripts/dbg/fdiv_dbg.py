import sys; sys.path.insert(0,'/root/repo')
import numpy as np
from desman_amd import _lib
rs = np.random.RandomState(5)
ends = [2.0 ** e for e in (-1074, -1060, -1022, -900, -600, -501, -499, -300, -1, 0, 1, 300, 499, 501, 600, 900, 1022, 1023)]
a = np.array([x * (1.0 + rs.rand()) for x in ends for _ in ends])
b = np.array([y * (1.0 + rs.rand()) for _ in ends for y in ends])
a = np.minimum(a, np.finfo(np.float64).max); b = np.minimum(b, np.finfo(np.float64).max)
with np.errstate(all="ignore"): ref = a / b
got = _lib.debug_fdiv(0, a, b)
w = np.where(np.isinf(got) != np.isinf(ref))[0]
for i in w: print(i, a[i].hex(), b[i].hex(), got[i], ref[i])
w = np.where(np.isnan(got) != np.isnan(ref))[0]
for i in w: print("nan", i, a[i].hex(), b[i].hex(), got[i], ref[i])
