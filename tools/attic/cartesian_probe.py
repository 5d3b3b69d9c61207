import sys, time
sys.path.insert(0,'/root/repo')
import numpy as np
from firecode_amd.utils import cartesian_product
def ref(*arrays):
    a=[np.asarray(x) for x in arrays]
    return np.stack(np.meshgrid(*a), -1).reshape(-1, len(a))
c=[(0,60,120,180,240,300)]*8
for k in range(3):
    t0=time.perf_counter(); mine=cartesian_product(*c); t1=time.perf_counter(); r=ref(*c); t2=time.perf_counter()
    print("native", round(t1-t0,4), "numpy", round(t2-t1,4), np.array_equal(mine,r))
t0=time.perf_counter(); x=np.empty((1679616,8),dtype=np.int64); x[:]=1; print("first touch of 107 MB", round(time.perf_counter()-t0,4))
