import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lp_mp_amd import engine as E, synthetic as S
import bench
H=W=1024; L=32
sp = torch.cuda.current_stream().cuda_stream
m, const, dual = bench.build_device_grid(torch, H, W, L, "dense", "colour_major", 1, E, S, sp)
for env in ("0","1"):
    os.environ["LPMP_NO_LB_TRACKING"]=env
    e = E.Engine(0); e.set_stream(sp)
    e.upload(m, const_dev=const.data_ptr(), dual_dev=dual.data_ptr(), keep=(const,dual))
    e.set_reparametrization(0); e.compute_pass(2); lb=e.lower_bound(); torch.cuda.synchronize()
    t0=time.perf_counter()
    for i in range(10):
        e.compute_pass(1); lb=e.lower_bound()
    torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/10
    t0=time.perf_counter()
    for i in range(10): lb=e.lower_bound()
    dl=(time.perf_counter()-t0)/10
    print("NO_LB_TRACKING=%s: pass+LB %.3f ms, LB alone %.3f ms, LB=%.6f"%(env, dt*1e3, dl*1e3, lb))
    e.close()
