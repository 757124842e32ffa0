"""where bench.py spends its wall time at C3: model build, upload, schedule construction, first pass"""
import time, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
t0=time.time()
import torch
print("import torch %.1f"%(time.time()-t0)); t=time.time()
from lp_mp_amd import engine as E, model as M, synthetic as S
import bench
torch.cuda.set_device(0)
sp = torch.cuda.current_stream().cuda_stream
m, const, dual = bench.build_device_grid(torch, 1024, 1024, 32, "dense", "colour_major", 1, E, S, sp)
print("build model %.1f"%(time.time()-t)); t=time.time()
eng = E.Engine(0); eng.set_stream(sp)
eng.upload(m, const_dev=const.data_ptr(), dual_dev=dual.data_ptr(), keep=(const, dual))
print("upload %.1f"%(time.time()-t)); t=time.time()
eng.set_reparametrization(0)
print("set_repam %.1f"%(time.time()-t)); t=time.time()
eng.compute_pass(1); torch.cuda.synchronize()
print("first pass %.1f"%(time.time()-t)); t=time.time()
eng.compute_pass(20); torch.cuda.synchronize()
print("20 passes %.2f"%(time.time()-t)); t=time.time()
print(eng.lower_bound()); print("lb %.2f"%(time.time()-t)); t=time.time()
eng.compute_pass_and_primal(30); torch.cuda.synchronize(); print("primal %.2f"%(time.time()-t)); t=time.time()
print("total %.1f"%(time.time()-t0))
