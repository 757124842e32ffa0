"""per-iteration cost of a Solve()-style loop (one pass + LowerBound per iteration) on C2 and C3"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lp_mp_amd import engine as E, synthetic as S
import bench
sp = torch.cuda.current_stream().cuda_stream
for name, (H, L, pw) in {"C2": (512, 8, "potts"), "C3": (1024, 32, "dense")}.items():
    m, const, dual = bench.build_device_grid(torch, H, H, L, pw, "colour_major", 1, E, S, sp)
    e = E.Engine(0); e.set_stream(sp)
    e.upload(m, const_dev=const.data_ptr(), dual_dev=dual.data_ptr(), keep=(const, dual))
    e.set_reparametrization(0); e.compute_pass(2); e.lower_bound(); torch.cuda.synchronize()
    def timed(f, n=50):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): f()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
    t_pass = timed(lambda: e.compute_pass(1))
    t_both = timed(lambda: (e.compute_pass(1), e.lower_bound()))
    t_lb = timed(lambda: e.lower_bound())
    # the same loop with passes that may run ahead of the caller (include/lpmp_engine.h, lpmp_set_speculation): what the
    # solver adapters switch on.  The bounds must be those of single calls.
    hist0 = []
    for _ in range(12):
        e.compute_pass(1); hist0.append(e.lower_bound())
    e.set_speculation(16)
    for _ in range(40):                              # builds the ticket lists of the batch sizes
        e.compute_pass(1); e.lower_bound()
    t_spec = timed(lambda: (e.compute_pass(1), e.lower_bound()), 64)
    st = e.speculation_stats()
    print("%s: pass %.3f ms, pass + LowerBound %.3f ms, LowerBound alone (nothing stale) %.3f ms; pass + LowerBound with passes running ahead %.3f ms (%s)"
          % (name, t_pass, t_both, t_lb, t_spec, st))
    e.close(); del const, dual
