"""C3: cost of a pass with primal rounding against a plain pass, and of EvaluatePrimal."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lp_mp_amd import engine as E, synthetic as S
H = W = 1024; L = 32
sp = torch.cuda.current_stream().cuda_stream
n = H * W; n_e = len(S.grid_edges(H, W)[0])
m = S.grid_model(H, W, L, order="colour_major", seed=1, device_const=True, compute_primal=True)
const = torch.empty(n_e * L * L, dtype=torch.float64, device="cuda:0")
E.synth_fill(const.data_ptr(), const.numel(), 1, n * L, sp)
torch.cuda.synchronize()
e = E.Engine(0); e.set_stream(sp)
e.upload(m, const_dev=const.data_ptr(), keep=(const,))
e.set_reparametrization(0)
e.compute_pass(3); e.compute_pass_and_primal(0); torch.cuda.synchronize()
def timed(f, n=10):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n): f(i)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("plain pass, one per call        %.3f ms" % timed(lambda i: e.compute_pass(1)))
print("forward + backward, unfused     %.3f ms" % timed(lambda i: (e.forward_pass(), e.backward_pass())))
print("pass with primal rounding       %.3f ms" % timed(lambda i: e.compute_pass_and_primal(i + 1)))
print("EvaluatePrimal                  %.3f ms" % timed(lambda i: e.evaluate_primal()))
print("LB %.4f  primal %.4f" % (e.lower_bound(), e.evaluate_primal()))
for it in range(11, 60):
    e.compute_pass(1)
    if it % 5 == 0:
        e.compute_pass_and_primal(it); print(it, "LB %.3f primal %.3f" % (e.lower_bound(), e.evaluate_primal()))
