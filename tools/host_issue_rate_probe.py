"""host time per call of what a lock-step exchange issues through torch.distributed (backend nccl = RCCL, world 1 on the test box):
all_to_all_single of a 13 MB buffer, beside the engine's C calls.   python tools/host_issue_rate_probe.py"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
import numpy as np, torch, torch.distributed as dist
from lp_mp_amd import engine as E, model as M, multi_gpu as MG, synthetic as S
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
dev = torch.device("cuda:0")
comm = MG.DistComm(dist, torch); comm._dev = dev
n = 13_000_000 // 8
send = torch.zeros(n, dtype=torch.float64, device=dev)
counts = np.array([n], np.int64)
for _ in range(20):
    comm.exchange(send, counts, counts)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(500):
    comm.exchange(send, counts, counts)
t_issue = (time.perf_counter() - t0) / 500
torch.cuda.synchronize()
t_total = (time.perf_counter() - t0) / 500
# the engine's C calls of one exchange / one run on a small model (host time only)
g = S.grid_model(32, 32, 16, seed=1)
e = E.Engine(0); e.set_stream(torch.cuda.current_stream().cuda_stream); e.upload(g); e.set_reparametrization(M.REPAM_ANISOTROPIC)
off = g.dual_offsets()[1024:1024 + 500]
h = e.halo_create(off, np.full(500, 16), off + 16, np.full(500, 16))
buf = torch.zeros(8000, dtype=torch.float64, device=dev)
e.compute_pass(1); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(500):
    e.halo_pack(h, buf.data_ptr()); e.halo_unpack(h, buf.data_ptr())
t_halo = (time.perf_counter() - t0) / 500
torch.cuda.synchronize()
p = e.plan
upd = p.update_order(0); om_off, om = p.omega(0, 0); mk_off, mk = p.mask(0, 0)
sid = e.schedule_create(upd, om_off, om, mk_off, mk)
e.schedule_run(sid); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(500):
    e.schedule_run(sid)
t_run = (time.perf_counter() - t0) / 500
torch.cuda.synchronize()
print(json.dumps({"all_to_all_single_13MB_host_us": round(t_issue * 1e6, 1), "all_to_all_single_13MB_with_device_us": round(t_total * 1e6, 1),
                  "halo_pack_plus_unpack_host_us": round(t_halo * 1e6, 1), "schedule_run_host_us": round(t_run * 1e6, 1)}))
e.halo_destroy(h); e.close()
dist.destroy_process_group()
