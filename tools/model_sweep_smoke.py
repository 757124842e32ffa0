"""torchrun smoke of ModelSweep (general partitioned model on real engines): C5 at full size, colour-major edge
variables, split by graph_partition_model.  On a 1-GPU box run with LPMP_DIST_BACKEND=gloo (ranks share the device)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.distributed as dist
from lp_mp_amd import model as M, multi_gpu as MG, synthetic as S
world = int(os.environ.get("WORLD_SIZE", "1")); rank = int(os.environ.get("RANK", "0"))
torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count())
dist.init_process_group(os.environ.get("LPMP_DIST_BACKEND", "nccl"), rank=rank, world_size=world)
gm = S.c5_model(512, 512, 8, 150000, 70000, 30000, seed=4, window=64, colour_edge_vars=True)
part = MG.graph_partition_model(gm, world)
sw = MG.ModelSweep(torch, dist, gm, part, M.REPAM_ANISOTROPIC, None, "pass")
lbs = [sw.lower_bound()]
sw.compute_pass(2); torch.cuda.synchronize(); dist.barrier()
t0 = time.perf_counter(); sw.compute_pass(10); torch.cuda.synchronize(); dist.barrier(); dt = (time.perf_counter() - t0) / 10
lbs.append(sw.lower_bound())
if rank == 0:
    print("world %d: %.3f ms per pass, ghosts on rank 0: %d, LB %.3f -> %.3f" % (world, dt * 1e3, sw.part.n_ghost, lbs[0], lbs[1]))
dist.destroy_process_group()
