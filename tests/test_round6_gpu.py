"""Round 6, on the GPU box: two PROCESSES hand DEVICE tensors to the collective backend — the code path of an RCCL run (device
all-to-all-v; `--overlap-exchange`'s asynchronous exchange_begin / exchange_end) with real data between two ranks.  RCCL refuses two
ranks on one device, so on the 1-GPU box the backend is gloo's own CUDA collectives (LPMP_DIST_DEVICE_COLLECTIVES=1: multi_gpu.DistComm
stops staging through the host); everything above the backend — buffers, split sizes, stream ordering between the engine's kernels,
pack / unpack and the collective — is what runs over RCCL."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu

PROBE = r"""
import os, sys, torch, torch.distributed as dist
dist.init_process_group("gloo")
r = dist.get_rank()
torch.cuda.set_device(0)
send = torch.arange(6, dtype=torch.float64, device="cuda") + 10 * r
out = torch.empty(5 if r == 0 else 7, dtype=torch.float64, device="cuda")
# rank 0 keeps 2 and ships 4; rank 1 ships 3 and keeps 3
w = dist.all_to_all_single(out, send, output_split_sizes=[2, 3] if r == 0 else [4, 3], input_split_sizes=[2, 4] if r == 0 else [3, 3], async_op=True)
w.wait()
torch.cuda.synchronize()
want = [0, 1, 10, 11, 12] if r == 0 else [2, 3, 4, 5, 13, 14, 15]
assert out.cpu().tolist() == [float(x) for x in want], out
t = torch.ones(1, dtype=torch.float64, device="cuda"); dist.all_reduce(t); assert t.item() == 2
dist.destroy_process_group()
print("DEVICE_COLLECTIVES_OK")
"""


def _backend_takes_device_tensors(tmp_path) -> bool:
    f = tmp_path / "probe.py"
    f.write_text(PROBE)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29611", str(f)], text=True, capture_output=True, timeout=600, env=env, cwd=ROOT)
    return p.returncode == 0 and p.stdout.count("DEVICE_COLLECTIVES_OK") == 2


def _bench(args, env=None, timeout=1500):
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LPMP_DIST_BACKEND")}
    e.update(env or {})
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "bench.py")] + args, text=True, cwd=ROOT, timeout=timeout, env=e)
    assert out.strip().splitlines()[-1].startswith('{"metric"'), out[-800:]
    return json.loads(out.strip().splitlines()[-1])


def test_two_processes_exchange_device_tensors_plain_and_overlapped(tmp_path):
    if not _backend_takes_device_tensors(tmp_path):
        pytest.skip("this torch's gloo has no device all_to_all_single: nothing but RCCL (one rank per device) can move device tensors here")
    env = {"LPMP_DIST_DEVICE_COLLECTIVES": "1"}
    # C5 miniature in lock step, 2 ranks: the state of BOTH ranks bit-identical to the oracle fixture, plain program ...
    c5 = ["--gpus", "2", "--workload", "c5", "--c5-small", "--steps", "4", "--warmup", "2", "--no-cpu-baseline"]
    d = _bench(c5, env)
    assert d["n_gpus"] == 2 and d["backend"] == "gloo" and d["schedule"] == "lockstep" and d["overlap_exchange"] is False
    assert d["oracle_check"]["duals_bit_identical_to_oracle"] is True and abs(d["dual_bound_gap"]) <= 1e-12
    assert d["exchange_bytes_per_pass"]["max"] > 0 and d["exchange_post_ms_per_pass"]["max"] == 0
    plain_lb = d["lower_bound_after"]
    # ... and with the collective posted asynchronously behind the cut-adjacent records, awaited before its first reader
    d = _bench(c5 + ["--overlap-exchange"], env)
    assert d["overlap_exchange"] is True and d["oracle_check"]["duals_bit_identical_to_oracle"] is True and abs(d["dual_bound_gap"]) <= 1e-12
    assert d["lower_bound_after"] == plain_lb
    # the posts (pack, copy, posting the collective) are booked as exchange time, not as compute: t_run means the same in both programs
    assert d["exchange_post_ms_per_pass"]["max"] > 0
    for r in range(2):
        pr = d["rank_stats"]["per_rank"]
        assert pr["exchange_ms_per_pass"][r] >= pr["exchange_post_ms_per_pass"][r] > 0
        assert abs(pr["total_ms_per_pass"][r] - pr["compute_ms_per_pass"][r] - pr["exchange_ms_per_pass"][r]) < 1e-6
    # the random graph in miniature (60 % of the edges cut: every exchange carries data in both directions), overlapped
    d = _bench(["--gpus", "2", "--workload", "c4", "--c4-nodes", "20000", "--c4-edges", "100000", "--steps", "3", "--warmup", "1", "--no-cpu-baseline",
                "--overlap-exchange"], env)
    assert abs(d["dual_bound_gap"]) <= 1e-12 and d["exchange_bytes_per_pass"]["max"] > 1e6 and "partitioner" in d["dual_bound_gap_detail"]["gap_config"]
    # the headline grid's overlap schedule (ghost rows refreshed by a device all-to-all every n passes)
    d = _bench(["--gpus", "2", "--grid", "128", "--steps", "6", "--warmup", "2", "--no-cpu-baseline"], env)
    assert d["schedule"] == "overlap" and abs(d["dual_bound_gap"]) <= 1e-12 and d["exchange_bytes_per_pass"]["max"] > 0
