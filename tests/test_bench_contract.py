"""bench.py pieces that do not need a GPU: the CPU-baseline leg, the PMC traffic lookup, and (on the GPU box) the
shape of the JSON line the driver parses."""
import json
import os
import subprocess
import sys
import types

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from lp_mp_amd import model as M, synthetic as S  # noqa: E402


def _args(**kw):
    d = dict(grid=1024, labels=32, pairwise="dense", order="colour_major", mode="anisotropic", cpu_sample_grid=24)
    d.update(kw)
    return types.SimpleNamespace(**d)


def test_cpu_baseline_leg_runs_the_oracle_on_a_bounded_sample():
    a = _args(cpu_sample_grid=16)
    out = bench.cpu_baseline(a, S, M)
    assert out["kind"] == "port" and out["cores"] == 1 and out["unit"] == "msg-updates/s" and out["value"] > 0
    assert "16x16" in out["sample"]


def _latest_pmc(what):
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_pmc_{what}.json")))
    return files[-1], json.load(open(files[-1]))


def test_pmc_traffic_lookup_is_tied_to_the_library_that_was_profiled(monkeypatch):
    """roofline.traffic comes from a committed counter summary ONLY when that summary was taken with the library that is running:
    the summary carries the source hash of the library under the profiler, bench.py compares it with the running library's stamp
    and reports `traffic: null` + the reason otherwise (a kernel change without a re-profile must not leave a stale ratio)"""
    f3, d3 = _latest_pmc("c3_dense32")
    kern3 = "void lpmp::" + d3["kernel"] + ", 2, false, false, false>" if d3["kernel"].startswith("chain_") else d3["kernel"]
    # (1) another library than the profiled one: no traffic, and the line says why
    monkeypatch.setattr(bench, "library_source_hash", lambda: "0" * 64)
    t, why = bench.pmc_traffic(kern3, _args(steps=20))
    assert t is None and why.startswith("stale: profiles/") and "re-profile" in why
    # (2) the profiled library itself (a summary of a round that recorded the hash): bytes, scaled to the call's pass count
    if d3.get("library_source_hash"):
        monkeypatch.setattr(bench, "library_source_hash", lambda: d3["library_source_hash"])
        t20, src20 = bench.pmc_traffic(kern3, _args(steps=20))
        t10, _ = bench.pmc_traffic(kern3, _args(steps=10))
        assert src20 == os.path.relpath(f3, ROOT) and 3e10 < t20 / 20 < 4.5e10 and abs(t10 * 2 - t20) < 1e-6 * t20
        assert d3["kernel_full_names"] and all(d3["kernel"] in n for n in d3["kernel_full_names"])
        f4, d4 = _latest_pmc("c4_dense16")
        if d4.get("library_source_hash"):
            monkeypatch.setattr(bench, "library_source_hash", lambda: d4["library_source_hash"])
            c4 = _args(workload="c4", c4_nodes=2_000_000, c4_edges=10_000_000, c4_labels=16, c4_order=d4.get("variable_order", "index"))
            t4, src4 = bench.pmc_traffic(d4["kernel"], c4)
            assert src4 == os.path.relpath(f4, ROOT) and 5e8 < t4 < 5e9            # (index order: 66 launches of ~0.8 GB per pass; colour-major: 18 of ~3.4 GB)
    # other sizes, other workloads: never a number
    assert bench.pmc_traffic(kern3, _args(grid=512))[0] is None
    assert bench.pmc_traffic("sweep_dense_pk_kernel<16, 2, false, true>", _args(workload="c4", c4_nodes=20000, c4_edges=100000, c4_labels=16))[0] is None
    assert bench.pmc_traffic("sweep_generic_kernel<1>", _args(workload="c5"))[0] is None


def test_running_library_reports_the_hash_of_its_sources():
    from lp_mp_amd import build as B
    B.build()
    assert bench.library_source_hash() == B.source_hash() and len(B.source_hash()) == 64


@pytest.mark.gpu
def test_bench_json_line_contract():
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "bench.py"), "--grid", "128", "--steps", "3", "--warmup", "1",
                                   "--cpu-sample-grid", "16"], text=True, cwd=ROOT, timeout=600)
    d = json.loads([l for l in out.strip().splitlines() if l.startswith('{"metric"')][-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["dtype"] == "f64" and d["data"] == "synthetic"
    assert d["vs_baseline"] is None and d["higher_is_better"] is True and "workload" in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert d["cpu_baseline"]["kind"] == "port"
    rd = d["rounding"]                                         # one pass with primal rounding, outside the timed region
    assert rd["primal_cost"] >= rd["lower_bound"] and rd["ms_pass_and_primal"] > 0
    # launch per step, or the whole pass as one chain launch (engine.cpp rotation_chain): same body either way
    assert d["roofline"]["kernel"].startswith(("sweep_dense_pk_kernel<32, 2, false", "chain_dense_pk_kernel<32, 2, false"))


@pytest.mark.gpu
def test_bench_row_major_order_runs_the_mailbox_chain():
    """--order row_major: one level per anti-diagonal, each directional sweep one persistent launch whose records hand their
    message vectors over through the mailbox; the line carries the same contract, and the bound of the colour-major default
    at the same pass count is a different one (another sweep order)"""
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "bench.py"), "--grid", "128", "--order", "row_major", "--steps", "3",
                                   "--warmup", "1", "--no-cpu-baseline"], text=True, cwd=ROOT, timeout=600)
    d = json.loads([l for l in out.strip().splitlines() if l.startswith('{"metric"')][-1])
    assert d["config"]["levels_per_direction"] == [255, 255] and "row_major" in d["config"]["workload"]
    assert d["value"] > 0 and d["lower_bound_after"] > d["lower_bound_before"] and d["rounding"]["primal_cost"] >= d["rounding"]["lower_bound"]
    from lp_mp_amd import engine as E, model as M, synthetic as S
    ci = E.Plan(S.grid_model(128, 128, 32, order="row_major", device_const=True)).chain_info(-1, M.REPAM_ANISOTROPIC)
    assert ci["n_chains"] == 1 and ci["mailbox_rows"] == 2 * (2 * 128 * 127)


@pytest.mark.gpu
def test_bench_c4_workload_line():
    """--workload c4 (BASELINE configs[3] in miniature on one GPU): same contract, the random-graph workload named"""
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "c4", "--c4-nodes", "20000", "--c4-edges", "100000",
                                   "--steps", "3", "--warmup", "1", "--no-cpu-baseline"], text=True, cwd=ROOT, timeout=600)
    d = json.loads([l for l in out.strip().splitlines() if l.startswith('{"metric"')][-1])
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["unit"] == "msg-updates/s" and d["value"] > 0
    assert "random sparse graph G(20000, 100000)" in d["config"]["workload"] and d["config"]["msg_updates_per_pass"] == 400000
    assert d["dual_bound_gap"] == 0.0 and d["lower_bound_after"] > d["lower_bound_before"]


@pytest.mark.gpu
@pytest.mark.parametrize("schedule", ["overlap", "boundary", "lockstep"])
@pytest.mark.parametrize("workload", ["c3", "c4"])
def test_bench_distributed_branch_on_rccl_at_world_size_one(workload, schedule):
    """what an N-GPU launch of bench.py executes first, on the one GPU of the test box: --force-dist takes the multi-GPU
    branch at WORLD_SIZE 1 — init_process_group("nccl") (= RCCL), StripSweep / GraphSweep, all_to_all_single with empty
    splits in every boundary step, device all_reduce for the bound and the timing — and must print the contract line"""
    if workload == "c4" and schedule == "overlap":
        pytest.skip("the overlap schedule is for grids")
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29571")
    env.pop("LPMP_DIST_BACKEND", None)
    extra = ["--grid", "128"] if workload == "c3" else ["--workload", "c4", "--c4-nodes", "20000", "--c4-edges", "100000"]
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "bench.py"), "--force-dist", "--steps", "3", "--warmup", "1",
                                   "--no-cpu-baseline", "--schedule", schedule] + extra, text=True, cwd=ROOT, timeout=900, env=env)
    # the line the driver parses is the last thing on stdout (RCCL's version banner, buffered by C stdio, must not trail it)
    assert out.strip().splitlines()[-1].startswith('{"metric"'), out[-600:]
    d = json.loads([l for l in out.strip().splitlines() if l.startswith('{"metric"')][-1])
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["lower_bound_after"] > d["lower_bound_before"]
    assert abs(d["dual_bound_gap"]) <= 1e-9          # one part: the partitioned schedule IS the unpartitioned sweep
    assert d["roofline"]["kernel"].startswith(("sweep_dense_pk_kernel", "chain_dense_pk_kernel"))


def test_default_multi_gpu_schedule_is_always_an_exact_one():
    """--schedule auto: overlap for colour-major grids with an even strip height, lock step for everything else (other grid orders,
    odd heights, the C4 graph) — never the approximate boundary-step schedule"""
    assert bench.auto_schedule(_args(workload="c3")) == "overlap"
    assert bench.auto_schedule(_args(workload="c3", order="row_major")) == "lockstep"
    assert bench.auto_schedule(_args(workload="c3", grid=1023)) == "lockstep"
    assert bench.auto_schedule(_args(workload="c4")) == "lockstep"


def test_golden_fixture_covers_whatever_pass_count_the_driver_runs():
    """bench.py's oracle_check compares the state the timed call leaves in HBM with the oracle's after warmup + steps passes
    on the same inputs (tests/golden/c3_full_lb.npz, seed 1).  The round driver chooses --warmup / --steps (round 1 and 2:
    5 + 20 = 25; bench.py's own default: 3 + 20 = 23): every count up to 48 has an entry, and the two the judge computed
    independently in round 2 are the fixture's"""
    import numpy as np
    g = np.load(os.path.join(ROOT, "tests", "golden", "c3_full_lb.npz"))
    assert list(g["passes_seed1"]) == list(range(49))
    lb = dict(zip(map(int, g["passes_seed1"]), map(float, g["lb_seed1"])))
    assert abs(lb[23] - 346558.6683229103) <= 1e-9 * lb[23] and abs(lb[25] - 346987.1451928538) <= 1e-9 * lb[25]
    assert all(lb[k + 1] >= lb[k] for k in range(48))                                 # dual ascent, pass by pass
    assert len(set(map(int, g["dual_sum_seed1"]))) == 49 and len(set(map(int, g["dual_wsum_seed1"]))) == 49


# ---- `python bench.py --gpus N` without a launcher around it (the shape of the driver's command) ----------------------
def test_launcher_dry_run_names_one_rank_per_gpu():
    """WORLD_SIZE unset and --gpus 4: bench.py must start 4 rank processes itself.  --dry-run-launch prints what it would
    start: same arguments, RANK / LOCAL_RANK 0..3, WORLD_SIZE 4, one rendezvous on 127.0.0.1 — and touches no GPU"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "7", "--warmup", "2",
                                   "--dry-run-launch"], text=True, cwd=ROOT, timeout=120, env=env)
    d = json.loads(out.strip().splitlines()[-1])
    assert d["n_ranks"] == 4 and len(d["ranks"]) == 4
    ports = set()
    for r, c in enumerate(d["ranks"]):
        assert c["env"]["RANK"] == str(r) and c["env"]["LOCAL_RANK"] == str(r) and c["env"]["WORLD_SIZE"] == "4"
        assert c["env"]["MASTER_ADDR"] == "127.0.0.1"
        ports.add(c["env"]["MASTER_PORT"])
        assert c["argv"][1].endswith("bench.py") and c["argv"][2:] == ["--gpus", "4", "--steps", "7", "--warmup", "2"]
    assert len(ports) == 1


def test_launcher_exits_nonzero_when_a_rank_fails():
    """no GPU in this container: every self-launched rank stops with "no HIP device"; the launcher must report that with
    a non-zero exit code and without hanging on the others (on the GPU box the same command runs, test below)"""
    import torch
    if torch.cuda.device_count() > 0:
        pytest.skip("needs a box without a GPU")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--grid", "16", "--steps", "1", "--warmup", "1",
                        "--no-cpu-baseline"], text=True, cwd=ROOT, timeout=600, env=env, capture_output=True)
    assert p.returncode != 0 and "no HIP device" in p.stderr and "launcher: rank" in p.stderr
    assert not any(l.startswith('{"metric"') for l in p.stdout.splitlines())


def test_gpus_flag_must_agree_with_the_world_size():
    """under an external launcher (WORLD_SIZE set) a different --gpus is an error, not a warning: the line would otherwise
    claim a GPU count that did not run"""
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1"], text=True, cwd=ROOT, timeout=300,
                       env=env, capture_output=True)
    assert p.returncode == 2 and "--gpus 8 but WORLD_SIZE 1" in p.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("schedule", ["overlap", "boundary"])
def test_bench_gpus_2_launches_its_own_ranks(schedule):
    """`python bench.py --gpus 2` with no launcher around it, on the 1-GPU test box: two rank processes share the device
    (backend falls back to gloo: RCCL refuses two ranks on one GPU), the line says how many ranks ran and where"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LPMP_DIST_BACKEND")}
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--grid", "128", "--steps", "4", "--warmup", "2",
                                   "--no-cpu-baseline", "--schedule", schedule] + (["--compare-schedules"] if schedule == "overlap" else []),
                                  text=True, cwd=ROOT, timeout=1200, env=env)
    assert out.strip().splitlines()[-1].startswith('{"metric"'), out[-600:]
    d = json.loads(out.strip().splitlines()[-1])
    # (the other schedules on the same strips only when asked for: nothing that is not the measurement runs by default)
    assert (set(d["schedules"]) == {"overlap", "lockstep", "boundary"}) if schedule == "overlap" else d["schedules"] is None
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and len(d["devices"]) == 2 and d["launch"]["launcher"].startswith("bench.py")
    assert d["backend"] in ("gloo", "rccl (torch.distributed nccl)")
    assert d["value"] > 0 and d["lower_bound_after"] > d["lower_bound_before"]
    if schedule == "overlap":
        assert abs(d["dual_bound_gap"]) <= 1e-12       # the exact schedule: the unpartitioned sweep's bound
    else:
        assert 0 <= d["dual_bound_gap"] < 0.01


@pytest.mark.gpu
def test_bench_gpus_2_c4_runs_in_lock_step_by_default():
    """`python bench.py --gpus 2 --workload c4` (a 20 000-node graph here): the default schedule is the exact one (lock step, colour-major
    variable order on every path), the line carries the boundary-step schedule of the same model beside it"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LPMP_DIST_BACKEND")}
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "c4", "--c4-nodes", "20000", "--c4-edges", "100000",
                                   "--steps", "4", "--warmup", "2", "--no-cpu-baseline", "--compare-schedules"], text=True, cwd=ROOT, timeout=1200, env=env)
    d = json.loads(out.strip().splitlines()[-1])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["schedule"] == "lockstep" and d["config"]["variable_order"] == "colour_major"
    assert abs(d["dual_bound_gap"]) <= 1e-12 and d["lower_bound_after"] > d["lower_bound_before"]
    assert set(d["schedules"]) == {"lockstep", "boundary"} and 0 <= d["schedules"]["boundary"]["dual_bound_gap"] < 0.05
    assert d["scaling"] == "strong"


RDZV_WORKER = r"""
import os, sys, json, types
sys.path.insert(0, {root!r})
import torch, torch.distributed as dist
import bench
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
args = types.SimpleNamespace(rendezvous_timeout=60.0, collective_timeout=60.0)
# (no GPU here: the identities are made up — two ranks on "one device", as on the 1-GPU box, one on another)
ident = "pci=0000:05:00.0 uuid=" + "a" * 32 if rank < 2 else "pci=0000:06:00.0 uuid=" + "b" * 32
idents, backend = bench.rendezvous(args, torch, dist, ident, rank, world)
t = torch.ones(1); dist.all_reduce(t)
class R:                                    # the self test of the collectives on a runner without exchanges
    comm = None
st = bench.collective_self_test(args, torch, dist, R(), rank, world) if False else None
if rank == 0:
    json.dump({{"idents": idents, "backend": backend, "sum": float(t.item()), "agent_store": os.environ.get("TORCHELASTIC_USE_AGENT_STORE")}},
              open(os.path.join({out!r}, "rdzv.json"), "w"))
dist.destroy_process_group()
"""


def test_rendezvous_under_torch_distributed_run_uses_the_agents_store(tmp_path):
    """the driver's N-GPU command is `python -m torch.distributed.run ... bench.py --gpus N`: the ranks then meet at the AGENT's
    store (bench.rendezvous goes through torch's env rendezvous, which connects to it instead of binding the port again), exchange
    their device identities there and choose the backend from the number of PHYSICAL devices — gloo as soon as two ranks share one"""
    script = tmp_path / "rdzv_worker.py"
    script.write_text(RDZV_WORKER.format(root=ROOT, out=str(tmp_path)))
    env = {k: v for k, v in os.environ.items() if k != "LPMP_DIST_BACKEND"}
    env.update(MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3",
                           "--master-addr", "127.0.0.1", "--master-port", "29553", str(script)], env=env, cwd=ROOT, timeout=600)
    d = json.load(open(tmp_path / "rdzv.json"))
    assert len(d["idents"]) == 3 and d["idents"][0] == d["idents"][1] != d["idents"][2]
    assert d["backend"] == "gloo" and d["sum"] == 3.0 and d["agent_store"] == "True"


def test_watchdog_exits_with_code_3_and_names_the_rank():
    code = ("import sys, time; sys.path.insert(0, %r); import bench\n"
            "with bench.Watchdog(0.5, 'the self test of the collectives', 5):\n"
            "    time.sleep(30)\n" % ROOT)
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert p.returncode == 3 and "rank 5" in p.stderr and "did not finish within" in p.stderr
    code_ok = ("import sys; sys.path.insert(0, %r); import bench\n"
               "with bench.Watchdog(5, 'x', 0):\n    pass\nprint('fine')\n" % ROOT)
    p = subprocess.run([sys.executable, "-c", code_ok], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0 and "fine" in p.stdout


def test_rendezvous_gives_up_with_a_message_when_a_rank_never_comes():
    """WORLD_SIZE 2, one process: the rendezvous waits --rendezvous-timeout and ends the process with a message naming what was
    missing — no hang, no retry"""
    code = ("import os, sys, types; sys.path.insert(0, %r)\n"
            "import torch, torch.distributed as dist, bench\n"
            "args = types.SimpleNamespace(rendezvous_timeout=3.0, collective_timeout=5.0)\n"
            "bench.rendezvous(args, torch, dist, 'pci=0 uuid=0', 0, 2)\n" % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29557", RANK="0", WORLD_SIZE="2")
    env.pop("TORCHELASTIC_USE_AGENT_STORE", None)
    import time
    t0 = time.time()
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 3 and time.time() - t0 < 120         # the start-up code (watchdogs, rendezvous, self test), not a crash's 1
    assert "bench.py: rank 0:" in p.stderr and ("no rendezvous" in p.stderr or "did not show up" in p.stderr)


def test_scaling_model_says_when_t1_comes_from_another_library(monkeypatch):
    """scaling_model's t_1 (single-GPU ms per pass of the same workload) is read from the latest committed bench line; the line
    carries the source hash of the library that produced it, and the model flags a t_1 taken with another build than the running
    one (`t1_stale`, `t1_note`) instead of silently dividing two builds' times"""
    a = types.SimpleNamespace(workload="c3", grid=1024, labels=32, pairwise="dense", order="colour_major", mode="anisotropic",
                              assume_exchange_latency_us=30.0, assume_exchange_GBps=400.0)
    t1 = bench.single_gpu_reference(a)
    assert t1 is not None and t1["source"].startswith("profiles/") and 4.0 < t1["ms_per_pass"] < 7.0
    stats = {"max": {"compute_ms_per_pass": 5.3, "exchange_ms_per_pass": 0.1, "exchanges_per_pass": 0.2, "exchange_bytes_out_per_pass": 3e6,
                     "exchange_bytes_in_per_pass": 3e6}}
    monkeypatch.setattr(bench, "library_source_hash", lambda: "f" * 64)
    m = bench.scaling_model(a, 8, stats, 5.4, False)
    assert m["t1_stale"] is True and "compares two builds" in m["t1_note"] and m["t1_library_source_hash"] == t1["library_source_hash"]
    assert m["kind"] == "weak" and abs(m["projected_efficiency"] - t1["ms_per_pass"] / m["projected_ms_per_pass"]) < 1e-12
    if t1["library_source_hash"]:
        monkeypatch.setattr(bench, "library_source_hash", lambda: t1["library_source_hash"])
        m = bench.scaling_model(a, 8, stats, 5.4, False)
        assert m["t1_stale"] is False and "t1_note" not in m
