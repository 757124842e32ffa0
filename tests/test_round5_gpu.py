"""Round 5 on the device: the rows-layout hand-over through the halo / boundary kernels, lock-step parts on the rows
layout, the per-engine switch of persistent launches, device identities, validation of halo vectors, the lazily built chain plan
of a joined pass, and the N-rank keys / the C5 workload of bench.py."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lp_mp_amd import engine as E, lockstep as LS, model as M, multi_gpu as MG, synthetic as S  # noqa: E402
from oracle.binding import Oracle  # noqa: E402

pytestmark = pytest.mark.gpu


# ---- rows layout: what the halo kernels read and write is never lost -----------------------------------------------------------
def _halo_over_all_pairwise_sides(eng, g, nv, L):
    off = g.dual_offsets()
    pw = np.arange(nv, g.n_factors)
    out_off = np.concatenate([off[pw], off[pw] + L]); ln = np.full(out_off.shape[0], L)
    return eng.halo_create(out_off, ln, out_off, ln), out_off


def test_rows_layout_halo_pack_sees_uploaded_duals():
    """(a) lpmp_upload_duals leaves the rows stale; a halo pack that follows must ship the uploaded vectors, not the rows' old ones"""
    L, g = 16, S.grid_model(6, 7, 16, seed=3)
    eng = E.Engine(0); eng.set_stream(torch.cuda.current_stream().cuda_stream)
    try:
        eng.upload(g, rows_layout=True); eng.set_reparametrization(M.REPAM_ANISOTROPIC)
        assert eng.rows_layout
        eng.compute_pass(2)
        h, out_off = _halo_over_all_pairwise_sides(eng, g, 42, L)
        new = np.random.default_rng(1).uniform(-1, 1, eng.download_duals().shape[0])
        eng.upload_duals(new)
        send = torch.zeros(out_off.shape[0] * L, dtype=torch.float64, device="cuda:0")
        eng.halo_pack(h, send.data_ptr()); torch.cuda.synchronize()
        assert np.array_equal(send.cpu().numpy(), np.concatenate([new[o:o + L] for o in out_off]))
        eng.halo_destroy(h)
    finally:
        eng.close()


def test_rows_layout_halo_unpack_after_a_hand_over_survives_the_next_pass():
    """(b) lpmp_synchronize on a BORROWED dual buffer marks the rows stale (the caller may write the buffer); a halo unpack that
    follows writes the rows — the next pass must compute on what was unpacked, as the oracle does"""
    L, g = 16, S.grid_model(6, 7, 16, seed=4)
    dual = torch.from_numpy(g.dual_data.copy()).to("cuda:0")
    eng = E.Engine(0); eng.set_stream(torch.cuda.current_stream().cuda_stream)
    o = Oracle(g); o.set_reparametrization(M.REPAM_ANISOTROPIC)
    try:
        eng.upload(g, dual_dev=dual.data_ptr(), keep=dual, rows_layout=True); eng.set_reparametrization(M.REPAM_ANISOTROPIC)
        eng.compute_pass(1); o.ComputePass(1)
        eng.synchronize()                                             # hand-over: rows -> packed, rows stale
        assert np.array_equal(dual.cpu().numpy(), o.duals())
        h, out_off = _halo_over_all_pairwise_sides(eng, g, 42, L)
        vals = np.random.default_rng(2).uniform(-1, 1, out_off.shape[0] * L)
        recv = torch.from_numpy(vals).to("cuda:0")
        eng.halo_unpack(h, recv.data_ptr())
        d = o.duals().copy()
        for k, off in enumerate(out_off):
            d[off:off + L] = vals[k * L:(k + 1) * L]
        o.set_duals(d)
        eng.compute_pass(2); o.ComputePass(2)
        assert np.array_equal(eng.download_duals(), o.duals())
        assert abs(eng.lower_bound() - o.LowerBound()) <= 1e-9 * max(1.0, abs(o.LowerBound()))
        eng.halo_destroy(h)
    finally:
        eng.close()


def test_rows_layout_direct_access_contract():
    """(c) a caller that reads / writes pairwise vectors of a borrowed buffer directly: lpmp_synchronize first, then the access,
    then lpmp_invalidate_lower_bounds — what it wrote is what the next pass computes on"""
    L, g = 8, S.counter_graph_model(400, 1500, 8, 2)
    dual = torch.from_numpy(g.dual_data.copy()).to("cuda:0")
    eng = E.Engine(0); eng.set_stream(torch.cuda.current_stream().cuda_stream)
    o = Oracle(g); o.set_reparametrization(M.REPAM_ANISOTROPIC)
    try:
        eng.upload(g, dual_dev=dual.data_ptr(), keep=dual, rows_layout=True); eng.set_reparametrization(M.REPAM_ANISOTROPIC)
        for _ in range(2):
            eng.compute_pass(1); o.ComputePass(1)
            eng.synchronize()
            d = dual.cpu().numpy()
            assert np.array_equal(d, o.duals())
            d = d.copy(); d[400 * L:] += 0.125                          # every pairwise vector, directly in the borrowed buffer
            dual.copy_(torch.from_numpy(d)); torch.cuda.synchronize()
            eng.invalidate_lower_bounds()
            o.set_duals(d)
            assert abs(eng.lower_bound() - o.LowerBound()) <= 1e-9 * max(1.0, abs(o.LowerBound()))
        eng.compute_pass(2); o.ComputePass(2)
        assert np.array_equal(eng.download_duals(), o.duals())
    finally:
        eng.close()


@pytest.mark.parametrize("rows", [True, False])
def test_lock_step_parts_on_the_rows_layout_equal_the_oracle(rows, monkeypatch):
    """several lock-step parts whose engines keep their dense pairwise factors as rows: the halo kernels address the rows, the
    result is the unpartitioned oracle's bit for bit.  With rows=False the environment asks for the rows layout
    (LPMP_ROWS_LAYOUT=1, the README's switch) and the drivers' explicit `rows_layout=False` must win"""
    from tests.test_lockstep import _graph, _global_of, _parts_of, _assert_equals_global
    if not rows:
        monkeypatch.setenv("LPMP_ROWS_LAYOUT", "1")
    c = _graph(1200, 5000, 16, 3, 4)
    ref = Oracle(_global_of(c)); ref.set_reparametrization(M.REPAM_ANISOTROPIC)
    sched, parts = _parts_of(c, M.REPAM_ANISOTROPIC)
    sweeps, tensors = [], []
    for p in parts:
        dual = torch.from_numpy(p.model.dual_data.copy()).to("cuda:0")
        eng = E.Engine(0); eng.set_stream(torch.cuda.current_stream().cuda_stream)
        eng.upload(p.model, dual_dev=dual.data_ptr(), keep=dual, rows_layout=rows)
        assert eng.rows_layout == rows
        eng.set_reparametrization(M.REPAM_ANISOTROPIC)
        sweeps.append(LS.LockstepSweep(torch, p, sched, eng, dual)); tensors.append(dual)
    try:
        for n in (1, 3):
            ref.ComputePass(n)
            LS.run_lockstep(sweeps, n)
            for s in sweeps:
                s.engine.synchronize()                                 # (rows layout: the packed buffer is written out here)
            _assert_equals_global(c, parts, [t.cpu().numpy() for t in tensors], ref)
            lb = sum(s.local_lower_bound() for s in sweeps)
            assert abs(lb - ref.LowerBound()) <= 1e-9 * max(1.0, abs(ref.LowerBound()))
    finally:
        for s in sweeps:
            s.close(); s.engine.close()


def test_overlap_driver_ignores_the_environments_rows_layout(monkeypatch):
    """OverlapStrips reads and writes the borrowed packed buffer between passes: it uploads with the packed layout whatever
    LPMP_ROWS_LAYOUT / LPMP_SPECULATION say, and its single-window run equals the plain engine"""
    from lp_mp_amd import overlap as OV
    monkeypatch.setenv("LPMP_ROWS_LAYOUT", "1"); monkeypatch.setenv("LPMP_SPECULATION", "8")
    r = OV.OverlapStrips(torch, None, 16, 16, 8, "dense", M.REPAM_ANISOTROPIC, seed=1, g=4)
    try:
        assert not r.engine.rows_layout
        r.compute_pass(3)
        o = Oracle(S.grid_model(16, 16, 8, order="colour_major", seed=1)); o.set_reparametrization(M.REPAM_ANISOTROPIC); o.ComputePass(3)
        assert abs(r.lower_bound() - o.LowerBound()) <= 1e-9 * abs(o.LowerBound())
    finally:
        r.engine.close()


# ---- C ABI additions -------------------------------------------------------------------------------------------------------------
def test_halo_and_boundary_vectors_are_validated():
    g = S.grid_model(5, 5, 4, seed=1)
    eng = E.Engine(0)
    try:
        eng.upload(g)
        n = eng.download_duals().shape[0]
        ok = eng.halo_create([0], [4], [4], [4]); eng.halo_destroy(ok)
        for bad_off, bad_len in (([n - 2], [4]), ([-1], [4]), ([2], [4]), ([0], [-1]), ([25 * 4 + 6], [4])):   # past the end, negative, across two factors (unaries; pairwise duals are 8 long)
            with pytest.raises(E.EngineError) as ei:
                eng.halo_create(bad_off, bad_len, [], [])
            assert ei.value.code == -1 and "halo" in str(ei.value)
            with pytest.raises(E.EngineError):
                eng.halo_create([], [], bad_off, bad_len)
        with pytest.raises(E.EngineError) as ei:
            eng.boundary_create([n], [4], [], [], [], [])
        assert ei.value.code == -1
    finally:
        eng.close()


def test_device_identity_names_the_physical_device():
    a, b = E.device_identity(0), E.device_identity(0)
    assert a == b and a.startswith("pci=") and " uuid=" in a and len(a.split("uuid=")[1]) == 32
    with pytest.raises(E.EngineError):
        E.device_identity(99)


def test_persistent_launches_can_be_switched_off_per_engine():
    """a deep schedule (row-major grid) runs as a chain launch by default; lpmp_set_persistent_launches(e, 0) makes the same
    engine run it launch by launch — same duals, and the kernel timing says which form ran"""
    m = S.grid_model(40, 36, 8, order="row_major", seed=3)
    res = {}
    for on in (True, False):
        e = E.Engine(0)
        try:
            e.upload(m); e.set_reparametrization(M.REPAM_ANISOTROPIC)
            assert e.persistent_launches
            e.set_persistent_launches(on)
            assert e.persistent_launches == on
            e.enable_kernel_timing(False)
            e.compute_pass(3)
            d = e.download_duals()
            e.reset_kernel_timing(); e.enable_kernel_timing(True)
            e.compute_pass(1); e.synchronize()
            kt = e.kernel_timing(); e.enable_kernel_timing(False)
            res[on] = (d, kt)
        finally:
            e.close()
    assert np.array_equal(res[True][0], res[False][0])
    o = Oracle(m); o.set_reparametrization(M.REPAM_ANISOTROPIC); o.ComputePass(3)
    assert np.array_equal(res[False][0], o.duals())


def test_single_pass_of_a_rotating_model_gets_its_chain_plan_lazily(monkeypatch):
    """a colour-major grid's passes join (one persistent launch for n passes), so its forward+backward schedule is planned without
    a chain plan of its own; a pass that then runs on its own (residual sends, one pass per call) builds it on first use —
    forced onto a small model with LPMP_BAND_MIN_BYTES; bit for bit against the oracle, before and after, joined passes included"""
    monkeypatch.setenv("LPMP_BAND_MIN_BYTES", "1000"); monkeypatch.setenv("LPMP_BAND_BYTES", "20000")
    m = S.grid_model(24, 20, 8, order="colour_major", seed=6)
    o = Oracle(m); o.set_reparametrization(M.REPAM_ANISOTROPIC)
    e = E.Engine(0)
    try:
        e.upload(m); e.set_reparametrization(M.REPAM_ANISOTROPIC)
        assert e.plan.pass_rotates(M.REPAM_ANISOTROPIC)
        e.compute_pass(3); o.ComputePass(3)
        assert np.array_equal(e.download_duals(), o.duals())
        e.set_reparametrization_type(1); o.set_reparametrization_type(1)          # residual: joined launch unavailable
        for _ in range(3):
            e.compute_pass(1); o.ComputePass(1)
            assert np.array_equal(e.download_duals(), o.duals())
        e.set_reparametrization_type(0); o.set_reparametrization_type(0)
        e.compute_pass(4); o.ComputePass(4)                                       # the joined launch again (templates rebuilt)
        assert np.array_equal(e.download_duals(), o.duals())
        assert abs(e.lower_bound() - o.LowerBound()) <= 1e-9 * max(1.0, abs(o.LowerBound()))
    finally:
        e.close()


# ---- bench.py: C5 workload, keys of an N-rank line, partition injection ---------------------------------------------------------
def _bench(args, env=None, timeout=1500):
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LPMP_DIST_BACKEND")}
    e.update(env or {})
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "bench.py")] + args, text=True, cwd=ROOT, timeout=timeout, env=e)
    assert out.strip().splitlines()[-1].startswith('{"metric"'), out[-800:]
    return json.loads(out.strip().splitlines()[-1])


N_RANK_KEYS = ("compute_ms_per_pass", "exchange_ms_per_pass", "exchange_bytes_per_pass", "exchanges_per_pass", "redundant_fraction",
               "slowest_rank", "rank_stats", "scaling_model")


@pytest.mark.parametrize("order", ["index", "colour_major", "suggested"])
def test_bench_c5_small_on_one_gpu_is_checked_against_the_oracle_fixture(order):
    d = _bench(["--workload", "c5", "--c5-small", "--c5-order", order, "--steps", "4", "--warmup", "2", "--cpu-sample-grid", "16"])
    assert d["n_gpus"] == 1 and "labeling-list" in d["config"]["workload"] and d["scaling"] == "strong"
    oc = d["oracle_check"]
    assert oc["passes"] == 6 and oc["duals_bit_identical_to_oracle"] is True and oc["lb_rel_err"] < 1e-9
    assert d["roofline"] is not None and d["cpu_baseline"]["kind"] == "port" and d["value"] > 0
    for k in N_RANK_KEYS:
        assert d[k] is None, k


@pytest.mark.parametrize("workload", ["c3", "c4", "c5"])
def test_n_rank_keys_at_world_one_on_rccl(workload):
    """--force-dist at WORLD_SIZE 1: the rendezvous over the store, the self test of the collectives, init_process_group("nccl"),
    the probe leg — the keys an N-rank line carries, here with one rank"""
    extra = {"c3": ["--grid", "128"], "c4": ["--workload", "c4", "--c4-nodes", "20000", "--c4-edges", "100000"],
             "c5": ["--workload", "c5", "--c5-small"]}[workload]
    env = dict(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29581")
    d = _bench(["--force-dist", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"] + extra, env)
    assert d["n_gpus"] == 1 and d["backend"].startswith("rccl") and d["launch"]["physical_devices"] == 1
    assert d["launch"]["self_test"]["all_reduce"] == "ok" and d["launch"]["persistent_launches"] is True
    assert d["launch"]["device_identities"][0].startswith("pci=")
    for k in N_RANK_KEYS:
        assert d[k] is not None, k
    st = d["rank_stats"]
    assert len(st["per_rank"]["compute_ms_per_pass"]) == 1 and d["slowest_rank"] == 0
    assert d["compute_ms_per_pass"]["max"] > 0 and d["exchange_bytes_per_pass"]["sum"] == 0
    assert d["scaling_model"]["assumed_latency_us_per_exchange"] == 30.0 and d["scaling_model"]["projected_ms_per_pass"] > 0
    assert abs(d["dual_bound_gap"]) <= 1e-9
    if workload == "c5":
        assert d["oracle_check"]["duals_bit_identical_to_oracle"] is True and d["config"]["partitioner"].startswith("none")


def test_bench_gpus_2_c5_and_a_partition_file(tmp_path):
    """`python bench.py --gpus 2 --workload c5 --c5-small` (two ranks share the test box's GPU: gloo, persistent launches off by
    the drivers' own device check): lock step, gap 0, the state of BOTH ranks bit-identical to the oracle fixture; then the C4
    miniature with a hand-made partition file"""
    d = _bench(["--gpus", "2", "--workload", "c5", "--c5-small", "--steps", "4", "--warmup", "2", "--no-cpu-baseline"])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["schedule"] == "lockstep"
    assert d["launch"]["physical_devices"] == 1 and d["launch"]["persistent_launches"] is False and d["backend"] == "gloo"
    assert d["oracle_check"]["duals_bit_identical_to_oracle"] is True and abs(d["dual_bound_gap"]) <= 1e-12
    assert d["config"]["partitioner"].startswith(("builtin", "metis")) and 0 < d["config"]["cut_fraction"] < 1
    assert len(d["rank_stats"]["per_rank"]["exchange_ms_per_pass"]) == 2 and d["exchanges_per_pass"] > 0 and d["exchange_bytes_per_pass"]["max"] > 0
    assert d["scaling_model"]["note"] is not None                       # shared device: the projection says it means nothing
    # the same model in the order the engine suggests for it (a chain of relations through all factors), again on two ranks
    d = _bench(["--gpus", "2", "--workload", "c5", "--c5-small", "--c5-order", "suggested", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"])
    assert d["oracle_check"]["duals_bit_identical_to_oracle"] is True and abs(d["dual_bound_gap"]) <= 1e-12 and max(d["config"]["levels_per_direction"]) <= 12
    pf = tmp_path / "c4.part"
    np.savetxt(pf, (np.arange(20000) * 7 % 2).astype(np.int64), fmt="%d")
    d = _bench(["--gpus", "2", "--workload", "c4", "--c4-nodes", "20000", "--c4-edges", "100000", "--steps", "3", "--warmup", "1", "--no-cpu-baseline",
                "--no-compare-schedules", "--partition-file", str(pf)])
    assert d["config"]["partitioner"] == "file c4.part" and abs(d["dual_bound_gap"]) <= 1e-12 and 0.4 < d["config"]["cut_fraction"] < 0.6


# ---- the order the engine suggests ------------------------------------------------------------------------------------------------
def test_suggested_order_on_the_device_two_levels_and_the_oracles_duals(capfd):
    """a C3-shaped grid inserted row by row (79 dependent levels per direction at 40 x 40: the engine says so, once, on stderr);
    lpmp_plan_suggest_order's answer applied as a chain of AddFactorRelation calls: 2 levels per direction, consecutive passes
    join, and the duals are the oracle's run in that order bit for bit"""
    m = S.grid_model(40, 40, 32, order="row_major", seed=2)
    e = E.Engine(0)
    try:
        e.upload(m); e.set_reparametrization(M.REPAM_ANISOTROPIC)
        e.compute_pass(1)
        err = capfd.readouterr().err
        assert err.count("lpmp_plan_suggest_order") == 1 and "79 dependent levels" in err
        rank, k = e.plan.suggest_order(0)
        assert k == 2
        m2 = m.with_factor_order(rank)
        e.upload(m2); e.set_reparametrization(M.REPAM_ANISOTROPIC)
        assert [e.plan.schedule_info(d, M.REPAM_ANISOTROPIC)["n_levels"] for d in (0, 1)] == [2, 2]
        assert e.plan.pass_rotates(M.REPAM_ANISOTROPIC)
        o = Oracle(m2); o.set_reparametrization(M.REPAM_ANISOTROPIC)
        for n in (1, 4):
            e.compute_pass(n); o.ComputePass(n)
            assert np.array_equal(e.download_duals(), o.duals())
        assert abs(e.lower_bound() - o.LowerBound()) <= 1e-9 * abs(o.LowerBound())
        assert "lpmp_plan_suggest_order" not in capfd.readouterr().err
    finally:
        e.close()


def test_bench_under_torch_distributed_run_the_drivers_own_command():
    """the driver's N-GPU command verbatim, at N = 2 on the one GPU of the test box: `python -m torch.distributed.run --nnodes=1
    --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...` — the ranks meet at the AGENT's store, find
    that they share a physical device (gloo, persistent launches off), pass the self test, run the overlap schedule exactly"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LPMP_DIST_BACKEND")}
    out = subprocess.check_output([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                                   "--master-port", "29591", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--grid", "128", "--steps", "4", "--warmup", "2",
                                   "--no-cpu-baseline", "--no-compare-schedules"], text=True, cwd=ROOT, timeout=1500, env=env, stderr=subprocess.DEVNULL)
    lines = [l for l in out.strip().splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1                                               # rank 0 prints the one line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["launch"]["launcher"].startswith("external") and d["schedule"] == "overlap"
    assert d["launch"]["physical_devices"] == 1 and d["backend"] == "gloo" and d["launch"]["persistent_launches"] is False
    assert d["launch"]["self_test"]["all_reduce"] == "ok" and abs(d["dual_bound_gap"]) <= 1e-12
    for k in N_RANK_KEYS:
        assert d[k] is not None, k
    assert len(d["rank_stats"]["per_rank"]["compute_ms_per_pass"]) == 2 and d["redundant_fraction"] > 0


@pytest.mark.parametrize("seed", range(12))
def test_random_models_in_the_suggested_order_equal_the_oracle(seed):
    """random grids, graphs, C5-style and multicut models (tests/test_graph_host.py) run in the order the engine suggests for them:
    the oracle's duals and bound in that order, bit for bit, plain and residual sends"""
    from tests.test_graph_host import _random_models
    from tests.test_fuzz_gpu import random_model
    # (odd seeds: models of every message schedule — higher factors updated too)
    m = _random_models(100 + seed) if seed % 2 == 0 else random_model(np.random.default_rng(9000 + seed))
    rank, _ = E.Plan(m).suggest_order(seed)
    m2 = m.with_factor_order(rank)
    o = Oracle(m2)
    e = E.Engine(0)
    try:
        e.upload(m2)
        for rtype in (0, 1):
            o.set_reparametrization_type(rtype); e.set_reparametrization_type(rtype)
            for mode in (M.REPAM_ANISOTROPIC, M.REPAM_DAMPED_UNIFORM):
                o.set_reparametrization(mode); e.set_reparametrization(mode)
                for n in (1, 2):
                    o.ComputePass(n); e.compute_pass(n)
                    assert np.array_equal(e.download_duals(), o.duals()), (seed, rtype, mode, n)
                assert abs(e.lower_bound() - o.LowerBound()) <= 1e-9 * max(1.0, abs(o.LowerBound()))
    finally:
        e.close()


@pytest.mark.parametrize("schedule", ["overlap", "lockstep", "boundary"])
def test_cpp_host_line_says_where_the_time_went(schedule):
    """tools/mgpu_rccl_driver.cpp --time K: after the timed passes the same passes run once more with every exchange (pack -> group
    of ncclSend / ncclRecv -> unpack) bracketed by events (rccl_world::probe_run): compute / exchange ms per pass, exchanges and
    bytes per pass in the line, maximum over the ranks — the C++ host's counterpart of bench.py's per-rank split.  Parts of one rank
    on the one GPU here (the transfers are device copies)."""
    from lp_mp_amd import build as B
    if not B.have_rccl():
        pytest.skip("no <rccl/rccl.h> on this box")
    exe = B.build_mgpu_driver()
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    args = [exe, "--H", "32", "--W", "32", "--L", "8", "--parts-per-rank", "2", "--passes", "4", "--time", "6"]
    args += [] if schedule == "boundary" else ["--schedule", schedule]
    out = subprocess.check_output(args, text=True, env=env, timeout=600, stderr=subprocess.DEVNULL)
    d = json.loads(out.strip().splitlines()[-1])
    assert d["ms_per_pass"] > 0 and d["compute_ms_per_pass"] > 0 and d["exchange_ms_per_pass"] > 0
    assert d["exchanges_per_pass"] > 0 and d["exchange_bytes_out_per_pass"] > 0 and d["lower_bound_after"] > d["lower_bound_before"]
    if schedule == "overlap":
        assert abs(d["exchanges_per_pass"] - 2 / 6) < 1e-3            # 6 passes in chunks of 5 + 1: two exchanges (printed with three digits)
