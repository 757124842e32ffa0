"""Lock-step partitioned sweep (lp_mp_amd/lockstep.py): several parts run THE unpartitioned sweep — duals and bound of the
oracle on the unpartitioned model, bit for bit, for any partition.  CPU: oracle-backed engine stand-ins in one process and
over torch.distributed (gloo, world size 2); GPU: real engines, several parts on the one device of the test box."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from lp_mp_amd import lockstep as LS
from lp_mp_amd import model as M
from lp_mp_amd import multi_gpu as MG
from lp_mp_amd import synthetic as S
from oracle.binding import Oracle
from tests.mgpu_helpers import OracleEngine

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _graph(n, m, L, world, seed=1, colour_major=False):
    rank = None
    if colour_major:                                  # LockstepGraph's default: the variables renamed colour by colour
        from lp_mp_amd import ordering as O
        rank = O.colour_major_order(n, *S.counter_graph_edges(n, m, seed), seed=seed)
    g = S.counter_graph_model(n, m, L, seed, rank=rank)
    ei = g.m_left[0::2].astype(np.int64); ej = g.m_left[1::2].astype(np.int64)
    return dict(n_vars=n, L=L, ei=ei, ej=ej, un=g.dual_data[: n * L], tables=g.const_data, potts=None, pairwise="dense",
                part_of=MG.graph_partition(n, ei, ej, world), world=world)


def _strips(H, W, L, world, pairwise, order, seed=1):
    ei, ej = MG.strip_global_edges(H, W, world, order)
    un, tables, potts = MG.strip_costs(H, W, L, world, pairwise, seed)
    return dict(n_vars=world * H * W, L=L, ei=ei, ej=ej, un=un, tables=tables, potts=potts, pairwise=pairwise,
                part_of=np.repeat(np.arange(world), H * W), world=world)


def _global_of(c):
    return S.mrf_model(c["n_vars"], c["L"], c["ei"], c["ej"], c["un"], tables=c["tables"], potts=c["potts"])


def _parts_of(c, mode):
    return LS.lockstep_mrf(c["n_vars"], c["L"], c["ei"], c["ej"], c["part_of"], c["world"], mode, c["un"], c["tables"], c["potts"], pairwise=c["pairwise"])


def _assert_equals_global(c, parts, local_duals, ref):
    gd, g_off = ref.duals(), ref_offsets(c)
    for p, d in zip(parts, local_duals):
        lo = p.model.dual_offsets(); nv = p.vars_global.shape[0]
        for fl in range(p.model.n_factors):
            if fl < nv and p.is_ghost[fl]:
                continue
            g = int(p.vars_global[fl]) if fl < nv else c["n_vars"] + int(p.edges_global[fl - nv])
            assert np.array_equal(d[lo[fl]:lo[fl + 1]], gd[g_off[g]:g_off[g + 1]]), (p.rank, fl)


def ref_offsets(c):
    return _global_of(c).dual_offsets()


CASES = {
    "random graph, 4 parts (61 % of the edges cut)": lambda: _graph(400, 1200, 4, 4),
    "random graph, 7 parts": lambda: _graph(300, 700, 3, 7, seed=3),
    "random graph in colour-major order, 5 parts": lambda: _graph(400, 1300, 4, 5, seed=4, colour_major=True),
    "colour-major strips, 3 parts": lambda: _strips(8, 8, 3, 3, "dense", "colour_major"),
    "row-major strips, 3 parts": lambda: _strips(6, 7, 3, 3, "dense", "row_major"),
    "Potts strips, 4 parts": lambda: _strips(6, 6, 4, 4, "potts", "colour_major"),
}


@pytest.mark.parametrize("name", list(CASES))
@pytest.mark.parametrize("mode", [M.REPAM_ANISOTROPIC, M.REPAM_DAMPED_UNIFORM])
def test_lockstep_parts_run_the_unpartitioned_sweep(name, mode):
    c = CASES[name]()
    ref = Oracle(_global_of(c)); ref.set_reparametrization(mode)
    sched, parts = _parts_of(c, mode)
    duals = [p.model.dual_data.copy() for p in parts]
    sweeps = [LS.LockstepSweep(torch, p, sched, OracleEngine(p.model, d), torch.from_numpy(d)) for p, d in zip(parts, duals)]
    for n in (1, 2, 3):                                     # separate calls: the copies agree when a call returns
        ref.ComputePass(n)
        LS.run_lockstep(sweeps, n)
        _assert_equals_global(c, parts, duals, ref)
        lb = sum(s.local_lower_bound() for s in sweeps)
        assert abs(lb - ref.LowerBound()) <= 1e-12 * max(1.0, abs(ref.LowerBound()))
    # every edge touching a part's variables is in the part; a pairwise factor is counted once
    assert sum(int(p.owned.sum()) for p in parts) == c["n_vars"] + c["ei"].shape[0]


@pytest.mark.parametrize("name", ["mixed_mrf", "multicut", "c5"])
@pytest.mark.parametrize("world,mode", [(2, M.REPAM_ANISOTROPIC), (3, M.REPAM_DAMPED_UNIFORM), (4, M.REPAM_ANISOTROPIC2)])
def test_lockstep_parts_of_any_left_schedule_model_run_the_unpartitioned_sweep(name, world, mode):
    """lockstep_model: ragged label counts, mixed dense / Potts edges, labeling-list factors with three and four variables (the
    whole dual of such a factor is one exchange unit and goes to every other rank holding it): duals of every copy and the bound
    equal the oracle's on the unpartitioned model, random partitions of the variables"""
    from tests.test_multi_gpu import _general_models
    gm = _general_models()[name]
    rng = np.random.default_rng(11 + world)
    is_right = np.zeros(gm.n_factors, bool); is_right[gm.m_right] = True
    part_of = rng.integers(0, world, gm.n_factors)
    part_of[np.nonzero(~is_right)[0][:world]] = np.arange(world)           # every rank has a variable
    ref = Oracle(gm); ref.set_reparametrization(mode)
    sched, parts = LS.lockstep_model(gm, part_of, world, mode)
    assert sum(int(p.owned.sum()) for p in parts) == gm.n_factors
    assert max(int(np.diff(sched.dest_off).max()), 0) >= 1 and (name == "mixed_mrf" or world == 2 or int(np.diff(sched.dest_off).max()) >= 2)
    duals = [p.model.dual_data.copy() for p in parts]
    sweeps = [LS.LockstepSweep(torch, p, sched, OracleEngine(p.model, d), torch.from_numpy(d)) for p, d in zip(parts, duals)]
    g_off = gm.dual_offsets()
    for n in (1, 2, 2):
        ref.ComputePass(n)
        LS.run_lockstep(sweeps, n)
        gd = ref.duals()
        for p, d in zip(parts, duals):
            lo = p.model.dual_offsets()
            var_ghost = dict(zip(p.vars_global.tolist(), p.is_ghost.tolist()))
            for fl, g in enumerate(p.factors_global.tolist()):
                if var_ghost.get(g, False):
                    continue
                assert np.array_equal(d[lo[fl]:lo[fl + 1]], gd[g_off[g]:g_off[g + 1]]), (p.rank, fl, g)
        lb = sum(s.local_lower_bound() for s in sweeps)
        assert abs(lb - ref.LowerBound()) <= 1e-12 * max(1.0, abs(ref.LowerBound()))


def test_isolated_variables_odd_label_counts_and_bad_partitions():
    """variables without any edge are updated where they live; three labels (run-time-dims classes on the device); a rank
    without variables is refused"""
    n, L, world = 60, 3, 3
    rng = np.random.default_rng(3)
    ei = rng.integers(0, 40, 90); ej = rng.integers(0, 40, 90)
    keep = ei != ej
    e = np.unique(np.stack([np.minimum(ei, ej)[keep], np.maximum(ei, ej)[keep]], 1), axis=0)      # variables 40 .. 59 have no edge
    c = dict(n_vars=n, L=L, ei=e[:, 0], ej=e[:, 1], un=S.u01(n * L, 4, 0), tables=S.u01(e.shape[0] * L * L, 4, n * L), potts=None,
             pairwise="dense", part_of=rng.integers(0, world, n), world=world)
    ref = Oracle(_global_of(c)); ref.set_reparametrization(M.REPAM_ANISOTROPIC2)
    sched, parts = _parts_of(c, M.REPAM_ANISOTROPIC2)
    duals = [p.model.dual_data.copy() for p in parts]
    sweeps = [LS.LockstepSweep(torch, p, sched, OracleEngine(p.model, d), torch.from_numpy(d)) for p, d in zip(parts, duals)]
    ref.ComputePass(2); LS.run_lockstep(sweeps, 2)
    _assert_equals_global(c, parts, duals, ref)
    assert abs(sum(s.local_lower_bound() for s in sweeps) - ref.LowerBound()) <= 1e-12 * max(1.0, abs(ref.LowerBound()))
    with pytest.raises(ValueError, match="without variables"):
        LS.lockstep_mrf(n, L, e[:, 0], e[:, 1], np.zeros(n, np.int64), 2, M.REPAM_ANISOTROPIC, c["un"], c["tables"])


def test_colour_major_strips_need_two_exchanges_per_pass():
    c = _strips(8, 8, 3, 4, "dense", "colour_major")
    sched, _ = _parts_of(c, M.REPAM_ANISOTROPIC)
    prog = sched.program(6)
    assert sched.n_levels == (4, 4)
    assert sum(1 for s in prog if s[0] == "halo") == 2 * 6 + 1          # + the one that closes the call
    one = _strips(8, 8, 3, 1, "dense", "colour_major")
    s1, _ = _parts_of(one, M.REPAM_ANISOTROPIC)
    assert [s[0] for s in s1.program(5)] == ["run"]                      # nothing cut: the whole call is one schedule


@pytest.mark.parametrize("world", [4, 6])
@pytest.mark.parametrize("pairwise,order", [("dense", "colour_major"), ("potts", "colour_major"), ("dense", "row_major")])
def test_strip_parts_from_the_three_strip_proxy_equal_those_of_the_true_world(world, pairwise, order):
    """bench.py's strips never build the global structure of `world` strips (strips_lockstep_part): rows, model, cost
    stream positions, exchange plans and the program must be those of the true global structure"""
    H, W, L = 6, 6, 3
    for rank in range(world):
        got_s, got = LS.strips_lockstep_part(H, W, L, pairwise, order, rank, world, M.REPAM_ANISOTROPIC, 5, proxy=True)
        ref_s, ref = LS.strips_lockstep_part(H, W, L, pairwise, order, rank, world, M.REPAM_ANISOTROPIC, 5, proxy=False)
        assert np.array_equal(got.model.m_left, ref.model.m_left) and np.array_equal(got.model.m_right, ref.model.m_right)
        assert np.array_equal(got.owned, ref.owned) and np.array_equal(got.is_ghost, ref.is_ghost)
        for a, b in zip(got.const_fill + got.dual_fill, ref.const_fill + ref.dual_fill):
            assert a[:3] == b[:3] and np.array_equal(a[3], b[3])
        for d in (0, 1):
            assert len(got.rows[d]) == len(ref.rows[d])
            for ra, rb in zip(got.rows[d], ref.rows[d]):
                assert all(np.array_equal(x, y) for x, y in zip(ra, rb))
        pa, pb = got_s.program(3), ref_s.program(3)
        assert [(s[0], s[1] if s[0] == "run" else None) for s in pa] == [(s[0], s[1] if s[0] == "run" else None) for s in pb]
        dummy = torch.zeros(int(ref.model.dual_sizes().sum()), dtype=torch.float64)
        sa, sb = LS.LockstepSweep(torch, got, got_s, None, dummy), LS.LockstepSweep(torch, ref, ref_s, None, dummy)
        for xa, xb in zip(pa, pb):
            if xa[0] == "halo":
                ha, hb = sa._halo_plan(xa[1]), sb._halo_plan(xb[1])
                assert torch.equal(ha[0], hb[0]) and torch.equal(ha[2], hb[2])
                assert np.array_equal(ha[1], hb[1]) and np.array_equal(ha[3], hb[3])


WORKER = r"""
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, {root!r})
from lp_mp_amd import model as M, multi_gpu as MG, lockstep as LS, synthetic as S
from tests.mgpu_helpers import OracleEngine
from tests.test_lockstep import _graph
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
c = _graph(300, 800, 3, world, seed=2)
sched, parts = LS.lockstep_mrf(c["n_vars"], c["L"], c["ei"], c["ej"], c["part_of"], world, M.REPAM_ANISOTROPIC, c["un"], c["tables"], only=rank)
p = parts[0]
d = p.model.dual_data.copy()
sw = LS.LockstepSweep(torch, p, sched, OracleEngine(p.model, d), torch.from_numpy(d))
comm = MG.DistComm(dist, torch)
sw.compute_pass(comm, 2); sw.compute_pass(comm, 1)
lb = comm.all_reduce_sum(sw.local_lower_bound())
np.save(os.path.join({out!r}, f"ls_duals_{{rank}}.npy"), d)
if rank == 0:
    np.save(os.path.join({out!r}, "ls_lb.npy"), np.array([lb]))
dist.destroy_process_group()
"""


@pytest.mark.parametrize("world", [2, 8])
def test_gloo_processes_equal_the_unpartitioned_oracle(tmp_path, world):
    """one process per part over torch.distributed (gloo), 2 and 8 ranks (the node size north_star names): every rank's duals are
    the unpartitioned oracle's"""
    script = tmp_path / "ls_worker.py"
    script.write_text(WORKER.format(root=ROOT, out=str(tmp_path)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
                           "--master-addr", "127.0.0.1", "--master-port", str(29534 + world), str(script)], env=env, cwd=ROOT, timeout=600)
    c = _graph(300, 800, 3, world, seed=2)
    ref = Oracle(_global_of(c)); ref.set_reparametrization(M.REPAM_ANISOTROPIC)
    ref.ComputePass(3)
    _, parts = _parts_of(c, M.REPAM_ANISOTROPIC)
    _assert_equals_global(c, parts, [np.load(tmp_path / f"ls_duals_{k}.npy") for k in range(world)], ref)
    assert abs(np.load(tmp_path / "ls_lb.npy")[0] - ref.LowerBound()) <= 1e-12 * abs(ref.LowerBound())


@pytest.mark.gpu
@pytest.mark.parametrize("name,L", [("strips", 32), ("strips", 8), ("graph", 16), ("potts", 8), ("row_major_strips", 16), ("row_major_potts", 5)])
def test_lockstep_on_device_equals_the_unpartitioned_engine_and_oracle(name, L):
    """real HIP engines, all parts on the one GPU of the test box: duals of the oracle on the unpartitioned model, bit for bit"""
    from lp_mp_amd import engine as E
    c = {"strips": lambda: _strips(12, 10, L, 3, "dense", "colour_major", 5), "graph": lambda: _graph(1500, 6000, L, 4, 2),
         "potts": lambda: _strips(10, 12, L, 4, "potts", "colour_major", 7),
         # deep custom schedules: every segment a chain launch whose records hand over through the mailbox (plan.cpp)
         "row_major_strips": lambda: _strips(14, 12, L, 3, "dense", "row_major", 9),
         "row_major_potts": lambda: _strips(12, 14, L, 3, "potts", "row_major", 11)}[name]()
    gm = _global_of(c)
    ref = Oracle(gm); ref.set_reparametrization(M.REPAM_ANISOTROPIC)
    sched, parts = _parts_of(c, M.REPAM_ANISOTROPIC)
    dev = torch.device("cuda:0")
    sweeps, tensors = [], []
    for p in parts:
        dual = torch.from_numpy(p.model.dual_data.copy()).to(dev)
        eng = E.Engine(0); eng.set_stream(torch.cuda.current_stream().cuda_stream)
        eng.upload(p.model, dual_dev=dual.data_ptr(), keep=dual)
        eng.set_reparametrization(M.REPAM_ANISOTROPIC)
        sweeps.append(LS.LockstepSweep(torch, p, sched, eng, dual)); tensors.append(dual)
    try:
        for n in (1, 3):
            ref.ComputePass(n)
            LS.run_lockstep(sweeps, n)
            torch.cuda.synchronize()
            _assert_equals_global(c, parts, [t.cpu().numpy() for t in tensors], ref)
            lb = sum(s.local_lower_bound() for s in sweeps)
            assert abs(lb - ref.LowerBound()) <= 1e-9 * max(1.0, abs(ref.LowerBound()))
    finally:
        for s in sweeps:
            s.engine.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name,world", [("mixed_mrf", 3), ("multicut", 3), ("c5", 2), ("c5", 4)])
def test_lockstep_of_general_models_on_device_equals_the_oracle(name, world):
    """lockstep_model on real engines (labeling-list kernels, run-time label counts): every copy of every factor and the bound
    are the oracle's on the unpartitioned model"""
    from lp_mp_amd import engine as E
    from tests.test_multi_gpu import _general_models
    gm = _general_models()[name]
    rng = np.random.default_rng(3 + world)
    is_right = np.zeros(gm.n_factors, bool); is_right[gm.m_right] = True
    part_of = rng.integers(0, world, gm.n_factors)
    part_of[np.nonzero(~is_right)[0][:world]] = np.arange(world)
    mode = M.REPAM_ANISOTROPIC
    ref = Oracle(gm); ref.set_reparametrization(mode)
    sched, parts = LS.lockstep_model(gm, part_of, world, mode)
    dev = torch.device("cuda:0")
    sweeps, tensors = [], []
    for p in parts:
        dual = torch.from_numpy(p.model.dual_data.copy()).to(dev)
        eng = E.Engine(0); eng.set_stream(torch.cuda.current_stream().cuda_stream)
        eng.upload(p.model, dual_dev=dual.data_ptr(), keep=dual); eng.set_reparametrization(mode)
        sweeps.append(LS.LockstepSweep(torch, p, sched, eng, dual)); tensors.append(dual)
    g_off = gm.dual_offsets()
    try:
        for n in (1, 3):
            ref.ComputePass(n); LS.run_lockstep(sweeps, n); torch.cuda.synchronize()
            gd = ref.duals()
            for p, t in zip(parts, tensors):
                d, lo = t.cpu().numpy(), p.model.dual_offsets()
                ghosts = set(p.vars_global[p.is_ghost].tolist())
                for fl, g in enumerate(p.factors_global.tolist()):
                    if g not in ghosts:
                        assert np.array_equal(d[lo[fl]:lo[fl + 1]], gd[g_off[g]:g_off[g + 1]]), (p.rank, fl, g)
            lb = sum(s.local_lower_bound() for s in sweeps)
            assert abs(lb - ref.LowerBound()) <= 1e-9 * max(1.0, abs(ref.LowerBound()))
    finally:
        for s in sweeps:
            s.engine.close()


@pytest.mark.gpu
@pytest.mark.parametrize("L,rows", [(16, False), (16, True), (5, False), (40, False)])
def test_halo_kernels_copy_the_listed_vectors(L, rows):
    """lpmp_halo_pack / _unpack through the C ABI: the listed slices of the packed dual array, in list order — also when the
    pairwise duals live in the engine-private rows layout (offsets are translated) and for vectors longer than a quarter wave"""
    from lp_mp_amd import engine as E
    g = S.grid_model(6, 7, L, seed=3)
    dev = torch.device("cuda:0")
    eng = E.Engine(0); eng.set_stream(torch.cuda.current_stream().cuda_stream)
    eng.upload(g, rows_layout=rows); eng.set_reparametrization(M.REPAM_ANISOTROPIC)
    eng.compute_pass(2)
    before = eng.download_duals()
    off = g.dual_offsets(); nv = 42
    rng = np.random.default_rng(L)
    pw = nv + rng.choice(g.n_factors - nv, 20, replace=False)
    side = rng.integers(0, 2, 20)
    out_off = off[pw] + side * L; out_len = np.full(20, L)
    un = rng.choice(nv, 9, replace=False)
    in_off = np.concatenate([off[pw[:7]] + (1 - side[:7]) * L, off[un]]); in_len = np.full(16, L)
    h = eng.halo_create(out_off, out_len, in_off, in_len)
    try:
        assert eng.halo_sizes(h) == (20 * L, 16 * L)
        send = torch.zeros(20 * L, dtype=torch.float64, device=dev)
        eng.halo_pack(h, send.data_ptr()); torch.cuda.synchronize()
        assert np.array_equal(send.cpu().numpy(), np.concatenate([before[o:o + L] for o in out_off]))
        assert np.array_equal(eng.download_duals(), before)                       # pack reads only
        recv = torch.from_numpy(rng.uniform(-1, 1, 16 * L)).to(dev)
        eng.halo_unpack(h, recv.data_ptr()); torch.cuda.synchronize()
        want = before.copy()
        for k, o in enumerate(in_off):
            want[o:o + L] = recv.cpu().numpy()[k * L:(k + 1) * L]
        assert np.array_equal(eng.download_duals(), want)
        o = Oracle(g); o.set_duals(want)
        assert abs(eng.lower_bound() - o.LowerBound()) <= 1e-9 * max(1.0, abs(o.LowerBound()))   # tracked bounds were dropped
    finally:
        eng.halo_destroy(h); eng.close()


WORKER_MODEL = r"""
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, {root!r})
from lp_mp_amd import model as M, multi_gpu as MG, lockstep as LS
from tests.mgpu_helpers import OracleEngine
from tests.test_multi_gpu import _general_models
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
gm = _general_models()["c5"]
part_of = MG.graph_partition_model(gm, world)
sched, parts = LS.lockstep_model(gm, part_of, world, M.REPAM_ANISOTROPIC, only=rank)
p = parts[0]
d = p.model.dual_data.copy()
sw = LS.LockstepSweep(torch, p, sched, OracleEngine(p.model, d), torch.from_numpy(d))
comm = MG.DistComm(dist, torch)
sw.compute_pass(comm, 2); sw.compute_pass(comm, 1)
lb = comm.all_reduce_sum(sw.local_lower_bound())
np.save(os.path.join({out!r}, f"gm_duals_{{rank}}.npy"), d)
np.save(os.path.join({out!r}, f"gm_factors_{{rank}}.npy"), p.factors_global)
if rank == 0:
    np.save(os.path.join({out!r}, "gm_lb.npy"), np.array([lb]))
dist.destroy_process_group()
"""


def test_three_process_gloo_run_of_the_c5_shape_equals_the_unpartitioned_oracle(tmp_path):
    """C5 in miniature (Potts grid + triplets + quadruples, one factor graph) in three processes over torch.distributed, the
    variables split by graph_partition_model: every rank's copies are the oracle's duals after 3 passes, the bound its bound"""
    from tests.test_multi_gpu import _general_models
    script = tmp_path / "gm_worker.py"
    script.write_text(WORKER_MODEL.format(root=ROOT, out=str(tmp_path)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3",
                           "--master-addr", "127.0.0.1", "--master-port", "29541", str(script)], env=env, cwd=ROOT, timeout=600)
    gm = _general_models()["c5"]
    ref = Oracle(gm); ref.set_reparametrization(M.REPAM_ANISOTROPIC); ref.ComputePass(3)
    gd, g_off = ref.duals(), gm.dual_offsets()
    part_of = MG.graph_partition_model(gm, 3)
    is_right = np.zeros(gm.n_factors, bool); is_right[gm.m_right] = True
    checked = 0
    for k in range(3):
        d, fk = np.load(tmp_path / f"gm_duals_{k}.npy"), np.load(tmp_path / f"gm_factors_{k}.npy")
        lo = np.concatenate([[0], np.cumsum(gm.dual_sizes()[fk])])
        for fl, g in enumerate(fk.tolist()):
            if is_right[g] or part_of[g] == k:               # (ghosts of remote variables are never updated)
                assert np.array_equal(d[lo[fl]:lo[fl + 1]], gd[g_off[g]:g_off[g + 1]]), (k, g)
                checked += 1
    assert checked > gm.n_factors                          # higher factors are held (and checked) on several ranks
    assert abs(np.load(tmp_path / "gm_lb.npy")[0] - ref.LowerBound()) <= 1e-12 * max(1.0, abs(ref.LowerBound()))


WORKER_STRIPS = r"""
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, {root!r})
from lp_mp_amd import model as M, multi_gpu as MG, lockstep as LS
from tests.mgpu_helpers import OracleEngine, materialise_fills
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
sched, p = LS.strips_lockstep_part(6, 6, 3, "dense", "colour_major", rank, world, M.REPAM_ANISOTROPIC, 5)   # 3-strip proxy, shifted
materialise_fills(p)
d = p.model.dual_data.copy()
sw = LS.LockstepSweep(torch, p, sched, OracleEngine(p.model, d), torch.from_numpy(d))
comm = MG.DistComm(dist, torch)
sw.compute_pass(comm, 3)
lb = comm.all_reduce_sum(sw.local_lower_bound())
np.save(os.path.join({out!r}, f"st_duals_{{rank}}.npy"), d)
if rank == 0:
    np.save(os.path.join({out!r}, "st_lb.npy"), np.array([lb]))
dist.destroy_process_group()
"""


def test_four_process_gloo_run_of_proxy_built_strips_equals_the_unpartitioned_oracle(tmp_path):
    """what bench.py --gpus 4 --schedule lockstep does per rank (part and schedule from the 3-strip proxy, cost stream positions
    and peer ranks shifted into the true world), over torch.distributed: the oracle's sweep of the whole 24 x 6 grid"""
    script = tmp_path / "st_worker.py"
    script.write_text(WORKER_STRIPS.format(root=ROOT, out=str(tmp_path)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4",
                           "--master-addr", "127.0.0.1", "--master-port", "29537", str(script)], env=env, cwd=ROOT, timeout=600)
    c = _strips(6, 6, 3, 4, "dense", "colour_major", 5)
    ref = Oracle(_global_of(c)); ref.set_reparametrization(M.REPAM_ANISOTROPIC)
    ref.ComputePass(3)
    _, parts = _parts_of(c, M.REPAM_ANISOTROPIC)
    _assert_equals_global(c, parts, [np.load(tmp_path / f"st_duals_{k}.npy") for k in range(4)], ref)
    assert abs(np.load(tmp_path / "st_lb.npy")[0] - ref.LowerBound()) <= 1e-12 * abs(ref.LowerBound())


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(10))
def test_lockstep_random_graphs_partitions_and_modes_on_device(seed):
    """random graph, random (unbalanced, scattered) partition, random label count / pairwise kind / weight mode, separate calls
    of random length: every part's duals and the summed bound are the oracle's on the unpartitioned model"""
    from lp_mp_amd import engine as E
    rng = np.random.default_rng(100 + seed)
    n = int(rng.integers(40, 400)); m = int(rng.integers(n, 4 * n)); L = int(rng.choice([2, 3, 4, 5, 8, 16, 32]))
    world = int(rng.integers(2, 6)); pairwise = "potts" if rng.uniform() < 0.4 else "dense"
    mode = [M.REPAM_ANISOTROPIC, M.REPAM_ANISOTROPIC2, M.REPAM_UNIFORM, M.REPAM_DAMPED_UNIFORM][int(rng.integers(0, 4))]
    a = rng.integers(0, n, 2 * m); b = rng.integers(0, n, 2 * m)
    keep = a != b
    e = np.unique(np.stack([np.minimum(a, b)[keep], np.maximum(a, b)[keep]], 1), axis=0)[:m]
    part_of = rng.integers(0, world, n); part_of[:world] = np.arange(world)          # every rank owns something
    c = dict(n_vars=n, L=L, ei=e[:, 0], ej=e[:, 1], un=S.u01(n * L, seed, 0), pairwise=pairwise, part_of=part_of, world=world,
             tables=S.u01(e.shape[0] * L * L, seed, n * L) if pairwise == "dense" else None,
             potts=S.u01(e.shape[0], seed, n * L) if pairwise == "potts" else None)
    ref = Oracle(_global_of(c)); ref.set_reparametrization(mode)
    sched, parts = _parts_of(c, mode)
    dev = torch.device("cuda:0")
    sweeps, tensors = [], []
    for p in parts:
        dual = torch.from_numpy(p.model.dual_data.copy()).to(dev)
        eng = E.Engine(0); eng.set_stream(torch.cuda.current_stream().cuda_stream)
        # (tests/stress_lockstep_mailbox.py runs these cases for many more seeds, also with LPMP_STRESS_OVERLAP=1: the overlapped
        # program, and LPMP_STRESS_ROWS=1: parts on the rows layout)
        eng.upload(p.model, dual_dev=dual.data_ptr(), keep=dual, rows_layout=bool(os.environ.get("LPMP_STRESS_ROWS"))); eng.set_reparametrization(mode)
        sweeps.append(LS.LockstepSweep(torch, p, sched, eng, dual, overlap_exchange=bool(os.environ.get("LPMP_STRESS_OVERLAP")))); tensors.append(dual)
    try:
        for k in rng.integers(1, 4, 3):
            ref.ComputePass(int(k)); LS.run_lockstep(sweeps, int(k))
            for s in sweeps:
                s.engine.synchronize()                     # (rows layout: the packed buffer is written out here)
            _assert_equals_global(c, parts, [t.cpu().numpy() for t in tensors], ref)
            lb = sum(s.local_lower_bound() for s in sweeps)
            assert abs(lb - ref.LowerBound()) <= 1e-9 * max(1.0, abs(ref.LowerBound())), (seed, lb, ref.LowerBound())
    finally:
        for s in sweeps:
            s.engine.close()


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(8))
def test_lockstep_random_general_models_partitions_and_modes_on_device(seed):
    """random multicut-style and C5-style models (labeling-list factors of three and four edge variables, optionally a Potts grid
    beside them), random scattered partitions, random weight mode, separate calls of random length: every copy of every factor and
    the summed bound are the oracle's on the unpartitioned model"""
    from lp_mp_amd import engine as E
    rng = np.random.default_rng(500 + seed)
    if rng.uniform() < 0.5:
        gm = S.multicut_triangle_model(int(rng.integers(8, 60)), int(rng.integers(5, 90)), seed=seed)
    else:
        gm = S.c5_model(int(rng.integers(2, 7)), int(rng.integers(2, 7)), int(rng.choice([2, 3, 4, 8])), int(rng.integers(12, 80)), int(rng.integers(4, 60)),
                        int(rng.integers(2, 30)), seed=seed, window=int(rng.integers(6, 40)), colour_edge_vars=bool(rng.uniform() < 0.5))
    world = int(rng.integers(2, 6))
    mode = [M.REPAM_ANISOTROPIC, M.REPAM_ANISOTROPIC2, M.REPAM_UNIFORM, M.REPAM_DAMPED_UNIFORM][int(rng.integers(0, 4))]
    is_right = np.zeros(gm.n_factors, bool); is_right[gm.m_right] = True
    variables = np.nonzero(~is_right)[0]
    if variables.shape[0] < world:
        world = 2
    part_of = rng.integers(0, world, gm.n_factors)
    part_of[variables[:world]] = np.arange(world)
    ref = Oracle(gm); ref.set_reparametrization(mode)
    sched, parts = LS.lockstep_model(gm, part_of, world, mode)
    dev = torch.device("cuda:0")
    sweeps, tensors = [], []
    for p in parts:
        dual = torch.from_numpy(p.model.dual_data.copy()).to(dev)
        eng = E.Engine(0); eng.set_stream(torch.cuda.current_stream().cuda_stream)
        eng.upload(p.model, dual_dev=dual.data_ptr(), keep=dual, rows_layout=bool(os.environ.get("LPMP_STRESS_ROWS"))); eng.set_reparametrization(mode)
        sweeps.append(LS.LockstepSweep(torch, p, sched, eng, dual, overlap_exchange=bool(os.environ.get("LPMP_STRESS_OVERLAP")))); tensors.append(dual)
    g_off = gm.dual_offsets()
    try:
        for k in rng.integers(1, 4, 3):
            ref.ComputePass(int(k)); LS.run_lockstep(sweeps, int(k)); torch.cuda.synchronize()
            gd = ref.duals()
            for p, t in zip(parts, tensors):
                d, lo = t.cpu().numpy(), p.model.dual_offsets()
                ghosts = set(p.vars_global[p.is_ghost].tolist())
                for fl, g in enumerate(p.factors_global.tolist()):
                    if g not in ghosts:
                        assert np.array_equal(d[lo[fl]:lo[fl + 1]], gd[g_off[g]:g_off[g + 1]]), (seed, p.rank, fl, g)
            lb = sum(s.local_lower_bound() for s in sweeps)
            assert abs(lb - ref.LowerBound()) <= 1e-9 * max(1.0, abs(ref.LowerBound())), (seed, lb, ref.LowerBound())
    finally:
        for s in sweeps:
            s.close(); s.engine.close()


@pytest.mark.gpu
def test_lockstep_model_driver_on_one_gpu():
    """LockstepModel (the one-process-per-GPU driver of lockstep_model) without a process group: one part = the whole model, its
    passes and bound are the oracle's"""
    from tests.test_multi_gpu import _general_models
    gm = _general_models()["c5"]
    d = LS.LockstepModel(torch, None, gm, np.zeros(gm.n_factors, np.int64), M.REPAM_ANISOTROPIC)
    ref = Oracle(gm); ref.set_reparametrization(M.REPAM_ANISOTROPIC)
    try:
        assert abs(d.lower_bound() - ref.LowerBound()) <= 1e-9 * max(1.0, abs(ref.LowerBound()))
        d.prepare_passes(3); d.compute_pass(3); ref.ComputePass(3)
        torch.cuda.synchronize()
        assert np.array_equal(d.dualt.cpu().numpy(), ref.duals())
        assert abs(d.lower_bound() - ref.LowerBound()) <= 1e-9 * max(1.0, abs(ref.LowerBound()))
        assert d.halo_steps_per_pass() == 0 and d.global_updates_per_pass > 0
    finally:
        d.sweep.close(); d.engine.close()


def test_colour_major_order_cuts_the_exchanges_of_a_random_graph():
    """LockstepGraph's default variable order (ordering.colour_major_order): one dependent level per colour, so the sweep
    needs an exchange per colour instead of one per level of the generator's index order"""
    n, m, L, world = 3000, 15000, 4, 4
    counts = {}
    for cm in (False, True):
        c = _graph(n, m, L, world, seed=2, colour_major=cm)
        sched, _ = _parts_of(c, M.REPAM_ANISOTROPIC)
        counts[cm] = (sched.n_levels, sum(1 for s in sched.program(4) if s[0] == "halo") / 4)
    assert counts[True][0][0] < counts[False][0][0] / 2 and counts[True][1] < counts[False][1] / 2, counts
    assert counts[True][1] <= 2 * max(counts[True][0]) + 1


# ---- the exchange off the critical path (LockstepSchedule.program_overlapped) ------------------------------------------------------
@pytest.mark.parametrize("name", list(CASES))
@pytest.mark.parametrize("mode", [M.REPAM_ANISOTROPIC, M.REPAM_DAMPED_UNIFORM])
def test_overlapped_exchanges_give_the_same_sweep_bit_for_bit(name, mode):
    """the collective of an exchange posted behind the cut-adjacent records and awaited before the first reader of what it ships,
    the interior records of the level in between (they commute with the cut-adjacent ones): the unpartitioned oracle's duals,
    every call; the overlapped program ships the same sets in the same order, and an exchange is split only where the run behind
    it begins with records that read none of it"""
    c = CASES[name]()
    ref = Oracle(_global_of(c)); ref.set_reparametrization(mode)
    sched, parts = _parts_of(c, mode)
    duals = [p.model.dual_data.copy() for p in parts]
    sweeps = [LS.LockstepSweep(torch, p, sched, OracleEngine(p.model, d), torch.from_numpy(d), overlap_exchange=True) for p, d in zip(parts, duals)]
    for n in (1, 2, 3):
        ref.ComputePass(n)
        LS.run_lockstep(sweeps, n)
        _assert_equals_global(c, parts, duals, ref)
    plain, over = sched.program(3), sched.program_overlapped(3)
    assert [s[2] for s in plain if s[0] == "halo"] == [s[2] for s in over if s[0] in ("halo", "halo_begin")]
    assert [x for s in plain if s[0] == "run" for x in s[1]] == [x for s in over if s[0] == "run" for x in s[1]]     # same sub-levels, same order
    for i, s in enumerate(over):
        if s[0] == "halo_begin":
            assert over[i + 1][0] == "run" and over[i + 2][0] == "halo_end" and over[i + 2][2] == s[2]
            shipped = np.zeros(sched.n_vecs, bool); shipped[s[1]] = True
            assert not any(shipped[sched.read[d][sl]].any() for (d, sl) in over[i + 1][1])
    assert sum(1 for s in over if s[0] == "halo_begin") >= 1


WORKER_OVERLAP = r"""
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, {root!r})
from lp_mp_amd import model as M, multi_gpu as MG, lockstep as LS
from tests.mgpu_helpers import OracleEngine
from tests.test_multi_gpu import _general_models
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
gm = _general_models()["c5"]
part_of = MG.graph_partition_model(gm, world)
def factory(part):
    d = part.model.dual_data.copy()
    return torch.from_numpy(d), OracleEngine(part.model, d)
lbs = []
for overlap in (False, True):
    sw = LS.LockstepModel(torch, dist, gm, part_of, M.REPAM_ANISOTROPIC, engine_factory=factory, overlap_exchange=overlap)
    sw.compute_pass(2); sw.compute_pass(1)
    lbs.append(sw.lower_bound())
    np.save(os.path.join({out!r}, f"ov_duals_{{int(overlap)}}_{{rank}}.npy"), sw.dualt.numpy())
    if overlap:
        st = sw.probe_passes(2)
if rank == 0:
    np.save(os.path.join({out!r}, "ov_lb.npy"), np.array(lbs + [st["exchanges_per_pass"]]))
dist.destroy_process_group()
"""


def test_two_process_gloo_run_with_overlapped_exchanges(tmp_path):
    """the LockstepModel DRIVER (engine stand-in) in two gloo processes, exchanges overlapped or not: same duals on every rank, the
    unpartitioned oracle's bound"""
    from tests.test_multi_gpu import _general_models
    script = tmp_path / "ov_worker.py"
    script.write_text(WORKER_OVERLAP.format(root=ROOT, out=str(tmp_path)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                           "--master-addr", "127.0.0.1", "--master-port", "29549", str(script)], env=env, cwd=ROOT, timeout=600)
    for r in range(2):
        assert np.array_equal(np.load(tmp_path / f"ov_duals_0_{r}.npy"), np.load(tmp_path / f"ov_duals_1_{r}.npy"))
    gm = _general_models()["c5"]
    ref = Oracle(gm); ref.set_reparametrization(M.REPAM_ANISOTROPIC); ref.ComputePass(3)
    lb = np.load(tmp_path / "ov_lb.npy")
    assert abs(lb[0] - ref.LowerBound()) <= 1e-12 * max(1.0, abs(ref.LowerBound())) and lb[0] == lb[1] and lb[2] > 0


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["graph", "strips", "c5"])
def test_overlapped_exchanges_on_device_equal_the_oracle(name):
    """real engines, all parts on the one GPU: the split schedules (cut-adjacent records | exchange posted | interior records |
    exchange awaited) leave the oracle's duals"""
    from lp_mp_amd import engine as E
    from tests.test_multi_gpu import _general_models
    mode = M.REPAM_ANISOTROPIC
    if name == "c5":
        gm = _general_models()["c5"]
        sched, parts = LS.lockstep_model(gm, MG.graph_partition_model(gm, 3), 3, mode)
    else:
        c = _graph(1500, 6000, 16, 4, 2) if name == "graph" else _strips(12, 10, 8, 3, "dense", "colour_major", 5)
        gm = _global_of(c)
        sched, parts = _parts_of(c, mode)
    ref = Oracle(gm); ref.set_reparametrization(mode)
    res = {}
    for overlap in (False, True):
        sweeps, tensors = [], []
        for p in parts:
            dual = torch.from_numpy(p.model.dual_data.copy()).to("cuda:0")
            eng = E.Engine(0); eng.set_stream(torch.cuda.current_stream().cuda_stream)
            eng.upload(p.model, dual_dev=dual.data_ptr(), keep=dual, rows_layout=False); eng.set_reparametrization(mode)
            sweeps.append(LS.LockstepSweep(torch, p, sched, eng, dual, overlap_exchange=overlap)); tensors.append(dual)
        try:
            LS.run_lockstep(sweeps, 2); LS.run_lockstep(sweeps, 1); torch.cuda.synchronize()
            res[overlap] = ([t.cpu().numpy() for t in tensors], sum(s.local_lower_bound() for s in sweeps))
        finally:
            for s in sweeps:
                s.close(); s.engine.close()
    ref.ComputePass(3)
    for a, b in zip(res[False][0], res[True][0]):
        assert np.array_equal(a, b)
    assert abs(res[True][1] - ref.LowerBound()) <= 1e-9 * max(1.0, abs(ref.LowerBound()))
