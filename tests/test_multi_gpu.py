"""Partitioned (multi-GPU) sweep.  CPU tests: partition + exchange logic with an oracle-backed engine
stand-in, incl. a world_size-2 gloo run; GPU test: the same schedule on real HIP engines, several parts
on one device.  Everything is checked against the oracle replaying the partition schedule on the
UNPARTITIONED model through the reference's iterator-range ComputePass."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from lp_mp_amd import model as M
from lp_mp_amd import multi_gpu as MG
from lp_mp_amd import synthetic as S
from oracle.binding import Oracle
from tests.mgpu_helpers import OracleEngine, attach_local_lists, gather_global_duals, global_replay

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _strip_setup(H, W, L, world, pairwise, order, seed):
    ei, ej = MG.strip_global_edges(H, W, world, order)
    un, tables, potts = MG.strip_costs(H, W, L, world, pairwise, seed)
    n_vars = world * H * W
    gm = S.mrf_model(n_vars, L, ei, ej, un, tables=tables, potts=potts)
    part_of = np.repeat(np.arange(world), H * W)
    parts = MG.partition_mrf(n_vars, L, ei, ej, part_of, world, un, tables=tables, potts=potts)
    return gm, parts


def _cpu_sweeps(parts, omega_b, every="sweep"):
    sweeps, duals = [], []
    for p in parts:
        d = p.model.dual_data.copy()
        eng = OracleEngine(p.model, d)
        sweeps.append(MG.PartitionedSweep(torch, p, eng, torch.from_numpy(d), M.REPAM_ANISOTROPIC, omega_b, every))
        duals.append(d)
    return sweeps, duals


@pytest.mark.parametrize("pairwise,order,world", [("dense", "colour_major", 2), ("potts", "row_major", 3),
                                                   ("dense", "row_major", 4)])
def test_strip_generator_equals_general_partitioner(pairwise, order, world):
    H, W, L = 4, 5, 3
    gm, parts = _strip_setup(H, W, L, world, pairwise, order, 7)
    for k in range(world):
        q = MG.strip_local_part(H, W, L, pairwise, order, k, world, 7)
        p = parts[k]
        for name in ("f_type", "f_kind", "f_dim0", "m_type", "m_left", "m_right", "rel_fwd", "rel_bwd", "const_data", "dual_data"):
            assert np.array_equal(getattr(p.model, name), getattr(q.model, name)), name
        for name in ("local_to_global", "local_msg_to_global", "out_peer", "out_ghost", "out_key", "in_peer", "in_unary", "in_key"):
            assert np.array_equal(getattr(p, name), getattr(q, name)), name
        assert (p.n_local, p.n_ghost) == (q.n_local, q.n_ghost)


@pytest.mark.parametrize("every", ["pass", "sweep"])
@pytest.mark.parametrize("pairwise,order,world", [("dense", "colour_major", 2), ("potts", "row_major", 3)])
def test_lockstep_parts_equal_oracle_replay_on_global_model(pairwise, order, world, every):
    H, W, L = 5, 6, 4
    gm, parts = _strip_setup(H, W, L, world, pairwise, order, 3)
    attach_local_lists(parts)
    sweeps, duals = _cpu_sweeps(parts, 0.5, every)
    MG.run_lockstep(sweeps, 4)
    MG.run_lockstep(sweeps, 1)
    if every == "pass" and order == "colour_major":
        assert [k for _, *k in sweeps[0].program(4)][0::2] == [["first"], ["mid"], ["mid"], ["last"]]
    o = global_replay(gm, parts, sweeps, [4, 1])
    got = gather_global_duals(gm, parts, duals)
    assert np.array_equal(got, o.duals())
    lb = sum(s.local_lower_bound() for s in sweeps)
    assert abs(lb - o.LowerBound()) <= 1e-9 * max(1.0, abs(lb))
    # dual-bound gap to the unpartitioned sweep after the same number of passes: reported, and bounded
    ref = Oracle(gm)
    ref.set_reparametrization(M.REPAM_ANISOTROPIC)
    lb0 = ref.LowerBound()
    ref.ComputePass(5)
    assert lb > lb0 and lb <= ref.LowerBound() + 0.2 * abs(ref.LowerBound())


def test_parts_generated_from_the_cost_stream_hold_no_host_copy_of_costs_or_duals():
    """the C4 parts (costs generated in HBM) are structure only: at 2 M / 10 M a host copy of the duals alone is 2.8 GB of zeros
    built and concatenated per rank; sizes, offsets and the host analysis do not depend on it"""
    from lp_mp_amd import engine as E
    n, m, L = 300, 900, 4
    part = MG.graph_local_part(n, m, L, 0, 1, seed=1)
    mdl = part.model
    assert mdl.dual_data is None and mdl.const_data is None
    ref = S.counter_graph_model(n, m, L, 1)
    assert int(mdl.dual_sizes().sum()) == ref.dual_data.shape[0] and int(mdl.const_sizes().sum()) == ref.const_data.shape[0]
    assert np.array_equal(mdl.m_left, ref.m_left) and np.array_equal(mdl.m_right, ref.m_right) and np.array_equal(mdl.rel_fwd, ref.rel_fwd)
    pa, pb = E.Plan(mdl), E.Plan(ref)
    for d in (0, 1):
        assert np.array_equal(pa.order(d), pb.order(d)) and np.array_equal(pa.update_order(d), pb.update_order(d))
        oa, ob = pa.omega(d, M.REPAM_ANISOTROPIC), pb.omega(d, M.REPAM_ANISOTROPIC)
        assert np.array_equal(oa[0], ob[0]) and np.array_equal(oa[1], ob[1])


def test_random_graph_general_partition_with_multi_cut_unaries():
    # random sparse graph, random 3-way split: unaries with several cut edges exercise the rounds
    n, m_edges, L, world = 40, 120, 3, 3
    g = S.random_graph_model(n, m_edges, L, seed=11)
    ei = g.m_left[0::2].astype(np.int64)
    ej = g.m_left[1::2].astype(np.int64)
    un = g.dual_data[: n * L]
    tables = g.const_data
    rng = np.random.default_rng(0)
    part_of = rng.integers(0, world, n)
    parts = MG.partition_mrf(n, L, ei, ej, part_of, world, un, tables=tables)
    assert max(np.bincount(p.in_unary).max() if p.in_unary.size else 0 for p in parts) >= 2
    attach_local_lists(parts)
    sweeps, duals = _cpu_sweeps(parts, None)
    lbs = [sum(s.local_lower_bound() for s in sweeps)]
    for _ in range(4):
        MG.run_lockstep(sweeps, 1)
        lbs.append(sum(s.local_lower_bound() for s in sweeps))
    assert all(b >= a - 1e-9 for a, b in zip(lbs, lbs[1:]))          # every step is a valid dual-ascent step
    o = global_replay(g, parts, sweeps, [1, 1, 1, 1])
    assert np.array_equal(gather_global_duals(g, parts, duals), o.duals())
    with pytest.raises(ValueError):
        _cpu_sweeps(parts, 0.9)                                       # weights of a multi-cut unary would exceed 1


def test_general_graph_parts_with_reserved_send_weight_equal_the_replay():
    """GraphSweep's schedule (multi_gpu.BOUNDARY_RESERVE: the main sweeps of a boundary variable keep back part of its
    send weight; boundary step before each directional sweep): still a sequence of iterator-range passes, so the oracle
    replay on the unpartitioned model matches bit for bit and every step ascends (what the reserve buys is measured on
    the C4-shaped graph, tests/gap_probe.py: it is a heuristic, not better on every instance)"""
    n, m_edges, L, world = 60, 200, 4, 3
    g = S.random_graph_model(n, m_edges, L, seed=5)
    ei = g.m_left[0::2].astype(np.int64); ej = g.m_left[1::2].astype(np.int64)
    part_of = MG.graph_partition(n, ei, ej, world)
    res = {}
    for reserve in (0.0, MG.BOUNDARY_RESERVE):
        parts = MG.partition_mrf(n, L, ei, ej, part_of, world, g.dual_data[: n * L], tables=g.const_data)
        attach_local_lists(parts)
        sweeps, duals = [], []
        for p in parts:
            d = p.model.dual_data.copy()
            sweeps.append(MG.PartitionedSweep(torch, p, OracleEngine(p.model, d), torch.from_numpy(d), M.REPAM_ANISOTROPIC, None, "sweep", reserve))
            duals.append(d)
        assert [s[0] for s in sweeps[0].program(1)] == ["boundary", "run", "boundary", "run"]
        lbs = [sum(s.local_lower_bound() for s in sweeps)]
        for _ in range(4):
            MG.run_lockstep(sweeps, 1)
            lbs.append(sum(s.local_lower_bound() for s in sweeps))
        assert all(b >= a - 1e-9 for a, b in zip(lbs, lbs[1:]))
        o = global_replay(g, parts, sweeps, [1, 1, 1, 1])
        assert np.array_equal(gather_global_duals(g, parts, duals), o.duals())
        res[reserve] = lbs[-1]
    print("bound after 4 passes without / with the reserve:", res)


WORKER = r"""
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, {root!r})
from lp_mp_amd import model as M, multi_gpu as MG
from tests.mgpu_helpers import OracleEngine
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
H, W, L = 5, 6, 4
part = MG.strip_local_part(H, W, L, "dense", "colour_major", rank, world, 3)
d = part.model.dual_data.copy()
sw = MG.PartitionedSweep(torch, part, OracleEngine(part.model, d), torch.from_numpy(d), M.REPAM_ANISOTROPIC, 0.5, "pass")
comm = MG.DistComm(dist, torch)
sw.compute_pass(comm, 3)
lb = comm.all_reduce_sum(sw.local_lower_bound())
np.save(os.path.join({out!r}, f"duals_{{rank}}.npy"), d)
if rank == 0:
    np.save(os.path.join({out!r}, "lb.npy"), np.array([lb]))
dist.destroy_process_group()
"""


def test_two_process_gloo_run_equals_lockstep(tmp_path):
    """world_size 2 over torch.distributed (gloo): all_to_all_single exchange of the cut-edge messages."""
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT, out=str(tmp_path)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                           "--master-addr", "127.0.0.1", "--master-port", "29533", str(script)], env=env, cwd=ROOT,
                          timeout=300)
    gm, parts = _strip_setup(5, 6, 4, 2, "dense", "colour_major", 3)
    attach_local_lists(parts)
    sweeps, duals = _cpu_sweeps(parts, 0.5, "pass")
    MG.run_lockstep(sweeps, 3)
    for k in range(2):
        assert np.array_equal(np.load(tmp_path / f"duals_{k}.npy"), duals[k])
    o = global_replay(gm, parts, sweeps, 3)
    assert abs(np.load(tmp_path / "lb.npy")[0] - o.LowerBound()) <= 1e-9 * abs(o.LowerBound())


@pytest.mark.gpu
@pytest.mark.parametrize("every", ["pass", "sweep"])
@pytest.mark.parametrize("pairwise,L,order,world", [("dense", 32, "colour_major", 2), ("potts", 8, "colour_major", 4),
                                                     ("dense", 16, "row_major", 3)])
def test_partitioned_sweep_on_device_equals_oracle_replay(pairwise, L, order, world, every):
    """real HIP engines, all parts on the one GPU of the test box, lock-stepped in process"""
    from lp_mp_amd import engine as E
    H, W = 10, 12
    gm, parts = _strip_setup(H, W, L, world, pairwise, order, 5)
    attach_local_lists(parts)
    dev = torch.device("cuda:0")
    sweeps, tensors, engines = [], [], []
    for p in parts:
        dual = torch.from_numpy(p.model.dual_data.copy()).to(dev)
        eng = E.Engine(0)
        eng.set_stream(torch.cuda.current_stream().cuda_stream)
        eng.upload(p.model, dual_dev=dual.data_ptr(), keep=dual)
        sweeps.append(MG.PartitionedSweep(torch, p, eng, dual, M.REPAM_ANISOTROPIC, 0.5, every))
        tensors.append(dual); engines.append(eng)
    MG.run_lockstep(sweeps, 3)
    torch.cuda.synchronize()
    o = global_replay(gm, parts, sweeps, 3)
    got = gather_global_duals(gm, parts, [t.cpu().numpy() for t in tensors])
    assert np.array_equal(got, o.duals())
    lb = sum(s.local_lower_bound() for s in sweeps)
    assert abs(lb - o.LowerBound()) <= 1e-5 * max(1.0, abs(lb))
    for e in engines:
        e.close()


def test_graph_partition_is_balanced_and_finds_locality():
    # a grid given as an anonymous graph: strips/bands should come out, far fewer cut edges than a random split
    H = W = 40
    a, b = S.grid_edges(H, W)
    part = MG.graph_partition(H * W, a, b, 4)
    sizes = np.bincount(part, minlength=4)
    assert sizes.max() <= np.ceil(H * W / 4 * 1.03) and sizes.min() >= H * W / 4 * 0.9       # balanced within the refinement's slack
    cut = int((part[a] != part[b]).sum())
    rnd = np.random.default_rng(0).integers(0, 4, H * W)
    assert cut < 0.2 * int((rnd[a] != rnd[b]).sum())
    # the refinement never cuts more than the chunked order it starts from, and recovers a good share of a random
    # graph's edges (where the order finds nothing)
    assert cut <= int((MG.graph_partition(H * W, a, b, 4, refine_rounds=0)[a] != MG.graph_partition(H * W, a, b, 4, refine_rounds=0)[b]).sum())
    ei, ej = S.counter_graph_edges(5000, 25000, 3)
    c0 = MG.graph_partition(5000, ei, ej, 8, refine_rounds=0); c1 = MG.graph_partition(5000, ei, ej, 8)
    assert (c1[ei] != c1[ej]).mean() < 0.8 * (c0[ei] != c0[ej]).mean()
    assert np.array_equal(c1, MG.graph_partition(5000, ei, ej, 8))                            # deterministic: every rank gets the same


@pytest.mark.gpu
def test_c4_style_random_graph_partitioned_on_device():
    """BASELINE configs[3] in miniature: random sparse graph, 16 labels, dense tables, split into 4 parts with the
    built-in partitioner (METIS is not available), real HIP engines, oracle replay on the unpartitioned model."""
    from lp_mp_amd import engine as E
    n, m_edges, L, world = 600, 2400, 16, 4
    g = S.random_graph_model(n, m_edges, L, seed=21)
    ei = g.m_left[0::2].astype(np.int64)
    ej = g.m_left[1::2].astype(np.int64)
    part_of = MG.graph_partition(n, ei, ej, world)
    parts = MG.partition_mrf(n, L, ei, ej, part_of, world, g.dual_data[: n * L], tables=g.const_data)
    attach_local_lists(parts)
    dev = torch.device("cuda:0")
    sweeps, tensors, engines = [], [], []
    for p in parts:
        dual = torch.from_numpy(p.model.dual_data.copy()).to(dev)
        eng = E.Engine(0)
        eng.set_stream(torch.cuda.current_stream().cuda_stream)
        eng.upload(p.model, dual_dev=dual.data_ptr(), keep=dual)
        sweeps.append(MG.PartitionedSweep(torch, p, eng, dual, M.REPAM_ANISOTROPIC))
        tensors.append(dual); engines.append(eng)
    lb0 = sum(s.local_lower_bound() for s in sweeps)
    MG.run_lockstep(sweeps, 3)
    torch.cuda.synchronize()
    o = global_replay(g, parts, sweeps, 3)
    assert np.array_equal(gather_global_duals(g, parts, [t.cpu().numpy() for t in tensors]), o.duals())
    lb = sum(s.local_lower_bound() for s in sweeps)
    assert abs(lb - o.LowerBound()) <= 1e-5 * max(1.0, abs(lb))
    ref = Oracle(g)
    ref.set_reparametrization(M.REPAM_ANISOTROPIC)
    ref.ComputePass(3)
    gap = (ref.LowerBound() - lb) / abs(ref.LowerBound())
    print("dual-bound gap to the unpartitioned sweep after 3 passes: %.2f %%" % (100 * gap))
    assert lb > lb0 and gap < 0.5       # dual ascent, and in the neighbourhood of the unpartitioned bound
    for e in engines:
        e.close()


@pytest.mark.parametrize("seed", range(8))
def test_random_partitions_random_graphs_exact_and_ascending(seed):
    """random sparse graphs (dense or Potts edges), random or locality-based partitions into 2-5 parts, both boundary
    placements: the lock-stepped parts equal the oracle's replay on the unpartitioned model bit for bit, and every
    step is a dual-ascent step."""
    rng = np.random.default_rng(7000 + seed)
    n, L, world = int(rng.integers(20, 60)), int(rng.choice([2, 3, 4])), int(rng.integers(2, 6))
    pairwise = "potts" if seed % 2 else "dense"
    g = S.random_graph_model(n, int(rng.integers(n, 3 * n)), L, seed=seed, pairwise=pairwise)
    ei, ej = g.m_left[0::2].astype(np.int64), g.m_left[1::2].astype(np.int64)
    part_of = rng.integers(0, world, n) if seed % 3 else MG.graph_partition(n, ei, ej, world)
    kw = dict(potts=g.const_data) if pairwise == "potts" else dict(tables=g.const_data)
    parts = MG.partition_mrf(n, L, ei, ej, part_of, world, g.dual_data[: n * L], **kw)
    attach_local_lists(parts)
    every = "pass" if seed % 2 else "sweep"
    sweeps, duals = _cpu_sweeps(parts, None, every)
    calls = [1, 3, 2]
    lbs = [sum(s.local_lower_bound() for s in sweeps)]
    for c in calls:
        MG.run_lockstep(sweeps, c)
        lbs.append(sum(s.local_lower_bound() for s in sweeps))
    assert all(b >= a - 1e-9 for a, b in zip(lbs, lbs[1:]))
    o = global_replay(g, parts, sweeps, calls)
    assert np.array_equal(gather_global_duals(g, parts, duals), o.duals())
    assert abs(lbs[-1] - o.LowerBound()) <= 1e-9 * max(1.0, abs(lbs[-1]))


def _general_models():
    rng = np.random.default_rng(31)
    out = {}
    # a) plain MRF with mixed label counts and mixed dense / Potts edges
    b = M.ModelBuilder(2, S.mrf_mtypes())
    dims = rng.choice([2, 3, 5], size=18)
    u = [b.add_vector_factors(0, rng.uniform(0, 1, (1, int(d))))[0] for d in dims]
    for _ in range(40):
        i, j = sorted(rng.choice(18, 2, replace=False))
        if dims[i] == dims[j] and rng.uniform() < 0.4:
            p = b.add_potts_pairwise(1, int(dims[i]), [rng.uniform(0, 1)])[0]
        else:
            p = b.add_dense_pairwise(1, rng.uniform(0, 1, (1, int(dims[i]), int(dims[j]))))[0]
        b.add_messages(0, u[i], p); b.add_messages(1, u[j], p)
        b.add_relations(u[i], p); b.add_relations(p, u[j])
    out["mixed_mrf"] = b.finish()
    # b) multicut-style labeling lists: edge variables, triplet factors
    out["multicut"] = S.multicut_triangle_model(14, 25, seed=3)
    # c) C5 in miniature: Potts grid + triplets + quadruples over binary edge variables, one factor graph
    out["c5"] = S.c5_model(5, 6, 4, 40, 25, 10, seed=2, window=12)
    return out


@pytest.mark.parametrize("name", ["mixed_mrf", "multicut", "c5"])
@pytest.mark.parametrize("world,every", [(2, "sweep"), (3, "pass")])
def test_general_partitioner_any_left_schedule_model(name, world, every):
    """partition_model on models that are not plain MRFs (ragged label counts, labeling-list factors): the parts run
    in lockstep, the oracle replays the same schedule on the unpartitioned model, duals bit-identical"""
    gm = _general_models()[name]
    rng = np.random.default_rng(5)
    part_of = rng.integers(0, world, gm.n_factors)
    parts = MG.partition_model(gm, part_of, world)
    assert sum(p.n_local for p in parts) + sum(int((p.model.f_kind != M.F_VECTOR).sum()) +
               int(((p.model.f_kind == M.F_VECTOR) & (np.arange(p.model.n_factors) >= p.n_local + p.n_ghost)).sum()) for p in parts) == gm.n_factors
    assert sum(p.n_ghost for p in parts) > 0 and sum(p.model.n_messages for p in parts) == gm.n_messages
    attach_local_lists(parts)
    sweeps, duals = _cpu_sweeps(parts, None, every)
    lbs = [sum(s.local_lower_bound() for s in sweeps)]
    ref = Oracle(gm)
    assert abs(lbs[0] - ref.LowerBound()) <= 1e-9 * max(1.0, abs(lbs[0]))
    for _ in range(3):
        MG.run_lockstep(sweeps, 1)
        lbs.append(sum(s.local_lower_bound() for s in sweeps))
    assert all(b >= a - 1e-9 for a, b in zip(lbs, lbs[1:]))
    o = global_replay(gm, parts, sweeps, [1, 1, 1])
    assert np.array_equal(gather_global_duals(gm, parts, duals), o.duals())
    assert abs(lbs[-1] - o.LowerBound()) <= 1e-9 * max(1.0, abs(lbs[-1]))


def test_general_partitioner_rejects_what_it_cannot_split():
    mt = [M.MsgType(0, 0, M.SCHED_LEFT, 0, 0, M.M_MINNORM, 0)]
    b = M.ModelBuilder(1, mt)
    f = b.add_vector_factors(0, np.zeros((3, 2)))
    b.add_messages(0, f[0], f[1]); b.add_messages(0, f[1], f[2])
    with pytest.raises(ValueError):
        MG.partition_model(b.finish(), np.zeros(3, np.int64), 2)
    mt = [M.MsgType(0, 1, M.SCHED_FULL, 0, 1, M.M_UNARY_PAIRWISE, 0), M.MsgType(0, 1, M.SCHED_FULL, 0, 1, M.M_UNARY_PAIRWISE, 1)]
    b = M.ModelBuilder(2, mt)
    u = b.add_vector_factors(0, np.zeros((2, 2)))
    p = b.add_dense_pairwise(1, np.zeros((1, 2, 2)))[0]
    b.add_messages(0, u[0], p); b.add_messages(1, u[1], p)
    with pytest.raises(ValueError):
        MG.partition_model(b.finish(), np.zeros(3, np.int64), 2)


WORKER_GENERAL = r"""
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, {root!r})
from lp_mp_amd import model as M, multi_gpu as MG
from tests.mgpu_helpers import OracleEngine
from tests.test_multi_gpu import _general_models
dist.init_process_group("gloo")
rank = dist.get_rank()
gm = _general_models()["c5"]
part_of = np.random.default_rng(5).integers(0, 2, gm.n_factors)
def factory(m):
    d = m.dual_data.copy()
    return torch.from_numpy(d), OracleEngine(m, d)
sw = MG.ModelSweep(torch, dist, gm, part_of, M.REPAM_ANISOTROPIC, None, "sweep", engine_factory=factory)
sw.compute_pass(3)
lb = sw.lower_bound()
np.save(os.path.join({out!r}, f"gduals_{{rank}}.npy"), sw.dualt.numpy())
if rank == 0:
    np.save(os.path.join({out!r}, "glb.npy"), np.array([lb]))
dist.destroy_process_group()
"""


def test_two_process_gloo_run_of_a_general_model(tmp_path):
    """C5 in miniature over torch.distributed (gloo, world_size 2) through ModelSweep: ragged all-to-all exchange"""
    script = tmp_path / "worker_general.py"
    script.write_text(WORKER_GENERAL.format(root=ROOT, out=str(tmp_path)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                           "--master-addr", "127.0.0.1", "--master-port", "29534", str(script)], env=env, cwd=ROOT,
                          timeout=300)
    gm = _general_models()["c5"]
    parts = MG.partition_model(gm, np.random.default_rng(5).integers(0, 2, gm.n_factors), 2)
    attach_local_lists(parts)
    sweeps, duals = _cpu_sweeps(parts, None, "sweep")
    MG.run_lockstep(sweeps, 3)
    for k in range(2):
        assert np.array_equal(np.load(tmp_path / f"gduals_{k}.npy"), duals[k])
    o = global_replay(gm, parts, sweeps, 3)
    assert abs(np.load(tmp_path / "glb.npy")[0] - o.LowerBound()) <= 1e-9 * max(1.0, abs(o.LowerBound()))


WORKER_C4 = r"""
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, {root!r})
from lp_mp_amd import model as M, multi_gpu as MG
from tests.mgpu_helpers import OracleEngine, materialise_fills
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
from lp_mp_amd import synthetic as S
calls = []
def partitioner():
    calls.append(rank)
    return MG.graph_partition({n}, *S.counter_graph_edges({n}, {m}, 1), world)
part_of = MG.broadcast_partition(torch, dist, {n}, "cpu", partitioner)                # computed on rank 0 only, broadcast
assert calls == ([0] if rank == 0 else [])
part = materialise_fills(MG.graph_local_part({n}, {m}, {L}, rank, world, seed=1, part=part_of))     # this rank's part only
d = part.model.dual_data.copy()
sw = MG.PartitionedSweep(torch, part, OracleEngine(part.model, d), torch.from_numpy(d), M.REPAM_ANISOTROPIC, None, "sweep")
comm = MG.DistComm(dist, torch)
sw.compute_pass(comm, {passes})
lb = comm.all_reduce_sum(sw.local_lower_bound())
np.save(os.path.join({out!r}, f"c4_duals_{{rank}}.npy"), d)
if rank == 0:
    np.save(os.path.join({out!r}, "c4_lb.npy"), np.array([lb]))
dist.destroy_process_group()
"""


def test_two_process_gloo_run_of_the_c4_graph(tmp_path):
    """BASELINE.json configs[3] in the shape bench.py --workload c4 runs it, at 20 000 nodes / 100 000 edges: every rank
    builds ONLY its own part (counter-generated structure, partition by graph_partition, costs from fill descriptors),
    cut messages travel over torch.distributed (gloo, world_size 2).  The result equals the in-process lock-step run
    of the partition of the global model, which equals the oracle's replay of the schedule on the unpartitioned
    model, bit for bit; the dual bound stays within a few percent of the unpartitioned sweep's."""
    n, m, L, passes = 20000, 100000, 4, 2
    script = tmp_path / "worker_c4.py"
    script.write_text(WORKER_C4.format(root=ROOT, out=str(tmp_path), n=n, m=m, L=L, passes=passes))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                           "--master-addr", "127.0.0.1", "--master-port", "29535", str(script)], env=env, cwd=ROOT, timeout=600)
    gm = S.counter_graph_model(n, m, L, 1)
    ei, ej = S.counter_graph_edges(n, m, 1)
    part_of = MG.graph_partition(n, ei, ej, 2)
    assert 0.2 < float((part_of[ei] != part_of[ej]).mean()) < 0.4 and abs(np.bincount(part_of)[0] / n - 0.5) < 0.02
    parts = MG.partition_mrf(n, L, ei, ej, part_of, 2, gm.dual_data[: n * L], tables=gm.const_data)
    attach_local_lists(parts)
    sweeps, duals = _cpu_sweeps(parts, None, "sweep")
    MG.run_lockstep(sweeps, passes)
    for k in range(2):
        assert np.array_equal(np.load(tmp_path / f"c4_duals_{k}.npy"), duals[k])
    o = global_replay(gm, parts, sweeps, passes)
    assert np.array_equal(gather_global_duals(gm, parts, duals), o.duals())
    lb = np.load(tmp_path / "c4_lb.npy")[0]
    assert abs(lb - o.LowerBound()) <= 1e-9 * abs(o.LowerBound())
    ref = Oracle(gm); ref.set_reparametrization(M.REPAM_ANISOTROPIC); ref.ComputePass(passes)
    assert 0 <= (ref.LowerBound() - lb) / abs(ref.LowerBound()) < 0.08


@pytest.mark.gpu
@pytest.mark.parametrize("name,world,every", [("c5", 3, "pass"), ("multicut", 4, "sweep"), ("mixed_mrf", 2, "pass")])
def test_general_partitioner_on_device(name, world, every):
    """BASELINE configs[4] in miniature across several parts (grid + labeling-list factors in one factor graph), and the
    other general models: real HIP engines, all parts on the one GPU, oracle replay on the unpartitioned model"""
    from lp_mp_amd import engine as E
    gm = _general_models()[name]
    part_of = np.random.default_rng(9).integers(0, world, gm.n_factors)
    parts = MG.partition_model(gm, part_of, world)
    attach_local_lists(parts)
    dev = torch.device("cuda:0")
    sweeps, tensors, engines = [], [], []
    for p in parts:
        dual = torch.from_numpy(p.model.dual_data.copy()).to(dev)
        eng = E.Engine(0)
        eng.set_stream(torch.cuda.current_stream().cuda_stream)
        eng.upload(p.model, dual_dev=dual.data_ptr(), keep=dual)
        sweeps.append(MG.PartitionedSweep(torch, p, eng, dual, M.REPAM_ANISOTROPIC, None, every))
        tensors.append(dual); engines.append(eng)
    lbs = [sum(s.local_lower_bound() for s in sweeps)]
    for _ in range(3):
        MG.run_lockstep(sweeps, 1)
        lbs.append(sum(s.local_lower_bound() for s in sweeps))
    torch.cuda.synchronize()
    o = global_replay(gm, parts, sweeps, [1, 1, 1])
    assert np.array_equal(gather_global_duals(gm, parts, [t.cpu().numpy() for t in tensors]), o.duals())
    assert abs(lbs[-1] - o.LowerBound()) <= 1e-5 * max(1.0, abs(lbs[-1]))
    assert all(b >= a - 1e-9 for a, b in zip(lbs, lbs[1:]))
    for e in engines:
        e.close()


def test_graph_partition_of_a_general_model_is_balanced_and_local():
    gm = S.c5_model(12, 12, 4, 600, 300, 150, seed=3, window=24)
    world = 4
    part = MG.graph_partition_model(gm, world)
    is_right = np.zeros(gm.n_factors, bool); is_right[gm.m_right] = True
    sizes = np.bincount(part[~is_right], minlength=world)
    assert sizes.max() <= np.ceil(sizes.sum() / world * 1.03) and sizes.min() >= sizes.sum() / world * 0.85
    parts = MG.partition_model(gm, part, world)
    cut = sum(p.n_ghost for p in parts)
    rnd = np.random.default_rng(0).integers(0, world, gm.n_factors)
    cut_rnd = sum(p.n_ghost for p in MG.partition_model(gm, rnd, world))
    assert 0 < cut < 0.35 * cut_rnd


@pytest.mark.parametrize("seed", range(12))
def test_general_partitioner_on_randomised_models(seed):
    """randomised MRFs of the GPU parity tests (odd orders, duplicate messages, every label count) and multicut /
    C5-style models, random partitions into 2..4 parts, both boundary cadences: lock-step parts == oracle replay"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("fuzz_models", os.path.join(ROOT, "tests", "test_fuzz_gpu.py"))
    F = importlib.util.module_from_spec(spec); spec.loader.exec_module(F)
    rng = np.random.default_rng(41000 + seed)
    kind = seed % 4
    if kind == 0:
        gm = F.random_mrf(rng)
    elif kind == 1:
        gm = F.random_mrf_any_labels(rng)
    elif kind == 2:
        gm = S.multicut_triangle_model(int(rng.integers(8, 16)), int(rng.integers(10, 30)), seed=seed)
    else:
        gm = S.c5_model(4, 5, 3, 30, int(rng.integers(8, 20)), int(rng.integers(4, 10)), seed=seed, window=int(rng.choice([8, 30])))
    world = int(rng.integers(2, 5))
    every = "pass" if rng.uniform() < 0.5 else "sweep"
    parts = MG.partition_model(gm, rng.integers(0, world, gm.n_factors), world)
    attach_local_lists(parts)
    sweeps, duals = _cpu_sweeps(parts, None, every)
    lbs = [sum(s.local_lower_bound() for s in sweeps)]
    for _ in range(2):
        MG.run_lockstep(sweeps, 1)
        lbs.append(sum(s.local_lower_bound() for s in sweeps))
    MG.run_lockstep(sweeps, 2)
    lbs.append(sum(s.local_lower_bound() for s in sweeps))
    assert all(b >= a - 1e-9 for a, b in zip(lbs, lbs[1:])), (seed, lbs)
    o = global_replay(gm, parts, sweeps, [1, 1, 2])
    assert np.array_equal(gather_global_duals(gm, parts, duals), o.duals()), seed


@pytest.mark.gpu
@pytest.mark.parametrize("pairwise,L,order,parts,every", [("dense", 16, "colour_major", 2, "pass"), ("potts", 8, "row_major", 3, "sweep"),
                                                           ("dense", 32, "colour_major", 3, "sweep")])
def test_cpp_rccl_driver_equals_the_python_partitioned_sweep(tmp_path, pairwise, L, order, parts, every):
    """tools/mgpu_rccl_driver.cpp (C++ host: lpmp_multi_gpu.hxx on the C ABI, exchange = ncclSend / ncclRecv in one group,
    bound = ncclAllReduce) at world 1 with several parts on the one GPU — their cut messages travel through RCCL to self —
    against lp_mp_amd/multi_gpu.py's lock-stepped PartitionedSweep on the same strips: duals bit-identical part by part"""
    from lp_mp_amd import build as B, engine as E
    H, W, passes = 10, 12, 3
    if not B.have_rccl():
        pytest.skip("no <rccl/rccl.h> on this box")
    exe = B.build_mgpu_driver()
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    out = subprocess.check_output([exe, "--H", str(H), "--W", str(W), "--L", str(L), "--pairwise", pairwise, "--order", order,
                                   "--passes", str(passes), "--parts-per-rank", str(parts), "--boundary", every,
                                   "--out", str(tmp_path / "duals")], text=True, env=env, timeout=600)
    import json
    line = json.loads(out.strip().splitlines()[-1])
    dev = torch.device("cuda:0")
    sweeps, tensors, engines = [], [], []
    for k in range(parts):
        p = MG.strip_local_part(H, W, L, pairwise, order, k, parts, 1)
        dual = torch.from_numpy(p.model.dual_data.copy()).to(dev)
        eng = E.Engine(0)
        eng.set_stream(torch.cuda.current_stream().cuda_stream)
        eng.upload(p.model, dual_dev=dual.data_ptr(), keep=dual)
        eng.set_reparametrization(M.REPAM_ANISOTROPIC)
        sweeps.append(MG.PartitionedSweep(torch, p, eng, dual, M.REPAM_ANISOTROPIC, None, every))
        tensors.append(dual); engines.append(eng)
    lb0 = sum(s.local_lower_bound() for s in sweeps)
    MG.run_lockstep(sweeps, passes)
    torch.cuda.synchronize()
    lb1 = sum(s.local_lower_bound() for s in sweeps)
    for k in range(parts):
        got = np.fromfile(tmp_path / f"duals.{k}.bin", dtype=np.float64)
        assert np.array_equal(got, tensors[k].cpu().numpy()), k
    assert abs(line["lower_bound_before"] - lb0) <= 1e-9 * abs(lb0) and abs(line["lower_bound_after"] - lb1) <= 1e-9 * abs(lb1)
    assert line["lower_bound_after"] > line["lower_bound_before"]
    for e in engines:
        e.close()


def _free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def test_nccl_unique_id_is_handed_out_over_tcp_not_through_a_file(tmp_path):
    """rccl_world::hand_out_id (lpmp_multi_gpu.hxx): rank 0 serves the ncclUniqueId on MASTER_ADDR : port to exactly the ranks
    of this launch — three rank processes here, no GPU and no RCCL call involved; a connection with a wrong greeting (another
    program, a rank of another launch) gets nothing and does not use up a slot; a rank without a rank 0 gives up after its
    timeout instead of waiting for ever; nothing is left on disk that a later launch could mistake for its own id"""
    import socket
    import struct
    import time
    from lp_mp_amd import build as B
    if not B.have_rccl():
        pytest.skip("no <rccl/rccl.h> on this box")
    B.build()
    exe = str(tmp_path / "test_id_handout")
    subprocess.check_call([B.hipcc(), "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "lp_mp_amd", "include"), "-o", exe,
                           os.path.join(ROOT, "tests", "cpp", "test_id_handout.cpp"), "-L", B.CSRC, "-llpmp_engine", "-lrccl", "-Wl,-rpath," + B.CSRC])
    port, world = _free_port(), 3
    procs = [subprocess.Popen([exe, str(r), str(world), str(port), "60"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in (1, 0)]
    # an intruder: connects, sends a greeting of the right length with the wrong magic, must receive nothing
    deadline = time.time() + 30
    while True:
        try:
            sk = socket.create_connection(("127.0.0.1", port), timeout=2)
            break
        except OSError:
            assert time.time() < deadline
            time.sleep(0.05)
    sk.sendall(struct.pack("<Qii", 0x1234, world, 2))
    sk.settimeout(5)
    try:
        assert sk.recv(256) == b""
    except socket.timeout:
        pass
    sk.close()
    procs.append(subprocess.Popen([exe, "2", str(world), str(port), "60"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=120) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    ids = {o[0].strip() for o in outs}
    assert len(ids) == 1 and len(next(iter(ids))) == 256 and next(iter(ids)) != "00" * 128
    # no rank 0 at all: bounded wait
    t0 = time.time()
    r = subprocess.run([exe, "1", "2", str(_free_port()), "2"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 1 and "no rank 0" in r.stderr and time.time() - t0 < 30
    src = open(os.path.join(ROOT, "lp_mp_amd", "include", "lpmp_multi_gpu.hxx")).read()
    assert "fopen" not in src and "/tmp/" not in src


def test_model_dump_holds_the_arrays_of_lpmp_model_h_in_order(tmp_path):
    """FlatModel.dump (what the C++ hosts read, model_file in lpmp_lockstep.hxx): header counts, then every array of lpmp_model.h in
    its order and type — parsed back here field by field"""
    gm = _general_models()["c5"]
    path = tmp_path / "m.bin"
    gm.dump(str(path))
    raw = path.read_bytes()
    h = np.frombuffer(raw, np.int64, 11)
    assert h[0] == 0x4C504D504D4F444C and list(h[1:9]) == [gm.n_ftypes, len(gm.mtypes), gm.tab_nleft.shape[0], gm.tab_data.shape[0], gm.n_factors, gm.n_messages,
                                                              gm.rel_fwd.shape[0], gm.rel_bwd.shape[0]]
    at = 88
    assert np.frombuffer(raw, np.float64, 1, at)[0] == gm.constant
    at += 8
    def take(dtype, n):
        nonlocal at
        a = np.frombuffer(raw, dtype, n, at); at += a.nbytes
        return a
    assert np.array_equal(take(np.uint8, gm.n_ftypes), gm.ftype_computes_primal)
    mt = take(np.int32, 8 * len(gm.mtypes)).reshape(-1, 8)
    assert [tuple(r) for r in mt] == [(t.left_ftype, t.right_ftype, t.schedule, t.n_left, t.n_right, t.kind, t.param, t.flags) for t in gm.mtypes]
    assert np.array_equal(take(np.int64, h[3] + 1), gm.tab_off) and np.array_equal(take(np.int32, h[4]), gm.tab_data) and np.array_equal(take(np.int32, h[3]), gm.tab_nleft)
    for name, dt in (("f_type", np.int32), ("f_kind", np.uint8), ("f_flags", np.uint8), ("f_dim0", np.int32), ("f_dim1", np.int32)):
        assert np.array_equal(take(dt, gm.n_factors), getattr(gm, name)), name
    assert np.array_equal(take(np.float64, h[9]), gm.const_data) and np.array_equal(take(np.float64, h[10]), gm.dual_data)
    for name in ("m_type", "m_left", "m_right"):
        assert np.array_equal(take(np.int32, gm.n_messages), getattr(gm, name)), name
    assert np.array_equal(take(np.int32, 2 * h[7]).reshape(-1, 2), gm.rel_fwd) and np.array_equal(take(np.int32, 2 * h[8]).reshape(-1, 2), gm.rel_bwd)
    assert at == len(raw)


def test_cpp_hosts_issue_sends_and_receives_in_one_global_order(tmp_path):
    """tests/cpp/test_transfer_order.cpp: the (source part, destination part) order every C++ exchange uses (for_each_transfer in
    lpmp_multi_gpu.hxx; boundary steps and lock step) pairs the k-th send of rank a to rank b with the k-th receive of b from a,
    for 1..8 ranks with 1..4 parts each — checked on the host, since two RCCL ranks cannot share the test box's GPU"""
    from lp_mp_amd import build as B
    if not B.have_rccl():
        pytest.skip("no <rccl/rccl.h> on this box")
    B.build()
    exe = str(tmp_path / "test_transfer_order")
    subprocess.check_call([B.hipcc(), "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "lp_mp_amd", "include"), "-o", exe,
                           os.path.join(ROOT, "tests", "cpp", "test_transfer_order.cpp"), "-L", B.CSRC, "-llpmp_engine", "-lrccl", "-Wl,-rpath," + B.CSRC])
    out = subprocess.check_output([exe], text=True, timeout=120)
    assert "pair up" in out, out
    # both exchanges go through it
    for f in ("lpmp_multi_gpu.hxx", "lpmp_lockstep.hxx"):
        assert "for_each_transfer(n_parts, w.rank, w.parts_per_rank" in open(os.path.join(ROOT, "lp_mp_amd", "include", f)).read(), f


@pytest.mark.gpu
def test_cpp_rccl_driver_two_processes_on_one_gpu_fail_or_agree(tmp_path):
    """two rank PROCESSES of the C++ RCCL driver (the multi-process path: id hand-out over TCP, ncclCommInitRank at world 2).
    On the 1-GPU test box RCCL refuses two ranks on one device — the run must then end with an error from both ranks within
    its timeout (no hang on a stale id); on a box with two GPUs it must equal the Python partitioned sweep like the
    single-process run above"""
    from lp_mp_amd import build as B
    if not B.have_rccl():
        pytest.skip("no <rccl/rccl.h> on this box")
    exe = B.build_mgpu_driver()
    port = _free_port()
    H, W, L, passes = 10, 12, 16, 3
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), NCCL_DEBUG="WARN")
        procs.append(subprocess.Popen([exe, "--H", str(H), "--W", str(W), "--L", str(L), "--passes", str(passes), "--out", str(tmp_path / "duals")],
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env))
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=300))
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail("the two-process RCCL driver hung")
    if torch.cuda.device_count() < 2:
        assert all(p.returncode != 0 for p in procs), outs      # RCCL: two ranks on one device
        return
    assert all(p.returncode == 0 for p in procs), outs
    import json
    line = json.loads(outs[0][0].strip().splitlines()[-1])
    assert line["lower_bound_after"] > line["lower_bound_before"]


@pytest.mark.gpu
@pytest.mark.parametrize("pairwise,L,H,W,parts,ghost,chunk", [("dense", 16, 12, 10, 2, 6, 0), ("dense", 32, 12, 9, 3, 6, 1), ("potts", 8, 16, 12, 3, 8, 2),
                                                               ("dense", 8, 8, 7, 4, 8, 0)])
def test_cpp_rccl_driver_overlap_schedule_equals_the_python_overlap_sweep_and_the_oracle(tmp_path, pairwise, L, H, W, parts, ghost, chunk):
    """tools/mgpu_rccl_driver.cpp --schedule overlap (lp_mp_amd/include/lpmp_overlap.hxx: windows with ghost rows, plain
    lpmp_compute_pass calls, ghost rows refreshed by ncclSend / ncclRecv of contiguous ranges of the packed duals) at world 1 with
    several parts on the one GPU — through RCCL to self — against lp_mp_amd/overlap.py on the same windows: every part's WHOLE dual
    array bit-identical (ghost rows included: both end a call with an exchange), and the owned rows against the oracle on the
    unpartitioned grid"""
    from lp_mp_amd import build as B, engine as E, overlap as OV
    from oracle.binding import Oracle
    if not B.have_rccl():
        pytest.skip("no <rccl/rccl.h> on this box")
    passes = 5
    exe = B.build_mgpu_driver()
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    out = subprocess.check_output([exe, "--H", str(H), "--W", str(W), "--L", str(L), "--pairwise", pairwise, "--passes", str(passes),
                                   "--parts-per-rank", str(parts), "--schedule", "overlap", "--ghost-rows", str(ghost), "--chunk", str(chunk),
                                   "--out", str(tmp_path / "duals")], text=True, env=env, timeout=600)
    import json
    line = json.loads(out.strip().splitlines()[-1])
    assert line["schedule"] == "overlap" and line["passes_between_exchanges"] == (chunk or (ghost - 2) // 2)
    dev = torch.device("cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    sweeps, tensors = [], []
    for k in range(parts):
        p = OV.strip_window_part(H, W, L, pairwise, k, parts, ghost, 1)
        m = p.model
        const = torch.empty(max(int(m.const_sizes().sum()), 2), dtype=torch.float64, device=dev)
        dual = torch.zeros(int(m.dual_sizes().sum()), dtype=torch.float64, device=dev)
        if m.const_data is not None and p.const_fill is None and m.const_data.size:
            const[: m.const_data.shape[0]] = torch.from_numpy(m.const_data).to(dev)
        MG.fill_device_costs(torch, E, p, const, dual, stream)
        eng = E.Engine(0); eng.set_stream(stream)
        eng.upload(m, const_dev=const.data_ptr(), dual_dev=dual.data_ptr(), keep=(const, dual))
        eng.set_reparametrization(M.REPAM_ANISOTROPIC)
        sweeps.append(OV.OverlapSweep(torch, p, eng, dual, chunk or None)); tensors.append(dual)
    try:
        lb0 = sum(s.local_lower_bound() for s in sweeps)
        OV.run_overlapped(sweeps, passes)
        torch.cuda.synchronize()
        lb1 = sum(s.local_lower_bound() for s in sweeps)
        for k in range(parts):
            got = np.fromfile(tmp_path / f"duals.{k}.bin", dtype=np.float64)
            assert np.array_equal(got, tensors[k].cpu().numpy()), k
        assert abs(line["lower_bound_before"] - lb0) <= 1e-9 * abs(lb0) and abs(line["lower_bound_after"] - lb1) <= 1e-9 * abs(lb1)
        gm = S.grid_model(parts * H, W, L, pairwise=pairwise, order="colour_major", seed=1)
        o = Oracle(gm); o.set_reparametrization(M.REPAM_ANISOTROPIC); o.ComputePass(passes)
        assert abs(line["lower_bound_after"] - o.LowerBound()) <= 1e-9 * abs(o.LowerBound())
    finally:
        for s in sweeps:
            s.engine.close()


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["strips dense colour-major", "strips dense row-major", "strips potts", "random graph", "random graph, colour-major order, partitioner"])
def test_cpp_rccl_driver_lockstep_schedule_equals_the_python_lockstep_sweep_and_the_oracle(tmp_path, case):
    """tools/mgpu_rccl_driver.cpp --schedule lockstep (lp_mp_amd/include/lpmp_lockstep.hxx: the global level structure from
    lpmp_plan_* on the structure of the whole model, runs of sub-levels as lpmp_schedule_create[_fused], halos through lpmp_halo_*
    and ncclSend / ncclRecv) at world 1 with several parts on the one GPU, against lp_mp_amd/lockstep.py on the same parts: every
    part's whole dual array bit-identical, and the bound against the oracle on the unpartitioned model"""
    import json
    from lp_mp_amd import build as B, engine as E, lockstep as LS
    from oracle.binding import Oracle
    if not B.have_rccl():
        pytest.skip("no <rccl/rccl.h> on this box")
    passes = 3
    exe = B.build_mgpu_driver()
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    if case.startswith("random graph"):
        n, m, L, parts, pairwise = 900, 3600, 16, 4, "dense"
        args = ["--graph", str(n), str(m), "--L", str(L)]
        var_rank = None
        part_of = (np.arange(n) * parts) // n
        if "colour-major" in case:                     # what bench.py --workload c4 runs: order and partition from the Python side, as files
            from lp_mp_amd import ordering as O
            var_rank = O.colour_major_order(n, *S.counter_graph_edges(n, m, 1), seed=1)
            part_of = MG.graph_partition(n, *S.counter_graph_edges(n, m, 1, var_rank), parts)
            var_rank.astype(np.int64).tofile(tmp_path / "order.bin"); part_of.astype(np.int64).tofile(tmp_path / "part.bin")
            args += ["--order-file", str(tmp_path / "order.bin"), "--part-file", str(tmp_path / "part.bin")]
        ei, ej = S.counter_graph_edges(n, m, 1, var_rank)
        gm = S.counter_graph_model(n, m, L, 1, rank=var_rank)
    else:
        H, W, parts = 8, 7, 3
        L, pairwise, order = {"strips dense colour-major": (16, "dense", "colour_major"), "strips dense row-major": (5, "dense", "row_major"),
                              "strips potts": (8, "potts", "colour_major")}[case]
        args = ["--H", str(H), "--W", str(W), "--L", str(L), "--pairwise", pairwise, "--order", order]
        ei, ej = MG.strip_global_edges(H, W, parts, order)
        n = parts * H * W
        part_of = np.repeat(np.arange(parts), H * W)
        un, tables, potts = MG.strip_costs(H, W, L, parts, pairwise, 1)
        gm = S.mrf_model(n, L, ei, ej, un, tables=tables, potts=potts)
    out = subprocess.check_output([exe] + args + ["--passes", str(passes), "--parts-per-rank", str(parts), "--schedule", "lockstep",
                                                  "--out", str(tmp_path / "duals")], text=True, env=env, timeout=600)
    line = json.loads(out.strip().splitlines()[-1])
    assert line["schedule"] == "lockstep" and line["parts"] == parts
    sched, lparts = LS.lockstep_mrf(n, L, ei, ej, part_of, parts, M.REPAM_ANISOTROPIC, stream_seed=1, pairwise=pairwise,
                                    unaries=None if pairwise == "dense" else np.zeros(n * L), potts=None if pairwise == "dense" else np.zeros(ei.shape[0]))
    assert list(sched.n_levels) == line["levels"]
    assert abs(sum(1 for s in sched.program(passes) if s[0] == "halo") / passes - line["exchanges_per_pass"]) < 1e-3
    dev = torch.device("cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    sweeps, tensors = [], []
    for p in lparts:
        mdl = p.model
        const = torch.empty(max(int(mdl.const_sizes().sum()), 2), dtype=torch.float64, device=dev)
        dual = torch.zeros(int(mdl.dual_sizes().sum()), dtype=torch.float64, device=dev)
        MG.fill_device_costs(torch, E, p, const, dual, stream)
        eng = E.Engine(0); eng.set_stream(stream)
        eng.upload(mdl, const_dev=const.data_ptr(), dual_dev=dual.data_ptr(), keep=(const, dual)); eng.set_reparametrization(M.REPAM_ANISOTROPIC)
        sweeps.append(LS.LockstepSweep(torch, p, sched, eng, dual)); tensors.append(dual)
    try:
        lb0 = sum(s.local_lower_bound() for s in sweeps)
        LS.run_lockstep(sweeps, passes); torch.cuda.synchronize()
        lb1 = sum(s.local_lower_bound() for s in sweeps)
        for k in range(parts):
            got = np.fromfile(tmp_path / f"duals.{k}.bin", dtype=np.float64)
            assert np.array_equal(got, tensors[k].cpu().numpy()), k
        assert abs(line["lower_bound_before"] - lb0) <= 1e-9 * abs(lb0) and abs(line["lower_bound_after"] - lb1) <= 1e-9 * abs(lb1)
        o = Oracle(gm); o.set_reparametrization(M.REPAM_ANISOTROPIC); o.ComputePass(passes)
        assert abs(line["lower_bound_after"] - o.LowerBound()) <= 1e-9 * abs(o.LowerBound())
    finally:
        for s in sweeps:
            s.close(); s.engine.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name,parts", [("mixed_mrf", 3), ("multicut", 3), ("c5", 2), ("c5", 4)])
def test_cpp_rccl_driver_lockstep_of_general_models_equals_the_python_sweep_and_the_oracle(tmp_path, name, parts):
    """tools/mgpu_rccl_driver.cpp --schedule lockstep --model-file: any `left`-schedule model (ragged label counts, labeling-list
    factors whose whole dual is one exchange unit) read from FlatModel.dump with its costs, random partition of the variables —
    every part's whole dual array equals lockstep.lockstep_model's on the same partition, the bound the oracle's"""
    import json
    from lp_mp_amd import build as B, engine as E, lockstep as LS
    from oracle.binding import Oracle
    if not B.have_rccl():
        pytest.skip("no <rccl/rccl.h> on this box")
    passes = 3
    gm = _general_models()[name]
    rng = np.random.default_rng(17 + parts)
    is_right = np.zeros(gm.n_factors, bool); is_right[gm.m_right] = True
    part_of = rng.integers(0, parts, gm.n_factors)
    part_of[np.nonzero(~is_right)[0][:parts]] = np.arange(parts)
    gm.dump(str(tmp_path / "model.bin")); part_of.astype(np.int64).tofile(tmp_path / "part.bin")
    exe = B.build_mgpu_driver()
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    out = subprocess.check_output([exe, "--schedule", "lockstep", "--model-file", str(tmp_path / "model.bin"), "--part-file", str(tmp_path / "part.bin"),
                                   "--passes", str(passes), "--parts-per-rank", str(parts), "--out", str(tmp_path / "duals")], text=True, env=env, timeout=600)
    line = json.loads(out.strip().splitlines()[-1])
    assert line["schedule"] == "lockstep" and line["parts"] == parts and line["factors"] == gm.n_factors
    sched, lparts = LS.lockstep_model(gm, part_of, parts, M.REPAM_ANISOTROPIC)
    assert list(sched.n_levels) == line["levels"]
    assert abs(sum(1 for s in sched.program(passes) if s[0] == "halo") / passes - line["exchanges_per_pass"]) < 1e-3
    dev = torch.device("cuda:0")
    sweeps, tensors = [], []
    for p in lparts:
        dual = torch.from_numpy(p.model.dual_data.copy()).to(dev)
        eng = E.Engine(0); eng.set_stream(torch.cuda.current_stream().cuda_stream)
        eng.upload(p.model, dual_dev=dual.data_ptr(), keep=dual); eng.set_reparametrization(M.REPAM_ANISOTROPIC)
        sweeps.append(LS.LockstepSweep(torch, p, sched, eng, dual)); tensors.append(dual)
    try:
        lb0 = sum(s.local_lower_bound() for s in sweeps)
        LS.run_lockstep(sweeps, passes); torch.cuda.synchronize()
        lb1 = sum(s.local_lower_bound() for s in sweeps)
        for k in range(parts):
            got = np.fromfile(tmp_path / f"duals.{k}.bin", dtype=np.float64)
            assert np.array_equal(got, tensors[k].cpu().numpy()), k
        assert abs(line["lower_bound_before"] - lb0) <= 1e-9 * max(1.0, abs(lb0)) and abs(line["lower_bound_after"] - lb1) <= 1e-9 * max(1.0, abs(lb1))
        o = Oracle(gm); o.set_reparametrization(M.REPAM_ANISOTROPIC); o.ComputePass(passes)
        assert abs(line["lower_bound_after"] - o.LowerBound()) <= 1e-9 * max(1.0, abs(o.LowerBound()))
    finally:
        for s in sweeps:
            s.close(); s.engine.close()
