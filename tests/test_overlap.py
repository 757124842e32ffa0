"""Overlapped strips (lp_mp_amd/overlap.py): every rank holds its strip of a big 2-colour grid plus g ghost rows on either
side, runs n = g / 2 - 1 plain passes without any exchange, then the owners refresh the ghost rows — the owned rows are the
UNPARTITIONED sweep's (the oracle on the whole grid), bit for bit, with one exchange per n passes.
CPU: oracle-backed engine stand-ins in one process and over torch.distributed (gloo); GPU: real engines, the parts on the
one device of the test box, incl. the joined-pass chain launch forced onto small windows."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from lp_mp_amd import model as M
from lp_mp_amd import overlap as OV
from lp_mp_amd import synthetic as S
from oracle.binding import Oracle
from tests.mgpu_helpers import OracleEngine, materialise_fills

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("GH,W", [(6, 5), (8, 8), (10, 7), (4, 2), (12, 3)])
def test_closed_form_enumeration_is_the_grid_models(GH, W):
    """a rank never builds the global grid: variable and edge numbers of its window rows come from closed forms"""
    v = S.grid_variable_order(GH, W, "colour_major")
    rr, cc = np.meshgrid(np.arange(GH), np.arange(W), indexing="ij")
    assert np.array_equal(OV.global_var_index(rr, cc, GH, W), v)
    a, b = S.grid_edges(GH, W)
    ar, ac = np.divmod(a, W)
    assert np.array_equal(OV.global_edge_index(ar, ac, b - a == W, GH, W), np.arange(a.shape[0]))
    upd, byt = OV.grid_pass_counts(GH, W, 4)
    assert upd == 4 * a.shape[0]


def _global(H, W, L, world, pairwise, seed):
    gm = S.grid_model(world * H, W, L, pairwise=pairwise, order="colour_major", seed=seed)
    n = world * H * W
    costs = {"unaries": gm.dual_data[: n * L], ("tables" if pairwise == "dense" else "potts"): gm.const_data}
    return gm, costs


def _assert_owned_equal_global(gm, parts, duals, ref):
    n = parts[0].world * parts[0].H * parts[0].W
    gd, goff = ref.duals(), gm.dual_offsets()
    seen = np.zeros(gm.n_factors, np.int64)
    for p, d in zip(parts, duals):
        lo = p.model.dual_offsets()
        gl = np.concatenate([p.vars_global, n + p.edges_global])
        for fl in np.nonzero(p.owned)[0]:
            g = int(gl[fl]); seen[g] += 1
            assert np.array_equal(d[lo[fl]:lo[fl + 1]], gd[goff[g]:goff[g + 1]]), (p.rank, int(fl))
    assert np.all(seen == 1)                       # every factor of the grid has exactly one owner


def _cpu_sweeps(parts, mode, chunk=None):
    duals = [p.model.dual_data.copy() for p in parts]
    sweeps = []
    for p, d in zip(parts, duals):
        e = OracleEngine(p.model, d); e.set_reparametrization(mode)
        sweeps.append(OV.OverlapSweep(torch, p, e, torch.from_numpy(d), chunk))
    return sweeps, duals


@pytest.mark.parametrize("H,W,L,world,g,pairwise", [(8, 6, 3, 3, 4, "dense"), (8, 7, 3, 4, 8, "dense"), (8, 6, 4, 2, 6, "potts"),
                                                    (10, 5, 2, 3, 10, "dense"), (6, 6, 3, 5, 6, "dense")])
@pytest.mark.parametrize("mode", [M.REPAM_ANISOTROPIC, M.REPAM_DAMPED_UNIFORM, M.REPAM_UNIFORM, M.REPAM_ANISOTROPIC2])
def test_overlapped_strips_run_the_unpartitioned_sweep(H, W, L, world, g, pairwise, mode):
    gm, costs = _global(H, W, L, world, pairwise, seed=5)
    ref = Oracle(gm); ref.set_reparametrization(mode)
    parts = [OV.strip_window_part(H, W, L, pairwise, r, world, g, costs=costs) for r in range(world)]
    sweeps, duals = _cpu_sweeps(parts, mode)
    assert sweeps[0].chunk == (g - 2) // 2
    for n in (1, 3, 2):                             # separate calls: the ghost rows are fresh when a call returns
        ref.ComputePass(n)
        OV.run_overlapped(sweeps, n)
        _assert_owned_equal_global(gm, parts, duals, ref)
        lb = sum(s.local_lower_bound() for s in sweeps)
        assert abs(lb - ref.LowerBound()) <= 1e-12 * max(1.0, abs(ref.LowerBound()))
    # one exchange per chunk of passes, not per level
    assert sweeps[0].exchanges == sum(len(sweeps[0].chunks(n)) for n in (1, 3, 2))


def test_ghost_depth_is_what_the_pass_count_needs():
    """2 n + 2 ghost rows for n passes between exchanges: what is wrong at the rim of a window moves two rows per pass.  More
    passes than that are refused — and would be wrong: forced, the owned rows differ from the unpartitioned sweep"""
    H, W, L, world = 8, 6, 3, 3
    gm, costs = _global(H, W, L, world, "dense", seed=2)
    assert [OV.max_passes_between_exchanges(g) for g in (4, 6, 12, 22)] == [1, 2, 5, 10]
    with pytest.raises(ValueError, match="at least 4"):
        OV.strip_window_part(H, W, L, "dense", 0, world, 2, costs=costs)
    parts = [OV.strip_window_part(H, W, L, "dense", r, world, 4, costs=costs) for r in range(world)]
    with pytest.raises(ValueError, match="ghost rows"):
        _cpu_sweeps(parts, M.REPAM_ANISOTROPIC, chunk=2)
    for bad in (5, 7):
        with pytest.raises(ValueError, match="even"):
            OV.strip_window_part(H, W, L, "dense", 0, world, bad, costs=costs)
    with pytest.raises(ValueError, match="even number of rows"):
        OV.strip_window_part(7, W, L, "dense", 0, world, 4, costs=costs)
    sweeps, duals = _cpu_sweeps(parts, M.REPAM_ANISOTROPIC)
    for s in sweeps:
        s.chunk = 2                                  # one pass too many for 4 ghost rows
    ref = Oracle(gm); ref.set_reparametrization(M.REPAM_ANISOTROPIC)
    ref.ComputePass(2); OV.run_overlapped(sweeps, 2)
    with pytest.raises(AssertionError):
        _assert_owned_equal_global(gm, parts, duals, ref)


def test_windows_generate_their_costs_from_the_global_stream():
    """bench.py's ranks fill their windows in HBM from the counter stream: the fill descriptors name the global model's
    stream positions (materialised on the host here), dense and Potts"""
    H, W, L, world, g = 6, 5, 3, 3, 4
    for pairwise in ("dense", "potts"):
        gm, costs = _global(H, W, L, world, pairwise, seed=7)
        for r in range(world):
            a = OV.strip_window_part(H, W, L, pairwise, r, world, g, seed=7)
            b = OV.strip_window_part(H, W, L, pairwise, r, world, g, costs=costs)
            materialise_fills(a)
            assert np.array_equal(a.model.dual_data, b.model.dual_data)
            assert np.array_equal(np.asarray(a.model.const_data).reshape(-1), np.asarray(b.model.const_data).reshape(-1))
            assert (a.r0, a.r1) == (max(0, r * H - g), min(world * H, (r + 1) * H + g))


WORKER = r"""
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, {root!r})
from lp_mp_amd import model as M, multi_gpu as MG, overlap as OV
from tests.mgpu_helpers import OracleEngine, materialise_fills
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
p = materialise_fills(OV.strip_window_part(8, 6, 3, "dense", rank, world, 6, seed=5))     # costs from the stream, as bench.py's ranks
d = p.model.dual_data.copy()
e = OracleEngine(p.model, d); e.set_reparametrization(M.REPAM_ANISOTROPIC)
sw = OV.OverlapSweep(torch, p, e, torch.from_numpy(d))
comm = MG.DistComm(dist, torch)
sw.compute_pass(comm, 5); sw.compute_pass(comm, 2)
lb = comm.all_reduce_sum(sw.local_lower_bound())
np.save(os.path.join({out!r}, f"ov_duals_{{rank}}.npy"), d)
if rank == 0:
    np.save(os.path.join({out!r}, "ov_lb.npy"), np.array([lb, sw.exchanges]))
dist.destroy_process_group()
"""


@pytest.mark.parametrize("world", [2, 4, 8])
def test_gloo_run_equals_the_unpartitioned_oracle(tmp_path, world):
    """what bench.py --gpus N does per rank, over torch.distributed (gloo): 7 passes in chunks of 2 = 4 exchanges"""
    script = tmp_path / "ov_worker.py"
    script.write_text(WORKER.format(root=ROOT, out=str(tmp_path)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
                           "--master-addr", "127.0.0.1", "--master-port", str(29540 + world), str(script)], env=env, cwd=ROOT, timeout=600)
    gm, costs = _global(8, 6, 3, world, "dense", seed=5)
    ref = Oracle(gm); ref.set_reparametrization(M.REPAM_ANISOTROPIC)
    ref.ComputePass(7)
    parts = [OV.strip_window_part(8, 6, 3, "dense", r, world, 6, costs=costs) for r in range(world)]
    _assert_owned_equal_global(gm, parts, [np.load(tmp_path / f"ov_duals_{k}.npy") for k in range(world)], ref)
    lb, ex = np.load(tmp_path / "ov_lb.npy")
    assert abs(lb - ref.LowerBound()) <= 1e-12 * abs(ref.LowerBound()) and ex == 3 + 1


def _device_sweeps(parts, mode, chunk=None):
    from lp_mp_amd import engine as E
    dev = torch.device("cuda:0")
    sweeps, tensors = [], []
    for p in parts:
        dual = torch.from_numpy(p.model.dual_data.copy()).to(dev)
        eng = E.Engine(0); eng.set_stream(torch.cuda.current_stream().cuda_stream)
        eng.upload(p.model, dual_dev=dual.data_ptr(), keep=dual)
        eng.set_reparametrization(mode)
        sweeps.append(OV.OverlapSweep(torch, p, eng, dual, chunk)); tensors.append(dual)
    return sweeps, tensors


@pytest.mark.gpu
@pytest.mark.parametrize("H,W,L,world,g,pairwise", [(12, 10, 32, 3, 6, "dense"), (16, 12, 8, 4, 8, "dense"), (12, 12, 8, 3, 4, "potts"),
                                                    (12, 10, 21, 2, 6, "dense"), (24, 16, 16, 3, 12, "dense")])
@pytest.mark.parametrize("mode", [M.REPAM_ANISOTROPIC, M.REPAM_DAMPED_UNIFORM])
def test_overlapped_strips_on_device_equal_the_unpartitioned_oracle(H, W, L, world, g, pairwise, mode):
    """real HIP engines, all parts on the one GPU of the test box"""
    gm, costs = _global(H, W, L, world, pairwise, seed=9)
    ref = Oracle(gm); ref.set_reparametrization(mode)
    parts = [OV.strip_window_part(H, W, L, pairwise, r, world, g, costs=costs) for r in range(world)]
    sweeps, tensors = _device_sweeps(parts, mode)
    try:
        for n in (1, 4, 7):
            ref.ComputePass(n)
            OV.run_overlapped(sweeps, n)
            torch.cuda.synchronize()
            _assert_owned_equal_global(gm, parts, [t.cpu().numpy() for t in tensors], ref)
            lb = sum(s.local_lower_bound() for s in sweeps)
            assert abs(lb - ref.LowerBound()) <= 1e-9 * max(1.0, abs(ref.LowerBound()))
    finally:
        for s in sweeps:
            s.engine.close()


@pytest.mark.gpu
def test_overlapped_strips_with_the_joined_pass_chain_forced(monkeypatch):
    """the path full-size windows take — n passes between two exchanges as ONE persistent launch in banded ticket order
    (engine.cpp rotation_chain) — forced onto small windows"""
    monkeypatch.setenv("LPMP_ROT_BANDS", "4")
    H, W, L, world, g = 32, 32, 32, 3, 8
    gm, costs = _global(H, W, L, world, "dense", seed=3)
    ref = Oracle(gm); ref.set_reparametrization(M.REPAM_ANISOTROPIC)
    parts = [OV.strip_window_part(H, W, L, "dense", r, world, g, costs=costs) for r in range(world)]
    sweeps, tensors = _device_sweeps(parts, M.REPAM_ANISOTROPIC)
    try:
        for n in (3, 6):
            ref.ComputePass(n); OV.run_overlapped(sweeps, n); torch.cuda.synchronize()
            _assert_owned_equal_global(gm, parts, [t.cpu().numpy() for t in tensors], ref)
        kt = sweeps[1].engine.kernel_timing()
    finally:
        for s in sweeps:
            s.engine.close()


@pytest.mark.gpu
def test_full_size_windows_equal_one_engine_on_the_whole_grid():
    """BASELINE configs[2] per rank, two ranks: the two 1024 x 1024 windows (12 ghost rows, costs generated in HBM from the global
    stream, the joined-pass chain launch of 5 passes between exchanges) against ONE engine on the whole 2048 x 1024 grid: every
    owned dual bit-identical after 5 + 2 passes (compared on the device), same bound.  (That single engine is the oracle's equal
    at 1024 x 1024: test_what_bench_times_against_the_oracle_at_full_size.)"""
    from lp_mp_amd import engine as E, multi_gpu as MG
    H = W = 1024; L = 32; world = 2; g = 12
    dev = torch.device("cuda:0")
    if torch.cuda.get_device_properties(0).total_memory < 100e9:
        pytest.skip("needs ~80 GB of device memory")
    stream = torch.cuda.current_stream().cuda_stream
    # the whole grid on one engine
    gm = S.grid_model(world * H, W, L, order="colour_major", seed=1, device_const=True, unaries=np.zeros(world * H * W * L))
    n_g, E_g = world * H * W, gm.n_factors - world * H * W
    gconst = torch.empty(E_g * L * L, dtype=torch.float64, device=dev)
    gdual = torch.zeros(n_g * L + E_g * 2 * L, dtype=torch.float64, device=dev)
    E.synth_fill(gconst.data_ptr(), gconst.numel(), 1, n_g * L, stream)
    E.synth_fill(gdual.data_ptr(), n_g * L, 1, 0, stream)
    torch.cuda.synchronize()
    ge = E.Engine(0); ge.set_stream(stream)
    ge.upload(gm, const_dev=gconst.data_ptr(), dual_dev=gdual.data_ptr(), keep=(gconst, gdual))
    ge.set_reparametrization(M.REPAM_ANISOTROPIC)
    sweeps, keep = [], []
    try:
        for r in range(world):
            p = OV.strip_window_part(H, W, L, "dense", r, world, g, 1)
            m = p.model
            const = torch.empty(int(m.const_sizes().sum()), dtype=torch.float64, device=dev)
            dual = torch.zeros(int(m.dual_sizes().sum()), dtype=torch.float64, device=dev)
            MG.fill_device_costs(torch, E, p, const, dual, stream)
            e = E.Engine(0); e.set_stream(stream)
            e.upload(m, const_dev=const.data_ptr(), dual_dev=dual.data_ptr(), keep=(const, dual))
            e.set_reparametrization(M.REPAM_ANISOTROPIC)
            sweeps.append(OV.OverlapSweep(torch, p, e, dual)); keep.append((const, dual))
        assert sweeps[0].chunk == 5
        goff = torch.from_numpy(gm.dual_offsets()).to(dev)
        for n in (5, 2):
            ge.compute_pass(n)
            OV.run_overlapped(sweeps, n)
            torch.cuda.synchronize()
            for s, (_, dual) in zip(sweeps, keep):
                p = s.part
                lo = torch.from_numpy(p.model.dual_offsets()).to(dev)
                nv = p.vars_global.shape[0]
                own_v = torch.from_numpy(np.nonzero(p.owned[:nv])[0]).to(dev)
                own_e = torch.from_numpy(np.nonzero(p.owned[nv:])[0]).to(dev)
                gv = torch.from_numpy(p.vars_global).to(dev)[own_v]
                gedge = torch.from_numpy(p.edges_global).to(dev)[own_e]
                ar = torch.arange(L, device=dev); ar2 = torch.arange(2 * L, device=dev)
                assert torch.equal(dual[(lo[own_v][:, None] + ar).reshape(-1)], gdual[(goff[gv][:, None] + ar).reshape(-1)])
                assert torch.equal(dual[(lo[nv + own_e][:, None] + ar2).reshape(-1)], gdual[(goff[n_g + gedge][:, None] + ar2).reshape(-1)])
            lb, lbg = sum(s.local_lower_bound() for s in sweeps), ge.lower_bound()
            assert abs(lb - lbg) <= 1e-12 * abs(lbg)
    finally:
        ge.close()
        for s in sweeps:
            s.engine.close()


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(12))
def test_overlapped_strips_random_shapes_modes_and_chunks_on_device(seed):
    """random strip height / width (odd widths too), label count, pairwise kind, number of strips, ghost depth, passes between
    exchanges (any value the depth allows) and weight mode; separate calls of random length: owned duals and the summed bound are
    the oracle's on the whole grid"""
    rng = np.random.default_rng(500 + seed)
    world = int(rng.integers(2, 6)); g = 2 * int(rng.integers(2, 6))
    H = 2 * int(rng.integers(max(1, g // 2), 9)); W = int(rng.integers(2, 14))
    L = int(rng.choice([2, 3, 4, 5, 8, 16, 21, 32])); pairwise = "potts" if rng.uniform() < 0.35 else "dense"
    mode = [M.REPAM_ANISOTROPIC, M.REPAM_ANISOTROPIC2, M.REPAM_UNIFORM, M.REPAM_DAMPED_UNIFORM][int(rng.integers(0, 4))]
    chunk = int(rng.integers(1, OV.max_passes_between_exchanges(g) + 1))
    gm, costs = _global(H, W, L, world, pairwise, seed=seed)
    ref = Oracle(gm); ref.set_reparametrization(mode)
    parts = [OV.strip_window_part(H, W, L, pairwise, r, world, g, costs=costs) for r in range(world)]
    sweeps, tensors = _device_sweeps(parts, mode, chunk)
    try:
        for n in rng.integers(1, 6, 3):
            ref.ComputePass(int(n)); OV.run_overlapped(sweeps, int(n)); torch.cuda.synchronize()
            _assert_owned_equal_global(gm, parts, [t.cpu().numpy() for t in tensors], ref)
            lb = sum(s.local_lower_bound() for s in sweeps)
            assert abs(lb - ref.LowerBound()) <= 1e-9 * max(1.0, abs(ref.LowerBound())), (seed, lb, ref.LowerBound())
    finally:
        for s in sweeps:
            s.engine.close()
