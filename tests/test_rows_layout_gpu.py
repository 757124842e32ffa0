"""Rows layout (lpmp_set_rows_layout, engine.cpp): every dense pairwise factor as one contiguous row [table | m1 | m2] of an
engine-private buffer — the kernels are the same (they add device offsets to the const / dual base pointers), the packed dual
array stays the format of every call that hands duals over.  Duals, bounds and labels of the oracle bit for bit on every kernel
class, through every hand-over: download / upload, a borrowed buffer read and written by the caller, the boundary kernels."""
import numpy as np
import pytest
import torch

from lp_mp_amd import engine as E
from lp_mp_amd import model as M
from lp_mp_amd import synthetic as S
from oracle.binding import Oracle

pytestmark = pytest.mark.gpu

MODELS = {
    "colour-major grid, 32 labels": lambda: S.grid_model(20, 18, 32, order="colour_major", seed=3, compute_primal=True),
    "row-major grid, 16 labels (mailbox chain)": lambda: S.grid_model(24, 20, 16, order="row_major", seed=4, compute_primal=True),
    "random graph, 16 labels (C4 shape)": lambda: S.counter_graph_model(3000, 15000, 16, 5),
    "random graph, 8 labels": lambda: S.random_graph_model(2000, 7000, 8, seed=6, compute_primal=True),
    "grid, 21 labels (run-time dims)": lambda: S.grid_model(14, 15, 21, order="colour_major", seed=7, compute_primal=True),
    "grid, 48 labels (streaming kernel)": lambda: S.grid_model(8, 9, 48, order="colour_major", seed=8, compute_primal=True),
    "Potts grid (nothing to lay out)": lambda: S.grid_model(16, 16, 8, pairwise="potts", order="colour_major", seed=9, compute_primal=True),
}


@pytest.mark.parametrize("name", list(MODELS))
def test_rows_layout_equals_the_oracle(name):
    m = MODELS[name]()
    o = Oracle(m)
    e = E.Engine(0)
    try:
        e.upload(m, rows_layout=True)
        assert e.rows_layout == ("Potts" not in name)
        for mode in (M.REPAM_ANISOTROPIC, M.REPAM_DAMPED_UNIFORM):
            o.set_reparametrization(mode); e.set_reparametrization(mode)
            o.ComputeForwardPass(); e.forward_pass()
            assert np.array_equal(e.download_duals(), o.duals()), (name, mode, "forward")
            o.ComputeBackwardPass(); e.backward_pass()
            for n in (1, 3):
                o.ComputePass(n); e.compute_pass(n)
                assert np.array_equal(e.download_duals(), o.duals()), (name, mode, n)
                lb, lbo = e.lower_bound(), o.LowerBound()
                assert abs(lb - lbo) <= 1e-9 * max(1.0, abs(lbo))
                e.invalidate_lower_bounds()            # tracked bounds against a full recomputation (from the rows)
                assert abs(e.lower_bound() - lb) <= 1e-12 * max(1.0, abs(lb))
        flb = e.factor_lower_bounds()
        assert np.max(np.abs(flb - np.array([o.factor_lower_bound(f) for f in range(m.n_factors)]))) <= 1e-9
        # upload: the caller's packed duals replace the rows' vectors
        d = o.duals().copy(); d[m.dual_offsets()[m.n_factors - 1]:] += 0.25
        o.set_duals(d); e.upload_duals(d)
        o.ComputePass(2); e.compute_pass(2)
        assert np.array_equal(e.download_duals(), o.duals())
        if "compute_primal" in MODELS[name].__code__.co_names or "grid" in name:
            o.ComputePassAndPrimal(5); e.compute_pass_and_primal(5)
            assert np.array_equal(e.download_duals(), o.duals()) and np.array_equal(e.download_primal(), o.primal())
            assert abs(e.evaluate_primal() - o.EvaluatePrimal()) <= 1e-9 * max(1.0, abs(o.EvaluatePrimal()))
    finally:
        e.close()


def test_borrowed_dual_buffer_is_the_callers_state_at_synchronize():
    """LPMP_MEM_DEVICE duals with the rows layout: lpmp_synchronize writes the rows' vectors out (the buffer holds the state),
    what the caller then writes into the buffer — unary AND pairwise entries — is what the next pass starts from"""
    m = S.counter_graph_model(1500, 6000, 16, 3)
    o = Oracle(m); o.set_reparametrization(M.REPAM_ANISOTROPIC)
    dev = torch.device("cuda:0")
    dual = torch.from_numpy(m.dual_data.copy()).to(dev)
    e = E.Engine(0)
    try:
        e.set_stream(torch.cuda.current_stream().cuda_stream)
        e.upload(m, dual_dev=dual.data_ptr(), keep=dual, rows_layout=True)
        e.set_reparametrization(M.REPAM_ANISOTROPIC)
        assert e.rows_layout
        o.ComputePass(2); e.compute_pass(2)
        e.synchronize()
        assert np.array_equal(dual.cpu().numpy(), o.duals())
        # the caller edits both kinds of entries in place
        d = o.duals().copy()
        off = m.dual_offsets()
        d[off[3]: off[4]] += 0.5                              # a unary
        d[off[m.n_factors - 2]: off[m.n_factors - 1]] -= 0.125  # a pairwise factor's two vectors
        o.set_duals(d); dual.copy_(torch.from_numpy(d).to(dev))
        e.invalidate_lower_bounds()
        assert abs(e.lower_bound() - o.LowerBound()) <= 1e-9 * max(1.0, abs(o.LowerBound()))
        o.ComputePass(1); e.compute_pass(1)
        e.synchronize()
        assert np.array_equal(dual.cpu().numpy(), o.duals())
        # lpmp_device_duals hands the packed array over as well
        o.ComputePass(1); e.compute_pass(1)
        assert e.device_duals_ptr() == dual.data_ptr()
        torch.cuda.synchronize()
        assert np.array_equal(dual.cpu().numpy(), o.duals())
    finally:
        e.close()


def test_partitioned_sweep_boundary_kernels_with_the_rows_layout():
    """multi_gpu.PartitionedSweep on engines with the rows layout, all parts on the one GPU: the boundary step reads and writes
    the unaries' and ghosts' vectors through lpmp_boundary_* (device offsets), the main sweeps the rows — against the same sweep
    on engines with the packed layout, bit for bit"""
    from lp_mp_amd import multi_gpu as MG
    dev = torch.device("cuda:0")
    H, W, L, world = 12, 10, 16, 3
    duals = {}
    for rows in (False, True):
        sweeps, tensors = [], []
        for k in range(world):
            p = MG.strip_local_part(H, W, L, "dense", "colour_major", k, world, 1)
            dual = torch.from_numpy(p.model.dual_data.copy()).to(dev)
            eng = E.Engine(0); eng.set_stream(torch.cuda.current_stream().cuda_stream)
            eng.upload(p.model, dual_dev=dual.data_ptr(), keep=dual, rows_layout=rows)
            eng.set_reparametrization(M.REPAM_ANISOTROPIC)
            assert eng.rows_layout == rows
            sweeps.append(MG.PartitionedSweep(torch, p, eng, dual, M.REPAM_ANISOTROPIC, None, "pass")); tensors.append(dual)
        MG.run_lockstep(sweeps, 3)
        lb = sum(s.local_lower_bound() for s in sweeps)
        for s in sweeps:
            s.engine.synchronize()
        duals[rows] = ([t.cpu().numpy() for t in tensors], lb)
        for s in sweeps:
            s.engine.close()
    for a, b in zip(duals[False][0], duals[True][0]):
        assert np.array_equal(a, b)
    assert abs(duals[False][1] - duals[True][1]) <= 1e-12 * abs(duals[False][1])
