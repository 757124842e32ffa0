"""Round 6, on the GPU box.
* Two PROCESSES hand DEVICE tensors to the collective backend — the code path of an RCCL run (device all-to-all-v; `--overlap-exchange`'s
  asynchronous exchange_begin / exchange_end) with real data between two ranks.  RCCL refuses two ranks on one device, so on the 1-GPU
  box the backend is gloo's own CUDA collectives (LPMP_DIST_DEVICE_COLLECTIVES=1: multi_gpu.DistComm stops staging through the host);
  everything above the backend — buffers, split sizes, stream ordering between the engine's kernels, pack / unpack and the
  collective — is what runs over RCCL.
* Hard constraints: +inf entries of pairwise tables, Potts differences of +inf.
* The joined-pass launch away from the headline size: wide grids (lag and depth from the reach of the dependencies; the tiled ticket
  order where no band order fits), the tiled order forced on small models.
* More than 32 labels: every LDS size class of the streaming kernel, ragged label counts in one launch."""
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
    assert d["n_gpus"] == 2 and d["backend"].startswith("gloo (device tensors") and d["launch"]["exchange_buffers"] == "device"
    assert d["schedule"] == "lockstep" and d["overlap_exchange"] is False
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


# ---- hard constraints: +inf entries in pairwise tables -------------------------------------------------------------------------
def _hard_grid(H, W, L, seed, order, frac=0.35, potts=False):
    """a grid MRF whose pairwise tables forbid label pairs (+inf, as the reference's users write hard constraints; its own `matrix`
    pads rows with +inf, include/vector.hxx:708-735); every row and column keeps a finite entry (the diagonal), so all min-marginals
    stay finite — an all-inf row makes the reference itself produce inf - inf"""
    import numpy as np
    from lp_mp_amd import synthetic as S
    rng = np.random.default_rng(seed)
    m = S.grid_model(H, W, L, pairwise="potts" if potts else "dense", order=order, seed=seed)
    c = np.asarray(m.const_data)
    if potts:
        c[rng.random(c.shape[0]) < frac] = np.inf                    # hard equality constraints
    else:
        T = c.reshape(-1, L, L)
        mask = rng.random(T.shape) < frac
        mask[:, np.arange(L), np.arange(L)] = False
        T[mask] = np.inf
    return m


@pytest.mark.parametrize("L", [3, 4, 8, 16, 32, 40])
@pytest.mark.parametrize("order", ["row_major", "colour_major"])
def test_forbidden_label_pairs_dense(L, order):
    import numpy as np
    from lp_mp_amd import engine as E, model as M
    from oracle.binding import Oracle
    m = _hard_grid(9, 7, L, 60 + L, order)
    e = E.Engine(0)
    try:
        for mode in (M.REPAM_ANISOTROPIC, M.REPAM_DAMPED_UNIFORM, M.REPAM_UNIFORM, M.REPAM_ANISOTROPIC2):
            o = Oracle(m); o.set_reparametrization(mode)
            e.upload(m); e.set_reparametrization(mode)
            for n in (1, 3):
                o.ComputePass(n); e.compute_pass(n)
                d = e.download_duals()
                assert np.isfinite(d).all() and np.array_equal(d, o.duals()), (L, order, mode, n)
                assert abs(e.lower_bound() - o.LowerBound()) <= 1e-9 * max(1.0, abs(o.LowerBound()))
    finally:
        e.close()


@pytest.mark.parametrize("L", [2, 5, 8, 32])
def test_hard_equality_constraints_potts(L):
    import numpy as np
    from lp_mp_amd import engine as E, model as M
    from oracle.binding import Oracle
    m = _hard_grid(8, 9, L, 80 + L, "colour_major", potts=True)
    e = E.Engine(0)
    try:
        for mode in (M.REPAM_ANISOTROPIC, M.REPAM_DAMPED_UNIFORM):
            o = Oracle(m); o.set_reparametrization(mode)
            e.upload(m); e.set_reparametrization(mode)
            for n in (1, 4):
                o.ComputePass(n); e.compute_pass(n)
                d = e.download_duals()
                assert np.isfinite(d).all() and np.array_equal(d, o.duals()), (L, mode, n)
                assert abs(e.lower_bound() - o.LowerBound()) <= 1e-9 * max(1.0, abs(o.LowerBound()))
    finally:
        e.close()


def test_forbidden_label_pairs_through_the_joined_pass_launch():
    """the headline path's launch (colour-major grid: n joined passes as one persistent launch) on a model with forbidden label pairs"""
    import numpy as np
    from lp_mp_amd import engine as E, model as M
    from oracle.binding import Oracle
    m = _hard_grid(96, 96, 32, 7, "colour_major", frac=0.2)
    o = Oracle(m); o.set_reparametrization(M.REPAM_ANISOTROPIC)
    e = E.Engine(0)
    try:
        e.upload(m); e.set_reparametrization(M.REPAM_ANISOTROPIC)
        e.compute_pass(6); o.ComputePass(6)
        assert np.array_equal(e.download_duals(), o.duals())
        e.compute_pass(1); o.ComputePass(1)
        assert np.array_equal(e.download_duals(), o.duals())
        assert abs(e.lower_bound() - o.LowerBound()) <= 1e-9 * abs(o.LowerBound())
    finally:
        e.close()


# ---- joined passes on WIDE grids: lag and depth of the Infinity-Cache ticket order follow the reach of the dependencies ----------
@pytest.mark.parametrize("H,W,want", [(16, 4096, "chain"), (5, 16384, "sweep")])
def test_wide_grids_joined_launch_geometry_against_the_oracle(H, W, want, capfd, monkeypatch):
    """HBM-sized colour-major grids whose ROWS are long (80 MB and 320 MB of a step's algorithmic bytes per grid row of one colour; the
    headline grid: 20 MB): the engine chooses lag and depth of the skewed ticket order from that reach (engine.cpp rot_geometry) —
    depth 2 and a lag that covers one row plus slack for the 4096-wide grid, no band order at all when even that window cannot sit in
    the Infinity Cache (16384 wide); calls of 4 and more passes take the tiled order there — and the duals equal the oracle's bit for
    bit every way"""
    import numpy as np
    from lp_mp_amd import engine as E, model as M, synthetic as S
    from oracle.binding import Oracle
    monkeypatch.setenv("LPMP_ROT_VERBOSE", "1")
    m = S.grid_model(H, W, 32, order="colour_major", seed=11)
    o = Oracle(m); o.set_reparametrization(M.REPAM_ANISOTROPIC)
    e = E.Engine(0)
    try:
        e.upload(m)
        assert e.L.lpmp_streaming_access(e.h) == 1                   # tables + duals > 1 GiB
        e.set_reparametrization(M.REPAM_ANISOTROPIC)
        e.enable_kernel_timing(True)
        e.compute_pass(3); o.ComputePass(3)
        kt = e.kernel_timing()
        e.enable_kernel_timing(False)
        names = [v["kernel"] for v in kt.values()]
        assert len(names) == 1 and names[0].startswith(want + "_dense_pk_kernel<32"), names
        assert np.array_equal(e.download_duals(), o.duals())
        e.compute_pass(9); o.ComputePass(9)                          # (more than ROT_EXPLICIT_MAX passes: the periodic template)
        assert np.array_equal(e.download_duals(), o.duals())
        assert abs(e.lower_bound() - o.LowerBound()) <= 1e-9 * abs(o.LowerBound())
        err = capfd.readouterr().err
        # the 9-pass call: rows this long take the TILED ticket order (tiles grown over the block dependencies: nothing in it depends on
        # the width of the grid), whether or not a band order would still fit
        assert "tiled order" in err, err[-800:]
        if want == "chain":
            assert "depth 2" in err, err[-800:]                      # the 3-pass call: band order, depth 2
        else:
            assert "reach further than the Infinity Cache window" in err, err[-800:]   # the 3-pass call: one launch per step
    finally:
        e.close()


# ---- more than 32 labels: the streaming dense class with LDS sized by the launch's label counts --------------------------------------
@pytest.mark.parametrize("L,H,W", [(33, 12, 14), (40, 9, 11), (64, 10, 9), (65, 5, 6), (130, 6, 7), (300, 3, 4)])
def test_streaming_class_every_lds_size_against_the_oracle(L, H, W):
    """33 ... 512 labels (class dense_big: one wave per unary, record and op fields as scalars, tables streamed in 16-row blocks): the
    LDS of a wave holds three vectors of the launch's largest label count rounded up to 64 — every size class, both orders, two
    weight modes and the residual send rule against the oracle bit for bit"""
    import numpy as np
    from lp_mp_amd import engine as E, model as M, synthetic as S
    from oracle.binding import Oracle
    e = E.Engine(0)
    try:
        for order in ("colour_major", "row_major"):
            m = S.grid_model(H, W, L, order=order, seed=L)
            o = Oracle(m)
            e.upload(m)
            for rtype in (0, 1):
                o.set_reparametrization_type(rtype); e.set_reparametrization_type(rtype)
                for mode in (M.REPAM_ANISOTROPIC, M.REPAM_DAMPED_UNIFORM):
                    o.set_reparametrization(mode); e.set_reparametrization(mode)
                    assert list(e.plan.schedule_classes(M.FORWARD, mode)) == ["dense_big"]
                    for n in (1, 3):
                        o.ComputePass(n); e.compute_pass(n)
                        assert np.array_equal(e.download_duals(), o.duals()), (L, order, rtype, mode, n)
                    assert abs(e.lower_bound() - o.LowerBound()) <= 1e-9 * max(1.0, abs(o.LowerBound()))
    finally:
        e.close()


def test_ragged_label_counts_in_one_launch_of_the_streaming_class():
    """variables with 34 ... 90 labels in one model (rectangular tables): the launch's LDS is sized by its largest label count"""
    import numpy as np
    from lp_mp_amd import engine as E, model as M
    from oracle.binding import Oracle
    rng = np.random.default_rng(5)
    n = 40
    labels = rng.integers(34, 91, n)
    from lp_mp_amd import synthetic as S
    b = M.ModelBuilder(2, S.mrf_mtypes())
    u = [int(b.add_vector_factors(0, rng.random((1, int(l))))[0]) for l in labels]
    for i in range(n - 1):
        for j in (i + 1, i + 7):
            if j >= n:
                continue
            p = int(b.add_dense_pairwise(1, rng.random((1, int(labels[i]), int(labels[j]))))[0])
            b.add_messages(0, u[i], p); b.add_messages(1, u[j], p)
            b.add_relations(u[i], p); b.add_relations(p, u[j])
    m = b.finish()
    o = Oracle(m); o.set_reparametrization(M.REPAM_ANISOTROPIC)
    e = E.Engine(0)
    try:
        e.upload(m); e.set_reparametrization(M.REPAM_ANISOTROPIC)
        assert "dense_big" in list(e.plan.schedule_classes(M.FORWARD, M.REPAM_ANISOTROPIC))
        for k in (1, 2, 4):
            o.ComputePass(k); e.compute_pass(k)
            assert np.array_equal(e.download_duals(), o.duals())
        assert abs(e.lower_bound() - o.LowerBound()) <= 1e-9 * max(1.0, abs(o.LowerBound()))
    finally:
        e.close()


# ---- tiled ticket order of the joined-pass launch (experiment, LPMP_ROT_TILES) -------------------------------------------------------
@pytest.mark.parametrize("pairwise,L,H,W", [("dense", 32, 40, 36), ("dense", 8, 60, 70), ("dense", 16, 33, 31)])
@pytest.mark.parametrize("tiles,depth", [(8, 4), (3, 2), (20, 6), (1, 4)])
def test_joined_passes_in_the_tiled_order(pairwise, L, H, W, tiles, depth, monkeypatch):
    """the joined-pass launch with its tickets in the TILED order (blocks grouped by tiles grown over the block dependencies; inside a
    group of `depth` steps a block runs in the phase of its tile or of its latest predecessor's): forced on small models, every
    pass count, against the oracle bit for bit"""
    import numpy as np
    from lp_mp_amd import engine as E, model as M, synthetic as S
    from oracle.binding import Oracle
    monkeypatch.setenv("LPMP_ROT_BANDS", "4"); monkeypatch.setenv("LPMP_ROT_TILES", str(tiles)); monkeypatch.setenv("LPMP_ROT_DEPTH", str(depth))
    monkeypatch.setenv("LPMP_ROT_VERBOSE", "1")
    m = S.grid_model(H, W, L, pairwise=pairwise, order="colour_major", seed=L + tiles)
    o = Oracle(m); o.set_reparametrization(M.REPAM_ANISOTROPIC)
    e = E.Engine(0)
    try:
        e.upload(m); e.set_reparametrization(M.REPAM_ANISOTROPIC)
        for n in (1, 5, 2, 9, 70, 12, 31, 13):
            e.enable_kernel_timing(True)
            e.compute_pass(n); o.ComputePass(n)
            kt = e.kernel_timing(); e.reset_kernel_timing(); e.enable_kernel_timing(False)
            assert all(v["kernel"].startswith("chain_") for v in kt.values()), kt
            assert np.array_equal(e.download_duals(), o.duals()), (n,)
        assert abs(e.lower_bound() - o.LowerBound()) <= 1e-9 * max(1.0, abs(o.LowerBound()))
    finally:
        e.close()
