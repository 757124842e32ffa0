"""Parity of the HIP sweep (through the C ABI) against the CPU oracle, plus size-independent
properties at BASELINE.json's full sizes.  Tolerance of the north star: lower bound within 1e-5
relative after the same number of passes on identical inputs; the duals themselves are required to
match to 1e-12 absolute (they are bit-identical in practice: min and + are exact and the evaluation
order is copied)."""
import os

import numpy as np
import pytest

from lp_mp_amd import model as M
from lp_mp_amd import synthetic as S
from lp_mp_amd import engine as E
from oracle.binding import Oracle

pytestmark = pytest.mark.gpu

LB_RTOL = 1e-5          # BASELINE.json north_star
DUAL_ATOL = 1e-12
MODES = (M.REPAM_ANISOTROPIC, M.REPAM_ANISOTROPIC2, M.REPAM_UNIFORM, M.REPAM_DAMPED_UNIFORM)


@pytest.fixture(scope="module")
def eng():
    e = E.Engine(0)
    yield e
    e.close()


def _check(eng, m, mode, passes, exact=True):
    o = Oracle(m)
    o.set_reparametrization(mode)
    eng.upload(m)
    eng.set_reparametrization(mode)
    lb0 = eng.lower_bound()
    assert abs(lb0 - o.LowerBound()) <= LB_RTOL * max(1.0, abs(lb0))
    prev = lb0
    for _ in range(passes):
        o.ComputePass(1)
        eng.compute_pass(1)
        lb, lbo = eng.lower_bound(), o.LowerBound()
        assert abs(lb - lbo) <= LB_RTOL * max(1.0, abs(lbo)), (lb, lbo)
        assert lb >= prev - 1e-8 * max(1.0, abs(prev))          # dual ascent
        prev = lb
    d, do = eng.download_duals(), o.duals()
    assert np.max(np.abs(d - do)) <= DUAL_ATOL
    if exact:
        assert np.array_equal(d, do)
    flb = eng.factor_lower_bounds()
    oflb = np.array([o.factor_lower_bound(f) for f in range(min(m.n_factors, 3000))])
    assert np.max(np.abs(flb[:oflb.shape[0]] - oflb)) <= DUAL_ATOL
    return prev


# ---- the non-temporal (streamed-once) kernel instantiations ------------------------------------
# bench.py's C3 model is HBM-sized, so the engine picks the NT = true instantiations of the exact dense / Potts /
# streaming kernels (engine.cpp: tables + duals > 1 GiB; kernel names end in ", true>" in BENCH_r*.json).  LPMP_NT
# forces the choice for a model of any size, so the SAME instantiations are compared with the oracle here, bit for bit.
@pytest.fixture(params=[0, 1], ids=["nt0", "nt1"])
def nt_eng(request, monkeypatch):
    monkeypatch.setenv("LPMP_NT", str(request.param))
    e = E.Engine(0)
    e.want_nt = request.param

    def upload_checked(m, *a, _up=e.upload, **k):
        _up(m, *a, **k)
        assert e.L.lpmp_streaming_access(e.h) == e.want_nt
    e.upload = upload_checked
    yield e
    e.close()


@pytest.mark.parametrize("L", [4, 8, 16, 32])
@pytest.mark.parametrize("order", ["row_major", "colour_major"])
def test_dense_fast_path_both_access_policies(nt_eng, L, order):
    m = S.grid_model(13, 11, L, order=order, seed=L)
    for mode in MODES:
        _check(nt_eng, m, mode, 3)
    assert list(nt_eng.plan.schedule_classes(M.FORWARD, M.REPAM_ANISOTROPIC)) == ["dense%d" % L]


@pytest.mark.parametrize("L", [4, 8, 16, 32])
@pytest.mark.parametrize("order", ["row_major", "colour_major"])
def test_potts_fast_path_both_access_policies(nt_eng, L, order):
    m = S.grid_model(12, 15, L, pairwise="potts", order=order, seed=10 + L)
    for mode in MODES:
        _check(nt_eng, m, mode, 3)
    assert list(nt_eng.plan.schedule_classes(M.FORWARD, M.REPAM_ANISOTROPIC)) == ["potts%d" % L]


@pytest.mark.parametrize("L", [33, 64, 130])
def test_streaming_dense_kernel_both_access_policies(nt_eng, L):
    m = S.grid_model(5, 6, L, order="colour_major", seed=200 + L)
    for mode in (M.REPAM_ANISOTROPIC, M.REPAM_DAMPED_UNIFORM):
        _check(nt_eng, m, mode, 2)
    assert list(nt_eng.plan.schedule_classes(M.FORWARD, M.REPAM_UNIFORM)) == ["dense_big"]


@pytest.mark.parametrize("pairwise,L", [("dense", 32), ("dense", 16), ("potts", 8)])
def test_multi_pass_rotated_schedule_both_access_policies(nt_eng, pairwise, L):
    """the launch sequence bench.py times: lpmp_compute_pass(n) on a colour-major grid = H, W, (K, W)^(n-1), T"""
    m = S.grid_model(14, 10, L, pairwise=pairwise, order="colour_major", seed=31)
    o = Oracle(m)
    o.set_reparametrization(M.REPAM_ANISOTROPIC)
    nt_eng.upload(m)
    nt_eng.set_reparametrization(M.REPAM_ANISOTROPIC)
    nt_eng.compute_pass(3); o.ComputePass(3)
    nt_eng.compute_pass(20); o.ComputePass(20)
    assert np.array_equal(nt_eng.download_duals(), o.duals())
    assert abs(nt_eng.lower_bound() - o.LowerBound()) <= LB_RTOL * max(1.0, abs(o.LowerBound()))
    nt_eng.enable_kernel_timing(True)
    nt_eng.compute_pass(2); o.ComputePass(2)
    kt = nt_eng.kernel_timing()
    nt_eng.enable_kernel_timing(False)
    assert np.array_equal(nt_eng.download_duals(), o.duals())
    (name,) = [v["kernel"] for v in kt.values()]
    assert name.endswith(", true>" if nt_eng.want_nt else ", false>"), name     # the instantiation BENCH names


@pytest.mark.parametrize("blocked", [True, False], ids=["blocked_chain", "launch_per_step"])
def test_hbm_sized_model_against_the_oracle(blocked, monkeypatch):
    """A model above the engine's 1 GiB streaming threshold (384 x 384, 32 labels, dense: 2.4 GB of tables), without
    any kernel override: the engine itself selects what it selects for bench.py's C3 — joined passes as ONE persistent
    chain launch with the Infinity-Cache ticket order (default), or (LPMP_NO_BLOCKED_PASSES=1) one launch per step
    with the NT = true kernels — and the duals after two + three passes equal the oracle's bit for bit."""
    if not blocked:
        monkeypatch.setenv("LPMP_NO_BLOCKED_PASSES", "1")
    H = W = 384; L = 32
    m = S.grid_model(H, W, L, order="colour_major", seed=5)
    o = Oracle(m)
    o.set_reparametrization(M.REPAM_ANISOTROPIC)
    e = E.Engine(0)
    try:
        e.upload(m)
        assert e.L.lpmp_streaming_access(e.h) == 1
        e.set_reparametrization(M.REPAM_ANISOTROPIC)
        e.enable_kernel_timing(True)
        e.compute_pass(2); o.ComputePass(2)
        kt = e.kernel_timing()
        e.enable_kernel_timing(False)
        want = "chain_dense_pk_kernel<32, 2, false, false>" if blocked else "sweep_dense_pk_kernel<32, 2, false, true>"
        assert [v["kernel"] for v in kt.values()] == [want]
        assert np.array_equal(e.download_duals(), o.duals())
        e.compute_pass(3); o.ComputePass(3)
        e.compute_pass(1); o.ComputePass(1)
        assert np.array_equal(e.download_duals(), o.duals())
        lb, lbo = e.lower_bound(), o.LowerBound()
        assert abs(lb - lbo) <= LB_RTOL * abs(lbo)
    finally:
        e.close()


@pytest.mark.parametrize("pairwise,L,H,W", [("dense", 32, 40, 36), ("dense", 8, 60, 70), ("potts", 8, 64, 48), ("dense", 21, 30, 31)])
@pytest.mark.parametrize("bands,lag,depth", [(8, 2, 4), (5, 1, 2), (16, 2, 3), (3, 4, 7)])
def test_joined_passes_as_one_blocked_chain_launch(pairwise, L, H, W, bands, lag, depth, monkeypatch):
    """lpmp_compute_pass(n) on a 2-colour order as ONE persistent launch whose tickets follow the skewed band order
    (engine.cpp, rotation_chain): forced on small models here (LPMP_ROT_BANDS), every n, against the oracle bit for bit.
    An order that would break a dependency is detected on the host and the lag widened."""
    monkeypatch.setenv("LPMP_ROT_BANDS", str(bands)); monkeypatch.setenv("LPMP_ROT_LAG", str(lag)); monkeypatch.setenv("LPMP_ROT_DEPTH", str(depth))
    m = S.grid_model(H, W, L, pairwise=pairwise, order="colour_major", seed=L + bands)
    o = Oracle(m); o.set_reparametrization(M.REPAM_ANISOTROPIC)
    e = E.Engine(0)
    try:
        e.upload(m); e.set_reparametrization(M.REPAM_ANISOTROPIC)
        assert e.plan.pass_rotates(M.REPAM_ANISOTROPIC)
        e.prepare_passes(5)
        cache = {}
        for n in (1, 5, 2, 1, 9, 70, 8, 12, 20, 31, 13, 32):   # 70: slices of 32 + 32 + 6 passes, one launch each
            e.enable_kernel_timing(True)
            e.compute_pass(n); o.ComputePass(n)
            kt = e.kernel_timing(); e.reset_kernel_timing(); e.enable_kernel_timing(False)
            assert all(v["kernel"].startswith("chain_") and v["chain_launches"] == (n + 31) // 32 for v in kt.values()), kt
            assert np.array_equal(e.download_duals(), o.duals()), (n,)
            assert abs(e.lower_bound() - o.LowerBound()) <= 1e-9 * max(1.0, abs(o.LowerBound()))
            cache[n] = e.chain_cache_bytes()
        if depth % 2 == 0:
            # calls of more than 7 passes run from a periodic template (prologue, ONE period, tail; engine.cpp rotation_chain):
            # ticket lists and flags of a launch do not grow with its pass count — 12, 20 and 32 passes (the same tail) added
            # nothing to what 8 had built, 31 and 13 nothing to what 9 had (depth 4; depth 2 has one tail for all)
            assert cache[12] == cache[8] and cache[20] == cache[8], cache
            assert cache[13] == cache[31] == cache[32], cache
        flb = e.factor_lower_bounds()
        assert np.max(np.abs(flb - np.array([o.factor_lower_bound(f) for f in range(m.n_factors)]))) <= DUAL_ATOL
    finally:
        e.close()


def _device_dual_checksums(torch, dual):
    """wrapping sums over the IEEE bit patterns of the packed duals, as tests/golden/make_c3_full.py computes them"""
    b = dual.view(torch.int64)
    w = torch.arange(b.numel(), dtype=torch.int64, device=b.device) * 2 + 1
    return int(b.sum().item()) & (2**64 - 1), int((b * w).sum().item()) & (2**64 - 1)


# ---- every kernel class x every weight mode ---------------------------------------------------
@pytest.mark.parametrize("L", [4, 8, 16, 32])
@pytest.mark.parametrize("order", ["row_major", "colour_major"])
def test_dense_fast_path(eng, L, order):
    m = S.grid_model(13, 11, L, order=order, seed=L)
    for mode in MODES:
        _check(eng, m, mode, 3)
    info = eng.plan.schedule_info(M.FORWARD, M.REPAM_ANISOTROPIC)
    assert info["n_levels"] == (2 if order == "colour_major" else 13 + 11 - 1)


@pytest.mark.parametrize("L", [4, 8, 16, 32])
@pytest.mark.parametrize("order", ["row_major", "colour_major"])
def test_potts_fast_path(eng, L, order):
    m = S.grid_model(12, 15, L, pairwise="potts", order=order, seed=10 + L)
    for mode in MODES:
        _check(eng, m, mode, 3)


def test_potts_ties_and_negative_coupling(eng):
    # equal minima in the message vectors exercise the two-min tie rule; negative diff flips the branch
    H, W, L = 9, 9, 8
    n = H * W
    E_ = len(S.grid_edges(H, W)[0])
    un = np.round(S.u01(n * L, 3) * 3.0) / 3.0               # many exact ties
    diffs = np.where(S.u01(E_, 4) < 0.5, -0.5, 0.75)
    m = S.grid_model(H, W, L, pairwise="potts", unaries=un, potts=diffs)
    _check(eng, m, M.REPAM_ANISOTROPIC, 4)
    _check(eng, m, M.REPAM_UNIFORM, 4)


@pytest.mark.parametrize("L", [2, 3, 5, 7, 12, 33, 64])
def test_odd_label_counts(eng, L):
    # up to 32 labels: the run-time-dims classes of the padded width; above: the streaming class
    _check(eng, S.grid_model(6, 7, L, seed=L), M.REPAM_ANISOTROPIC, 2)
    cls = eng.plan.schedule_classes(M.FORWARD, M.REPAM_ANISOTROPIC)
    want = "dense_big" if L > 32 else "dense_v%d" % (4 if L <= 4 else 8 if L <= 8 else 16 if L <= 16 else 32)
    assert list(cls) == [want] and cls[want] == 42
    _check(eng, S.grid_model(6, 7, L, pairwise="potts", seed=L), M.REPAM_DAMPED_UNIFORM, 2)
    cls = eng.plan.schedule_classes(M.FORWARD, M.REPAM_DAMPED_UNIFORM)
    assert list(cls) == ["dense_big" if L > 32 else want.replace("dense", "potts")]


@pytest.mark.parametrize("L", [33, 47, 64, 65, 100, 130, 200])
@pytest.mark.parametrize("order", ["row_major", "colour_major"])
def test_streaming_dense_kernel_many_labels(eng, L, order):
    m = S.grid_model(5, 6, L, order=order, seed=200 + L)
    for mode in MODES:
        _check(eng, m, mode, 2)
    assert list(eng.plan.schedule_classes(M.FORWARD, M.REPAM_UNIFORM)) == ["dense_big"]


def test_streaming_dense_kernel_rectangular_residual_and_duplicates(eng):
    # tables of very different dims around one hub, a duplicate message, residual sends; 512 labels = the limit
    rng = np.random.default_rng(77)
    dims = [70, 3, 512, 33, 70, 18]
    b = M.ModelBuilder(2, S.mrf_mtypes())
    u = [b.add_vector_factors(0, rng.uniform(0, 1, (1, d)))[0] for d in dims]
    for i, j in ((0, 1), (0, 2), (0, 3), (0, 4), (1, 5), (3, 4), (2, 3)):
        p = b.add_dense_pairwise(1, rng.uniform(0, 1, (1, dims[i], dims[j])))[0]
        b.add_messages(0, u[i], p); b.add_messages(1, u[j], p)
        if (i, j) == (0, 4):
            b.add_messages(0, u[i], p)                      # duplicate message into the same vector
        b.add_relations(u[i], p); b.add_relations(p, u[j])
    m = b.finish()
    for mode in MODES:
        _check(eng, m, mode, 3)
    o = Oracle(m); o.set_reparametrization_type(1); o.set_reparametrization(M.REPAM_ANISOTROPIC)
    eng.upload(m); eng.set_reparametrization_type(1); eng.set_reparametrization(M.REPAM_ANISOTROPIC)
    eng.compute_pass(3); o.ComputePass(3)
    assert np.array_equal(eng.download_duals(), o.duals())
    eng.set_reparametrization_type(0)


@pytest.mark.parametrize("L", [1, 2, 3, 6, 9, 15, 17, 21, 31])
@pytest.mark.parametrize("pairwise", ["dense", "potts"])
@pytest.mark.parametrize("order", ["row_major", "colour_major"])
def test_any_label_count_fast_path(eng, L, pairwise, order):
    m = S.grid_model(11, 13, L, pairwise=pairwise, order=order, seed=100 + L)
    for mode in MODES:
        _check(eng, m, mode, 3)
    cls = eng.plan.schedule_classes(M.BACKWARD, M.REPAM_UNIFORM)
    assert "generic" not in cls and all("_v" in c for c in cls)


def test_rectangular_tables(eng):
    # pairwise factors between variables of different label counts (run-time-dims dense classes)
    b = M.ModelBuilder(2, S.mrf_mtypes())
    rng = np.random.default_rng(5)
    dims = [3, 6, 4, 9, 2]
    u = [b.add_vector_factors(0, rng.uniform(0, 1, (1, d)))[0] for d in dims]
    for i in range(4):
        p = b.add_dense_pairwise(1, rng.uniform(0, 1, (dims[i], dims[i + 1])))[0]
        b.add_messages(0, u[i], p)
        b.add_messages(1, u[i + 1], p)
        b.add_relations(u[i], p)
        b.add_relations(p, u[i + 1])
    m = b.finish()
    for mode in MODES:
        _check(eng, m, mode, 3)


def test_mixed_dense_and_potts_edges(eng):
    H, W, L = 7, 8, 8
    var = S.grid_variable_order(H, W, "row_major").reshape(-1)
    a, bb = S.grid_edges(H, W)
    b = M.ModelBuilder(2, S.mrf_mtypes())
    u = b.add_vector_factors(0, S.u01(H * W * L, 1).reshape(-1, L))
    rng = np.random.default_rng(2)
    for k in range(len(a)):
        if k % 2 == 0:
            p = b.add_dense_pairwise(1, rng.uniform(0, 1, (L, L)))[0]
        else:
            p = b.add_potts_pairwise(1, L, [rng.uniform(0, 1)])[0]
        b.add_messages(0, u[var[a[k]]], p)
        b.add_messages(1, u[var[bb[k]]], p)
        b.add_relations(u[var[a[k]]], p)
        b.add_relations(p, u[var[bb[k]]])
    _check(eng, b.finish(), M.REPAM_ANISOTROPIC, 3)
    cls = eng.plan.schedule_classes(M.FORWARD, M.REPAM_ANISOTROPIC)
    assert set(cls) <= {"dense_v8", "dense8", "potts8"} and cls.get("dense_v8", 0) > 0   # both kinds of edges: packed run-time-dims class
    _check(eng, b.finish(), M.REPAM_DAMPED_UNIFORM, 3)


@pytest.mark.parametrize("L", [33, 64, 100])
def test_streaming_kernel_potts_many_labels_and_ties(eng, L):
    H, W = 6, 5
    E_ = len(S.grid_edges(H, W)[0])
    un = np.round(S.u01(H * W * L, 3) * 3.0) / 3.0               # exact ties exercise the two-min tie rule
    diffs = np.where(S.u01(E_, 4) < 0.5, -0.5, 0.75)
    m = S.grid_model(H, W, L, pairwise="potts", unaries=un, potts=diffs)
    for mode in MODES:
        _check(eng, m, mode, 3)
    assert list(eng.plan.schedule_classes(M.FORWARD, M.REPAM_UNIFORM)) == ["dense_big"]


def test_chain_c1(eng):
    """BASELINE.json configs[0]: 4-label Potts chain, 100 variables."""
    m = S.chain_model(100, 4)
    lb = _check(eng, m, M.REPAM_ANISOTROPIC, 10)
    # a chain is a tree: the LP bound is tight after one forward + backward sweep
    o = Oracle(m)
    o.set_reparametrization(M.REPAM_ANISOTROPIC)
    o.ComputePass(10)
    assert abs(lb - o.LowerBound()) <= 1e-9


def test_random_sparse_graph(eng):
    for mode in (M.REPAM_ANISOTROPIC, M.REPAM_DAMPED_UNIFORM):
        _check(eng, S.random_graph_model(300, 1200, 16, seed=8), mode, 3)


def test_labeling_list_model(eng):
    m = S.multicut_triangle_model(40, 60, seed=9)
    for mode in MODES:
        _check(eng, m, mode, 4)


def test_all_schedules_and_roles(eng):
    from tests.test_plan_host import _full_schedule_model, _toy
    for mode in MODES:
        _check(eng, _full_schedule_model(), mode, 3)
    eng.upload(_toy())
    eng.set_reparametrization(M.REPAM_ANISOTROPIC)
    eng.compute_pass(1000)                                  # reference test/test_model.cpp:43-45
    assert abs(eng.lower_bound() - 1.0) <= 1e-8


def test_reference_known_answers_on_device(eng, golden_dir):
    """the reference's recorded lower bounds reproduced by the HIP path"""
    g = np.load(golden_dir + "/survey_grids.npz")
    c = g["costs_8x8_L4"]
    eng.upload(S.grid_model(8, 8, 4, unaries=c[:256], tables=c[256:]))
    eng.set_reparametrization(M.REPAM_ANISOTROPIC)
    exp = g["lb_8x8_L4_pass0_1_4"]
    assert eng.lower_bound() == pytest.approx(exp[0], abs=1e-9)
    eng.compute_pass(1)
    assert eng.lower_bound() == pytest.approx(exp[1], abs=1e-9)
    eng.compute_pass(3)
    assert eng.lower_bound() == pytest.approx(exp[2], abs=1e-9)
    c = g["costs_16x16_L4"]
    for k, mode in enumerate((M.REPAM_ANISOTROPIC, M.REPAM_UNIFORM, M.REPAM_DAMPED_UNIFORM)):
        eng.upload(S.grid_model(16, 16, 4, unaries=c[:1024], tables=c[1024:]))
        eng.set_reparametrization(mode)
        eng.compute_pass(1)
        assert eng.lower_bound() == pytest.approx(g["lb_16x16_L4_pass1_aniso_uniform_damped"][k], abs=1e-9)
    # test/graphical_model.cpp:90-137: chain and frustrated cycle, LB 0
    from tests.test_oracle_kat import _binary_mrf, NEG, POS
    for m in (_binary_mrf(5, [(0, 1, NEG), (1, 2, POS), (2, 3, POS), (3, 4, POS)]),
              _binary_mrf(4, [(0, 1, NEG), (1, 2, POS), (2, 3, POS), (0, 3, POS)])):
        eng.upload(m)
        eng.set_reparametrization(M.REPAM_ANISOTROPIC)
        eng.compute_pass(100)
        assert abs(eng.lower_bound()) <= 1e-8


def test_custom_pass_and_single_directions(eng):
    m = S.grid_model(9, 8, 8, seed=12)
    o = Oracle(m)
    eng.upload(m)
    o.set_reparametrization(M.REPAM_ANISOTROPIC)
    eng.set_reparametrization(M.REPAM_ANISOTROPIC)
    o.ComputeForwardPass(); eng.forward_pass()
    assert np.array_equal(eng.download_duals(), o.duals())
    o.ComputeBackwardPass(); eng.backward_pass()
    assert np.array_equal(eng.download_duals(), o.duals())
    # iterator-range ComputePass with a partition-style sub-list and its own weights
    order = o.order(M.FORWARD)
    sub = order[: order.shape[0] // 2]
    om_off, om, mk_off, mk = o.anisotropic_weights_sublist(sub)
    upd = np.array([f for f in sub if f in set(o.update_order(M.FORWARD))], np.int32)
    o.compute_pass_custom(upd, om_off, om, mk_off, mk)
    eng.compute_pass_custom(upd, om_off, om, mk_off, mk)
    assert np.array_equal(eng.download_duals(), o.duals())
    # dual upload round trip
    d = o.duals() * 0.5
    eng.upload_duals(d)
    assert np.array_equal(eng.download_duals(), d)


def test_many_levels_graph_replay(eng):
    # 40x30 row-major: 69 dependent levels per direction -> captured into a hipGraph and replayed
    m = S.grid_model(40, 30, 8, seed=13)
    _check(eng, m, M.REPAM_ANISOTROPIC, 4)


@pytest.mark.parametrize("pairwise,L,H,W", [("dense", 32, 96, 80), ("dense", 8, 150, 170), ("dense", 21, 60, 70), ("potts", 8, 200, 150), ("potts", 5, 64, 90)])
def test_chain_executor_deep_schedules(pairwise, L, H, W, monkeypatch):
    """row-major grids: one level per anti-diagonal.  The chain executor runs a whole pass as ONE persistent launch whose
    workgroups wait for their predecessors' completion flags (kernels.hip); it must equal the oracle bit for bit, and
    the same passes as one launch per level (LPMP_NO_CHAIN=1)."""
    m = S.grid_model(H, W, L, pairwise=pairwise, order="row_major", seed=L + H)
    o = Oracle(m)
    engines = []
    for env in ("0", "1"):
        monkeypatch.setenv("LPMP_NO_CHAIN", env)
        e = E.Engine(0); e.upload(m); engines.append(e)
    try:
        for mode in (M.REPAM_ANISOTROPIC, M.REPAM_DAMPED_UNIFORM):
            o.set_reparametrization(mode)
            for e in engines:
                e.set_reparametrization(mode)
            for n in (1, 3):
                o.ComputePass(n)
                for e in engines:
                    e.compute_pass(n)
                    assert np.array_equal(e.download_duals(), o.duals()), (mode, n)
                    assert abs(e.lower_bound() - o.LowerBound()) <= LB_RTOL * max(1.0, abs(o.LowerBound()))
            o.ComputeForwardPass(); o.ComputeBackwardPass()
            for e in engines:
                e.forward_pass(); e.backward_pass()
                assert np.array_equal(e.download_duals(), o.duals())
        # residual sends run in the chain executor too
        o.set_reparametrization_type(1)
        for e in engines:
            e.set_reparametrization_type(1)
        o.ComputePass(2)
        for e in engines:
            e.compute_pass(2)
            assert np.array_equal(e.download_duals(), o.duals())
    finally:
        for e in engines:
            e.close()


@pytest.mark.parametrize("pairwise,L,H,W,band_bytes", [("dense", 8, 40, 50, 6000), ("dense", 32, 10, 12, 40000), ("dense", 5, 40, 36, 3000),
                                                        ("dense", 16, 16, 16, 100000), ("dense", 21, 9, 14, 20000)])
def test_few_big_launches_run_as_a_banded_chain(pairwise, L, H, W, band_bytes, monkeypatch):
    """the colour steps of an HBM-sized dense grid — a directional sweep, a fused pass in a mode that does not rotate, a
    fused custom schedule like the per-pass schedule of a multi-GPU part — run as one chain launch in Infinity-Cache
    order (plan.cpp make_schedule).  Forced onto small models here (LPMP_BAND_MIN_BYTES / LPMP_BAND_BYTES); bit for bit
    against the oracle"""
    from lp_mp_amd.multi_gpu import _cat_rows
    monkeypatch.setenv("LPMP_BAND_MIN_BYTES", "1000"); monkeypatch.setenv("LPMP_BAND_BYTES", str(band_bytes))
    m = S.grid_model(H, W, L, pairwise=pairwise, order="colour_major", seed=L + W)
    o = Oracle(m)
    e = E.Engine(0); e.upload(m)
    try:
        banded = 0
        for mode in MODES:
            o.set_reparametrization(mode); e.set_reparametrization(mode)
            banded += e.plan.chain_info(M.FORWARD, mode)["n_chains"] + e.plan.chain_info(-1, mode)["n_chains"]
            for step in range(2):
                e.forward_pass(); o.ComputeForwardPass()
                assert np.array_equal(e.download_duals(), o.duals()), (mode, "forward")
                e.backward_pass(); o.ComputeBackwardPass()
                assert np.array_equal(e.download_duals(), o.duals()), (mode, "backward")
                e.compute_pass(1); o.ComputePass(1)
                assert np.array_equal(e.download_duals(), o.duals()), (mode, "pass")
                assert abs(e.lower_bound() - o.LowerBound()) <= LB_RTOL * max(1.0, abs(o.LowerBound()))
            rows = []
            for d in (M.FORWARD, M.BACKWARD, M.FORWARD):
                oo, om = o.omega(d, mode); mo, mk = o.mask(d, mode)
                rows.append((o.update_order(d), oo, om, mo, mk))
            cat = _cat_rows(*rows)
            sid = e.schedule_create(*cat, fuse=True)
            for _ in range(2):
                e.schedule_run(sid); o.compute_pass_custom(*cat)
                assert np.array_equal(e.download_duals(), o.duals()), (mode, "custom")
            e.schedule_destroy(sid)
        # (the run-time-dims classes are not banded: slower that way, engine.cpp plan_rotation_chain)
        assert banded >= 4 if L in (4, 8, 16, 32) else banded == 0
    finally:
        e.close()


def test_tiny_levels_of_the_lane_per_factor_class_three_ways(monkeypatch):
    """178 levels of labeling-list factors (C5 with local triples, in miniature): by default ONE workgroup walks the
    levels (level_loop_kernel); LPMP_NO_LEVEL_LOOP=1 replays a hipGraph of launches; LPMP_CHAIN_ALL=1 runs the ticket
    form of the generic chain kernel.  All three against the oracle bit for bit, residual sends included"""
    m = S.c5_model(24, 24, 8, 400, 300, 100, seed=5, window=16)
    o = Oracle(m)
    engines = []
    e = E.Engine(0); e.upload(m); engines.append(e)
    ci = e.plan.chain_info(M.BACKWARD, M.REPAM_ANISOTROPIC)
    assert ci["n_chains"] == 1 and ci["n_tickets"] == 0
    monkeypatch.setenv("LPMP_NO_LEVEL_LOOP", "1")
    e = E.Engine(0); e.upload(m); engines.append(e)
    assert e.plan.chain_info(M.BACKWARD, M.REPAM_ANISOTROPIC)["n_chains"] == 0
    monkeypatch.delenv("LPMP_NO_LEVEL_LOOP"); monkeypatch.setenv("LPMP_CHAIN_ALL", "1")
    e = E.Engine(0); e.upload(m); engines.append(e)
    assert e.plan.chain_info(M.BACKWARD, M.REPAM_ANISOTROPIC)["n_tickets"] > 0
    try:
        for rtype in (0, 1):
            o.set_reparametrization_type(rtype)
            for e in engines:
                e.set_reparametrization_type(rtype)
            for mode in (M.REPAM_ANISOTROPIC, M.REPAM_UNIFORM):
                o.set_reparametrization(mode)
                for e in engines:
                    e.set_reparametrization(mode)
                o.ComputePass(2); o.ComputeBackwardPass()
                for e in engines:
                    e.compute_pass(2); e.backward_pass()
                    assert np.array_equal(e.download_duals(), o.duals()), (rtype, mode)
                    assert abs(e.lower_bound() - o.LowerBound()) <= LB_RTOL * max(1.0, abs(o.LowerBound()))
    finally:
        for e in engines:
            e.close()


def test_chain_executor_repeated_runs_are_deterministic():
    """the flags of a chain run carry the run's epoch: many runs back to back on one schedule, every result equal to
    the one-launch-per-level path"""
    import torch
    H = W = 192; L = 32
    res = {}
    for env in ("0", "1"):
        os.environ["LPMP_NO_CHAIN"] = env
        try:
            m, const, dual, n, n_e = _device_grid(torch, H, W, L, "row_major", 9)
            e = E.Engine(0)
            e.upload(m, const_dev=const.data_ptr(), dual_dev=dual.data_ptr(), keep=(const, dual))
            e.set_reparametrization(M.REPAM_ANISOTROPIC)
            sums = []
            for _ in range(12):
                e.compute_pass(1)
                e.synchronize()
                sums.append(_device_dual_checksums(torch, dual))
            res[env] = sums
            e.close()
        finally:
            os.environ.pop("LPMP_NO_CHAIN")
    assert res["0"] == res["1"]


def test_error_paths(eng):
    e2 = E.Engine(0)
    with pytest.raises(E.EngineError):
        e2.compute_pass(1)                      # no model
    e2.upload(S.grid_model(3, 3, 2))
    with pytest.raises(E.EngineError):
        e2.compute_pass(1)                      # no reparametrization mode set (reference LP_MP.h:458)
    with pytest.raises(E.EngineError):
        e2.set_reparametrization(M.REPAM_MIXED)
    e2.close()


# ---- device-resident inputs and full-size properties -----------------------------------------
def _energy(torch, theta, pw, T, ei, ej, x):
    """E(x) = sum_i theta_i(x_i) + sum_ij T_ij(x_i,x_j) + m1_ij(x_i) + m2_ij(x_j): invariant under any
    reparametrisation (messages only move cost between factors)."""
    n, L = theta.shape
    ar_n = torch.arange(n, device=theta.device)
    ar_e = torch.arange(ei.shape[0], device=theta.device)
    xi, xj = x[ei], x[ej]
    return (theta[ar_n, x].sum() + T[ar_e, xi, xj].sum() + pw[ar_e, xi].sum() + pw[ar_e, L + xj].sum()).item()


def _device_grid(torch, H, W, L, order, seed):
    m = S.grid_model(H, W, L, order=order, seed=seed, device_const=True)
    n = H * W
    n_e = len(S.grid_edges(H, W)[0])
    dev = torch.device("cuda:0")
    const = torch.empty(n_e * L * L, dtype=torch.float64, device=dev)
    E.synth_fill(const.data_ptr(), const.numel(), seed, n * L, torch.cuda.current_stream().cuda_stream)
    dual = torch.zeros(n * L + n_e * 2 * L, dtype=torch.float64, device=dev)
    dual[: n * L] = torch.from_numpy(m.dual_data[: n * L]).to(dev)
    torch.cuda.synchronize()
    return m, const, dual, n, n_e


def test_device_buffers_and_synth_fill_match_host(eng):
    import torch
    H, W, L = 20, 24, 16
    m, const, dual, n, n_e = _device_grid(torch, H, W, L, "colour_major", 21)
    host = S.grid_model(H, W, L, order="colour_major", seed=21)
    assert np.array_equal(const.cpu().numpy(), host.const_data)           # device generator == numpy generator
    e2 = E.Engine(0)
    e2.upload(m, const_dev=const.data_ptr(), dual_dev=dual.data_ptr(), keep=(const, dual))
    e2.set_reparametrization(M.REPAM_ANISOTROPIC)
    e2.compute_pass(3)
    e2.synchronize()
    o = Oracle(host)
    o.set_reparametrization(M.REPAM_ANISOTROPIC)
    o.ComputePass(3)
    assert np.array_equal(dual.cpu().numpy(), o.duals())                  # updated in place, zero copy
    e2.close()


@pytest.mark.parametrize("cfg", ["C2_512x512_L8_potts", "C3_1024x1024_L32_dense", "C3_1024x1024_L32_dense_launch_per_step"])
def test_full_size_properties(cfg, monkeypatch):
    """BASELINE.json configs[1] and [2] at full size: dual ascent (LB non-decreasing), energy of fixed
    labelings invariant under the sweep, LB <= energy of any labeling, and the oracle where it is cheap."""
    import torch
    if cfg.endswith("launch_per_step"):
        monkeypatch.setenv("LPMP_NO_BLOCKED_PASSES", "1")
    e2 = E.Engine(0)
    dev = torch.device("cuda:0")
    if cfg.startswith("C2"):
        H = W = 512; L = 8
        m = S.grid_model(H, W, L, pairwise="potts", order="colour_major", seed=2)
        e2.upload(m)
        e2.set_reparametrization(M.REPAM_ANISOTROPIC)
        o = Oracle(m)
        o.set_reparametrization(M.REPAM_ANISOTROPIC)
        lbs = [e2.lower_bound()]
        for _ in range(3):
            e2.compute_pass(1); o.ComputePass(1)
            lbs.append(e2.lower_bound())
            assert abs(lbs[-1] - o.LowerBound()) <= LB_RTOL * abs(o.LowerBound())
        assert np.array_equal(e2.download_duals(), o.duals())
        assert all(b >= a - 1e-7 for a, b in zip(lbs, lbs[1:]))
    else:
        H = W = 1024; L = 32
        m, const, dual, n, n_e = _device_grid(torch, H, W, L, "colour_major", 3)
        e2.upload(m, const_dev=const.data_ptr(), dual_dev=dual.data_ptr(), keep=(const, dual))
        e2.set_reparametrization(M.REPAM_ANISOTROPIC)
        var = S.grid_variable_order(H, W, "colour_major").reshape(-1)
        a, b = S.grid_edges(H, W)
        ei = torch.from_numpy(np.minimum(var[a], var[b])).to(dev)
        ej = torch.from_numpy(np.maximum(var[a], var[b])).to(dev)
        T = const.view(n_e, L, L)
        gen = torch.Generator(device="cpu").manual_seed(0)
        xs = [torch.randint(0, L, (n,), generator=gen).to(dev) for _ in range(2)]
        xs.append(dual[: n * L].view(n, L).argmin(1))          # unary-greedy labeling
        def energies():
            th = dual[: n * L].view(n, L)
            pw = dual[n * L:].view(n_e, 2 * L)
            return [_energy(torch, th, pw, T, ei, ej, x) for x in xs]
        e0 = energies()
        # the oracle ran ONCE on exactly this model in the build container (tests/golden/make_c3_full.py, seed 3): lower
        # bound after 0..3 passes and exact checksums of the packed duals.  This run uses the kernel instantiation
        # BENCH names: the joined passes as one persistent chain launch (chain_dense_pk_kernel<32, 2, false, false>) or,
        # in the launch-per-step variant, sweep_dense_pk_kernel<32, 2, false, true> (tables + duals > 1 GiB).
        g = np.load(os.path.join(os.path.dirname(__file__), "golden", "c3_full_lb.npz"))
        assert (int(g["H"]), int(g["W"]), int(g["L"])) == (H, W, L) and list(g["passes_seed3"]) == [0, 1, 2, 3]
        assert e2.L.lpmp_streaming_access(e2.h) == 1
        lbs = [e2.lower_bound()]
        for k in range(4):
            if k:
                e2.compute_pass(1)
                lbs.append(e2.lower_bound())
            e2.synchronize()
            assert abs(lbs[-1] - g["lb_seed3"][k]) <= LB_RTOL * abs(g["lb_seed3"][k]), (k, lbs[-1], g["lb_seed3"][k])
            assert abs(lbs[-1] - g["lb_seed3"][k]) <= 1e-9 * abs(g["lb_seed3"][k])          # only the summation order differs
            c0, c1 = _device_dual_checksums(torch, dual)
            assert (c0, c1) == (int(g["dual_sum_seed3"][k]), int(g["dual_wsum_seed3"][k])), ("duals differ from the oracle's after pass", k)
        e2.enable_kernel_timing(True)
        e2.compute_pass(2)
        kt = e2.kernel_timing()
        e2.enable_kernel_timing(False)
        assert [v["kernel"] for v in kt.values()] == ["sweep_dense_pk_kernel<32, 2, false, true>" if cfg.endswith("launch_per_step")
                                                     else "chain_dense_pk_kernel<32, 2, false, false>"]
        e2.synchronize()
        e1 = energies()
        for x0, x1 in zip(e0, e1):
            assert abs(x0 - x1) <= 1e-9 * abs(x0)                # reparametrisation invariance
        assert all(b >= a - 1e-7 * abs(a) for a, b in zip(lbs, lbs[1:]))
        assert lbs[-1] <= min(e1) + 1e-6                          # weak duality
        assert lbs[-1] > lbs[0]
    e2.close()


def test_what_bench_times_against_the_oracle_at_full_size():
    """The driver's BENCH command (--warmup 5 --steps 20) on the exact C3 inputs (seed 1): 5 passes in one call, then 20
    passes as ONE persistent chain launch with the automatic band / lag choice of the 1024 x 1024 model — duals and bound
    against the oracle's run on the same inputs (tests/golden/c3_full_lb.npz holds every pass count 0 ... 48) at 5 and at 25
    passes; then four more passes the way a Solve loop asks for them (one pass + the bound per call, passes running ahead),
    checked pass by pass."""
    import torch
    H = W = 1024; L = 32
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "c3_full_lb.npz"))
    assert list(g["passes_seed1"]) == list(range(49))
    m, const, dual, n, n_e = _device_grid(torch, H, W, L, "colour_major", 1)
    e = E.Engine(0)
    try:
        e.set_stream(torch.cuda.current_stream().cuda_stream)
        e.upload(m, const_dev=const.data_ptr(), dual_dev=dual.data_ptr(), keep=(const, dual))
        e.set_reparametrization(M.REPAM_ANISOTROPIC)
        def check(k, lb):
            e.synchronize()
            assert abs(lb - g["lb_seed1"][k]) <= 1e-9 * abs(g["lb_seed1"][k]), (k, lb, g["lb_seed1"][k])
            assert _device_dual_checksums(torch, dual) == (int(g["dual_sum_seed1"][k]), int(g["dual_wsum_seed1"][k])), ("duals differ from the oracle's after pass", k)
        check(0, e.lower_bound())
        e.compute_pass(5)
        check(5, e.lower_bound())
        e.enable_kernel_timing(True)
        e.compute_pass(20)
        kt = e.kernel_timing()
        e.enable_kernel_timing(False)
        assert [(v["kernel"], v["launches"], v["chain_launches"]) for v in kt.values()] == [("chain_dense_pk_kernel<32, 2, false, false>", 1, 1)]
        check(25, e.lower_bound())
        e.set_speculation(8)
        lbs = []
        for k in range(26, 30):
            e.compute_pass(1)
            lbs.append(e.lower_bound())           # the bound of pass k, while the device may already be further
        for k, lb in zip(range(26, 30), lbs):
            assert abs(lb - g["lb_seed1"][k]) <= 1e-9 * abs(g["lb_seed1"][k]), (k, lb)
        st = e.speculation_stats()
        assert st["batches"] == 2 and st["passes_used"] == 4 and st["passes_launched"] == 6, st     # batches of 2 and 4
        check(29, lbs[-1])                         # synchronize settles: two passes of the second batch are rolled back
        assert e.speculation_stats()["rollbacks"] == 1
    finally:
        e.close()


@pytest.mark.parametrize("pairwise,L", [("dense", 32), ("dense", 8), ("potts", 16)])
@pytest.mark.parametrize("order", ["colour_major", "row_major"])
def test_multi_pass_call_fused_and_rotated_schedules(eng, pairwise, L, order):
    """lpmp_compute_pass(n >= 2): forward+backward fused into one sequence, and on 2-colour orders the tail of
    a pass joined with the head of the next one; must equal n sequential reference passes exactly."""
    m = S.grid_model(14, 10, L, pairwise=pairwise, order=order, seed=31)
    for mode in MODES:
        o = Oracle(m)
        o.set_reparametrization(mode)
        eng.upload(m)
        eng.set_reparametrization(mode)
        eng.compute_pass(5); o.ComputePass(5)
        assert np.array_equal(eng.download_duals(), o.duals())
        eng.compute_pass(2); o.ComputePass(2)
        eng.compute_pass(1); o.ComputePass(1)
        assert np.array_equal(eng.download_duals(), o.duals())
    info = eng.plan.pass_schedule_info(M.REPAM_ANISOTROPIC)
    assert info["n_levels"] == (3 if order == "colour_major" else 2 * (14 + 10 - 1) - 1)


def test_residual_send_rule(eng):
    """--reparametrizationType residual (reference update_factor_residual, factors_messages.hxx:2270-2279,2960-3007)"""
    from tests.test_plan_host import _full_schedule_model
    models = [S.grid_model(9, 7, 32, order="colour_major", seed=41), S.grid_model(8, 9, 8, order="row_major", seed=42),
              S.grid_model(7, 8, 16, pairwise="potts", seed=43), S.grid_model(5, 6, 5, seed=44),
              S.multicut_triangle_model(25, 30, seed=45), _full_schedule_model()]
    try:
        for m in models:
            for mode in (M.REPAM_ANISOTROPIC, M.REPAM_DAMPED_UNIFORM):
                o = Oracle(m)
                o.set_reparametrization_type(1)
                o.set_reparametrization(mode)
                eng.upload(m)
                eng.set_reparametrization_type(1)
                eng.set_reparametrization(mode)
                eng.compute_pass(3); o.ComputePass(3)
                eng.compute_pass(1); o.ComputePass(1)
                assert np.array_equal(eng.download_duals(), o.duals())
                assert abs(eng.lower_bound() - o.LowerBound()) <= LB_RTOL * max(1.0, abs(o.LowerBound()))
        with pytest.raises(E.EngineError):
            eng.set_reparametrization_type(4)      # adaptive: needs op-level improvement functions, not on the device
    finally:
        eng.set_reparametrization_type(0)


def test_c5_style_mixed_model(eng):
    """BASELINE.json configs[4] in miniature: grid pairwise MRF plus labeling-list higher-order factors of mixed
    arity (edge factors with 1 labeling, triplets with 4, quadruple-style factors with 7) in ONE factor graph:
    several kernel classes inside the same level."""
    H, W, L = 8, 9, 8
    mt = S.mrf_mtypes() + [M.MsgType(2, 3, M.SCHED_LEFT, 0, 1, M.M_LABELING, k) for k in range(3)] + \
        [M.MsgType(2, 4, M.SCHED_LEFT, 0, 1, M.M_LABELING, 3 + k) for k in range(4)]
    b = M.ModelBuilder(5, mt)
    quad = [(1, 1, 0, 0), (0, 1, 1, 0), (0, 0, 1, 1), (1, 0, 0, 1), (1, 1, 1, 1), (1, 0, 1, 0), (0, 1, 0, 1)]
    for k in range(3):
        b.add_labeling_table(S.EDGE_LABELINGS, S.TRIPLET_LABELINGS, (k,))
    for k in range(4):
        b.add_labeling_table(S.EDGE_LABELINGS, quad, (k,))
    var = S.grid_variable_order(H, W, "colour_major").reshape(-1)
    a, bb = S.grid_edges(H, W)
    u = b.add_vector_factors(0, S.u01(H * W * L, 5).reshape(-1, L))
    p = b.add_potts_pairwise(1, L, S.u01(len(a), 6))
    i, j = np.minimum(var[a], var[bb]), np.maximum(var[a], var[bb])
    b.add_interleaved_messages(np.tile(np.array([0, 1], np.int32), len(a)), np.stack([u[i], u[j]], 1).reshape(-1), np.repeat(p, 2))
    b.add_relations(np.stack([u[i], p], 1).reshape(-1), np.stack([p, u[j]], 1).reshape(-1))
    rng = np.random.default_rng(3)
    e = b.add_vector_factors(2, (2 * S.u01(40, 7) - 1).reshape(-1, 1), implicit_origin=True)
    for _ in range(25):
        t = b.add_vector_factors(3, np.zeros((1, 4)), implicit_origin=True)[0]
        for k, ei in enumerate(rng.choice(40, 3, replace=False)):
            b.add_messages(2 + k, e[ei], t); b.add_relations(e[ei], t)
    for _ in range(10):
        q = b.add_vector_factors(4, np.zeros((1, 7)), implicit_origin=True)[0]
        for k, ei in enumerate(rng.choice(40, 4, replace=False)):
            b.add_messages(5 + k, e[ei], q); b.add_relations(e[ei], q)
    m = b.finish()
    for mode in MODES:
        _check(eng, m, mode, 4)


def test_tracked_lower_bounds_equal_recomputed_ones():
    """lpmp_lower_bound after a pass sums bounds the sweep kernels tracked as a by-product; an engine with
    LPMP_NO_LB_TRACKING=1 recomputes every factor from memory.  Both must agree with the oracle per factor."""
    import os
    for m in (S.grid_model(16, 12, 32, order="colour_major", seed=51), S.grid_model(11, 13, 8, pairwise="potts", seed=52),
              S.grid_model(9, 10, 16, order="row_major", seed=53)):
        engines = []
        for env in ("0", "1"):
            os.environ["LPMP_NO_LB_TRACKING"] = env
            e = E.Engine(0)
            e.upload(m)
            engines.append(e)
        os.environ.pop("LPMP_NO_LB_TRACKING")
        o = Oracle(m)
        for mode in (M.REPAM_ANISOTROPIC, M.REPAM_UNIFORM):
            o.set_reparametrization(mode)
            for e in engines:
                e.set_reparametrization(mode)
            for step in (1, 3, 1):
                o.ComputePass(step)
                ref = np.array([o.factor_lower_bound(f) for f in range(m.n_factors)])
                for e in engines:
                    e.compute_pass(step)
                    assert abs(e.lower_bound() - o.LowerBound()) <= 1e-9 * max(1.0, abs(o.LowerBound()))
                    assert np.max(np.abs(e.factor_lower_bounds() - ref)) <= DUAL_ATOL
            o.ComputeForwardPass()
            for e in engines:
                e.forward_pass()
                assert abs(e.lower_bound() - o.LowerBound()) <= 1e-9 * max(1.0, abs(o.LowerBound()))
        d = o.duals() * 0.75
        o.set_duals(d)
        for e in engines:
            e.upload_duals(d)
            assert abs(e.lower_bound() - o.LowerBound()) <= 1e-9 * max(1.0, abs(o.LowerBound()))
            e.close()


def test_size_limits_isolated_factors_and_empty_schedules(eng):
    # largest factors the generic kernel takes: 512 doubles of dual (256-label pairwise, 512-label unary)
    _check(eng, S.grid_model(2, 3, 256, pairwise="potts", seed=61), M.REPAM_ANISOTROPIC, 2)
    _check(eng, S.grid_model(2, 2, 200, seed=62), M.REPAM_UNIFORM, 1)
    # beyond it: refused loudly, not silently computed elsewhere
    big = M.ModelBuilder(1, [M.MsgType(0, 0, M.SCHED_LEFT, 0, 0, M.M_MINNORM, 0)])
    f = big.add_vector_factors(0, np.zeros((2, 600)))
    big.add_messages(0, f[0], f[1])
    eng.upload(big.finish())
    with pytest.raises(E.EngineError) as ei:
        eng.set_reparametrization(M.REPAM_ANISOTROPIC)
    assert ei.value.code == -2                                   # LPMP_ERR_UNSUPPORTED
    # factors without messages: never updated, still counted in the bound; a model without messages: passes are no-ops
    b = M.ModelBuilder(2, S.mrf_mtypes())
    u = b.add_vector_factors(0, np.array([[3.0, 1.0, 2.0], [5.0, 4.0, 6.0], [0.5, 0.7, 0.2], [9.0, 8.0, 7.0]]))
    pw = b.add_dense_pairwise(1, np.arange(9.0).reshape(1, 3, 3))
    b.add_messages(0, u[0], pw[0]); b.add_messages(1, u[2], pw[0])
    b.add_relations(u[0], pw[0]); b.add_relations(pw[0], u[2])
    _check(eng, b.finish(), M.REPAM_ANISOTROPIC, 3)
    b = M.ModelBuilder(1, [])
    b.add_vector_factors(0, np.array([[3.0, 1.0], [5.0, 4.0]]))
    b.constant = 2.5
    eng.upload(b.finish())
    eng.set_reparametrization(M.REPAM_DAMPED_UNIFORM)
    eng.compute_pass(3)
    assert eng.lower_bound() == 2.5 + 1.0 + 4.0                  # constant_ + sum of minima (LP_MP.h:1510)
    eng.compute_pass_custom(np.zeros(0, np.int32), [0], [], [0], [])   # empty iterator range
    assert eng.lower_bound() == 7.5


def test_duplicate_messages_into_one_vector(eng):
    """two messages of one type between the same unary and pairwise factor (both sides variable count): both
    receives rewrite the same vector, so the kernels that request several receives at once must step aside"""
    mt = [M.MsgType(0, 1, M.SCHED_LEFT, 0, 0, M.M_UNARY_PAIRWISE, 0), M.MsgType(0, 1, M.SCHED_LEFT, 0, 0, M.M_UNARY_PAIRWISE, 1)]
    for L in (8, 32, 5):
        rng = np.random.default_rng(L)
        b = M.ModelBuilder(2, mt)
        u = b.add_vector_factors(0, rng.uniform(0, 1, (6, L)))
        for k in range(5):
            p = b.add_dense_pairwise(1, rng.uniform(0, 1, (1, L, L)))[0]
            b.add_messages(0, u[k], p); b.add_messages(1, u[k + 1], p)
            b.add_messages(0, u[k], p)                      # duplicate
            b.add_relations(u[k], p); b.add_relations(p, u[k + 1])
        m = b.finish()
        for mode in MODES:
            _check(eng, m, mode, 3)


@pytest.mark.parametrize("window,coloured", [(64, False), (64, True), (150000, False)])
def test_c5_full_size(window, coloured):
    """BASELINE.json configs[4] at full size on one GPU: 512 x 512 Potts grid + 100 k labeling-list factors of arity
    3 and 4 over 150 k binary edge variables; duals bit-identical to the oracle.  Local triples / quads (window 64)
    chain the edge variables into ~20 k dependent steps per backward sweep (hipGraph replay); global ones into ~10."""
    m = S.c5_model(512, 512, 8, 150000, 70000, 30000, seed=4, window=window, colour_edge_vars=coloured)
    o = Oracle(m); o.set_reparametrization(M.REPAM_ANISOTROPIC)
    e = E.Engine(0)
    try:
        e.upload(m); e.set_reparametrization(M.REPAM_ANISOTROPIC)
        lbs = [e.lower_bound()]
        assert abs(lbs[0] - o.LowerBound()) <= LB_RTOL * max(1.0, abs(lbs[0]))
        for _ in range(2):
            e.compute_pass(1); o.ComputePass(1)
            lbs.append(e.lower_bound())
            assert abs(lbs[-1] - o.LowerBound()) <= LB_RTOL * max(1.0, abs(lbs[-1]))
        assert np.array_equal(e.download_duals(), o.duals())
        assert lbs[0] <= lbs[1] + 1e-7 and lbs[1] <= lbs[2] + 1e-7
        cls = e.plan.schedule_classes(M.BACKWARD, M.REPAM_ANISOTROPIC)
        assert cls == {"potts8": 262144, "small": cls["small"]} and cls["small"] > 100000
        n_levels = e.plan.schedule_info(M.BACKWARD, M.REPAM_ANISOTROPIC)["n_levels"]
        assert (n_levels > 5000) == (window == 64 and not coloured)
        if coloured:
            assert n_levels <= 16                                # one level per colour of the edge variables
    finally:
        e.close()


def test_c4_shape_mid_size():
    """BASELINE.json configs[3]'s shape (random sparse graph, 16 labels, dense tables, mean degree 10) at 1/40 of
    the size, where the oracle still finishes in seconds: duals bit-identical after two passes"""
    m = S.random_graph_model(50000, 250000, 16, seed=3)
    o = Oracle(m); o.set_reparametrization(M.REPAM_ANISOTROPIC)
    e = E.Engine(0)
    try:
        e.upload(m); e.set_reparametrization(M.REPAM_ANISOTROPIC)
        e.compute_pass(2); o.ComputePass(2)
        assert np.array_equal(e.download_duals(), o.duals())
        assert abs(e.lower_bound() - o.LowerBound()) <= LB_RTOL * abs(o.LowerBound())
    finally:
        e.close()


@pytest.mark.parametrize("pairwise", ["dense", "potts"])
def test_hub_variable_beyond_the_packet_slab(eng, pairwise):
    """a variable with more active messages than the packed kernels' LDS slab holds (C4 has a few among 2 M): that record
    runs on the streaming kernel, every other record of its launch stays packed; duals bit-identical in all weight modes"""
    from tests.test_plan_host import _hub_model
    for n_spokes in (40, 70):
        m = _hub_model(16, n_spokes, pairwise, seed=n_spokes)
        for mode in MODES:
            _check(eng, m, mode, 3)


def test_c4_full_size_properties():
    """BASELINE.json configs[3] at FULL size on one GPU: random sparse graph, 2 M unaries / 10 M dense 16 x 16 pairwise
    factors (20.5 GB of tables generated in HBM by the per-rank generator bench.py --workload c4 uses, here with one
    part).  Too big for the oracle, so the size-independent properties: the bound never decreases, a sweep only
    reparametrises (the energy of fixed labelings is unchanged), weak duality; and the generator equals the host
    generator on a sample of tables."""
    import torch
    from lp_mp_amd import multi_gpu as MG
    n, m, L = 2_000_000, 10_000_000, 16
    torch.cuda.set_device(0)
    sw = MG.GraphSweep(torch, None, n, m, L, M.REPAM_ANISOTROPIC, seed=1)
    try:
        dev = sw.dualt.device
        ei, ej = S.counter_graph_edges(n, m, 1)
        T = sw.const[: m * L * L].view(m, L, L)
        for e in (0, 12345, m - 1):                                      # table of edge e = stream block at n L + e L^2
            assert np.array_equal(T[e].cpu().numpy().reshape(-1), S.u01(L * L, 1, n * L + e * L * L))
        ei_t, ej_t = torch.from_numpy(ei).to(dev), torch.from_numpy(ej).to(dev)
        gen = torch.Generator(device="cpu").manual_seed(0)
        xs = [torch.randint(0, L, (n,), generator=gen).to(dev), sw.dualt[: n * L].view(n, L).argmin(1)]

        def energies():
            th = sw.dualt[: n * L].view(n, L)
            pw = sw.dualt[n * L:].view(m, 2 * L)
            return [_energy(torch, th, pw, T, ei_t, ej_t, x) for x in xs]
        e0 = energies()
        lbs = [sw.lower_bound()]
        for _ in range(3):
            sw.compute_pass(1)
            lbs.append(sw.lower_bound())
        sw.engine.synchronize()
        e1 = energies()
        for a, b in zip(e0, e1):
            assert abs(a - b) <= 1e-9 * abs(a)
        assert all(b >= a - 1e-7 * abs(a) for a, b in zip(lbs, lbs[1:])) and lbs[-1] > lbs[0]
        assert lbs[-1] <= min(e1) + 1e-6
        cls = sw.engine.plan.schedule_classes(M.FORWARD, M.REPAM_ANISOTROPIC)
        assert sum(cls.values()) > 0 and "generic" not in cls
    finally:
        sw.engine.close()


def _scheduled_grid(H, W, L, sched, seed, order="colour_major", dims=None):
    """grid MRF whose unary-pairwise messages have the given schedule (right / full: the pairwise factors are updated)"""
    mt = [M.MsgType(0, 1, sched, 0, 1, M.M_UNARY_PAIRWISE, 0), M.MsgType(0, 1, sched, 0, 1, M.M_UNARY_PAIRWISE, 1)]
    b = M.ModelBuilder(2, mt)
    n = H * W
    var = S.grid_variable_order(H, W, order).reshape(-1)
    a, bb = S.grid_edges(H, W)
    i, j = np.minimum(var[a], var[bb]), np.maximum(var[a], var[bb])
    rng = np.random.default_rng(seed)
    d = np.full(n, L) if dims is None else rng.choice(dims, size=n)
    u = np.array([b.add_vector_factors(0, rng.uniform(0, 1, (1, int(x))))[0] for x in d])
    p = np.array([b.add_dense_pairwise(1, rng.uniform(0, 1, (1, int(d[x]), int(d[y]))))[0] for x, y in zip(i, j)])
    b.add_interleaved_messages(np.tile(np.array([0, 1], np.int32), len(a)), np.stack([u[i], u[j]], 1).reshape(-1), np.repeat(p, 2))
    b.add_relations(np.stack([u[i], p], 1).reshape(-1), np.stack([p, u[j]], 1).reshape(-1))
    return b.finish()


@pytest.mark.parametrize("sched", [M.SCHED_RIGHT, M.SCHED_FULL])
@pytest.mark.parametrize("L", [5, 8, 16, 21, 32])
def test_updated_pairwise_factors_packed_kernel(eng, sched, L):
    m = _scheduled_grid(7, 6, L, sched, seed=L)
    for mode in MODES:
        _check(eng, m, mode, 3)
    cls = eng.plan.schedule_classes(M.BACKWARD, M.REPAM_UNIFORM)
    want = "pairwise%d" % (8 if L <= 8 else 16 if L <= 16 else 32)
    assert cls.get(want, 0) > 0 and "generic" not in cls
    # residual sends recompute the min-marginals after every send: those launches run on the generic kernel
    o = Oracle(m); o.set_reparametrization_type(1); o.set_reparametrization(M.REPAM_ANISOTROPIC)
    eng.upload(m); eng.set_reparametrization_type(1); eng.set_reparametrization(M.REPAM_ANISOTROPIC)
    eng.compute_pass(2); o.ComputePass(2)
    assert np.array_equal(eng.download_duals(), o.duals())
    eng.set_reparametrization_type(0)


@pytest.mark.parametrize("L", [5, 8, 21])
def test_updated_potts_factors_packed_kernel(eng, L):
    # MPLP-style schedule on Potts tables, incl. ties and negative couplings
    mt = [M.MsgType(0, 1, M.SCHED_RIGHT, 0, 1, M.M_UNARY_PAIRWISE, 0), M.MsgType(0, 1, M.SCHED_RIGHT, 0, 1, M.M_UNARY_PAIRWISE, 1)]
    b = M.ModelBuilder(2, mt)
    H, W = 6, 7
    a, bb = S.grid_edges(H, W)
    u = b.add_vector_factors(0, (np.round(S.u01(H * W * L, 5) * 3.0) / 3.0).reshape(-1, L))
    p = b.add_potts_pairwise(1, L, np.where(S.u01(len(a), 6) < 0.5, -0.5, 0.75))
    b.add_interleaved_messages(np.tile(np.array([0, 1], np.int32), len(a)), np.stack([u[a], u[bb]], 1).reshape(-1), np.repeat(p, 2))
    b.add_relations(np.stack([u[a], p], 1).reshape(-1), np.stack([p, u[bb]], 1).reshape(-1))
    m = b.finish()
    for mode in MODES:
        _check(eng, m, mode, 3)
    cls = eng.plan.schedule_classes(M.FORWARD, M.REPAM_ANISOTROPIC)
    assert "generic" not in cls and any(k.startswith("pairwise") for k in cls)


def test_updated_pairwise_factors_rectangular_tables(eng):
    m = _scheduled_grid(6, 7, 0, M.SCHED_FULL, seed=3, order="row_major", dims=[2, 3, 7, 12, 30])
    for mode in MODES:
        _check(eng, m, mode, 3)
    cls = eng.plan.schedule_classes(M.FORWARD, M.REPAM_ANISOTROPIC)
    assert sum(v for k, v in cls.items() if k.startswith("pairwise")) > 0


@pytest.mark.parametrize("pairwise,L", [("dense", 32), ("dense", 16), ("potts", 8), ("dense", 21)])
def test_bounds_stay_tracked_through_uniform_weight_passes(pairwise, L):
    """uniform / damped_uniform (the rounding iterations of MpRoundingSolver): every message is received and then sent by the same
    record, so a pairwise factor's last touch is a SEND — its bound after that send is omega * min(theta_snapshot) up to rounding
    (kernels.hip, dense_pk_body), and LP::LowerBound after such a pass is a sum over tracked values, not a scan of all tables"""
    m = S.grid_model(24, 26, L, pairwise=pairwise, order="colour_major", seed=L, compute_primal=True)
    o = Oracle(m)
    e = E.Engine(0)
    try:
        e.upload(m)
        for mode in (M.REPAM_DAMPED_UNIFORM, M.REPAM_UNIFORM, M.REPAM_ANISOTROPIC):
            o.set_reparametrization(mode); e.set_reparametrization(mode)
            e.lower_bound()                                  # everything evaluated once
            for n in (1, 2):
                o.ComputePass(n); e.compute_pass(n)
                lb = e.lower_bound()
                assert e.lower_bound_recomputed() == 0, (mode, n, e.lower_bound_recomputed())
                assert abs(lb - o.LowerBound()) <= 1e-9 * max(1.0, abs(o.LowerBound()))
                e.invalidate_lower_bounds()
                assert abs(e.lower_bound() - lb) <= 1e-12 * max(1.0, abs(lb)) and e.lower_bound_recomputed() == m.n_factors
        # a rounding pass under damped_uniform, then the bound: still nothing to recompute
        o.set_reparametrization(M.REPAM_DAMPED_UNIFORM); e.set_reparametrization(M.REPAM_DAMPED_UNIFORM)
        o.ComputePassAndPrimal(9); e.compute_pass_and_primal(9)
        lb = e.lower_bound()
        assert e.lower_bound_recomputed() == 0 and abs(lb - o.LowerBound()) <= 1e-9 * max(1.0, abs(o.LowerBound()))
        assert np.array_equal(e.download_duals(), o.duals())
    finally:
        e.close()


@pytest.mark.gpu
@pytest.mark.parametrize("order", ["colour_major", "index"])
def test_c4_is_the_same_problem_on_every_path(order):
    """bench.py --workload c4 --c4-order: the one-GPU run (GraphSweep with costs generated in HBM per part), the lock-step driver
    (LockstepGraph) and a plain engine on the host-built counter_graph_model in that variable order run the same sweep: duals of
    the unpartitioned model bit for bit, the oracle's bound"""
    import torch
    from lp_mp_amd import lockstep as LS, multi_gpu as MG
    from oracle.binding import Oracle
    n, m, L, passes = 3000, 12000, 16, 3
    torch.cuda.set_device(0)
    a = MG.GraphSweep(torch, None, n, m, L, M.REPAM_ANISOTROPIC, seed=1, order=order)
    b = LS.LockstepGraph(torch, None, n, m, L, M.REPAM_ANISOTROPIC, seed=1, order=order)
    try:
        assert (a.rank_of is None) == (order == "index") and (a.rank_of is None or np.array_equal(a.rank_of, b.rank_of))
        gm = S.counter_graph_model(n, m, L, 1, rank=a.rank_of)
        o = Oracle(gm); o.set_reparametrization(M.REPAM_ANISOTROPIC); o.ComputePass(passes)
        a.compute_pass(passes); b.compute_pass(passes)
        torch.cuda.synchronize()
        assert np.array_equal(a.dualt.cpu().numpy(), o.duals())          # one part: local order = global order
        assert np.array_equal(b.dualt.cpu().numpy(), o.duals())
        for lb in (a.lower_bound(), b.lower_bound()):
            assert abs(lb - o.LowerBound()) <= 1e-9 * abs(o.LowerBound())
        assert a.levels is None                                            # (one GPU: planned when asked for)
        a.query_info()
        assert a.levels == list(b.levels) and (a.levels[0] <= 12 if order == "colour_major" else a.levels[0] > 12)
        assert a.global_updates_per_pass == 4 * m
    finally:
        a.engine.close(); b.close()
