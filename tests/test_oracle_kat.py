"""Pins the CPU oracle (oracle/lpmp_oracle.c) against

 (1) every known-answer value the reference's own tests hold for the sweep path, and
 (2) outputs of the reference itself recorded in SURVEY.md (the surveyor ran the reference's
     LP<FMC> in this container; the reference is not buildable without stand-in headers).

Reference test files are cited per test (paths relative to /root/reference).
"""
import numpy as np
import pytest

from lp_mp_amd import model as M
from lp_mp_amd import synthetic as S
from oracle.binding import Oracle, mt19937_u01, synth_u01


# ---- test/test_model.cpp:18-48 -------------------------------------------------------------
def _toy_model():
    b = M.ModelBuilder(1, [M.MsgType(0, 0, M.SCHED_LEFT, 0, 0, M.M_MINNORM, 0)])
    f = b.add_vector_factors(0, [[0, 1], [1, 0], [0, 0]])
    b.add_messages(0, f[0], f[1])
    b.add_messages(0, f[0], f[2])
    return b.finish()


def test_toy_model_counts_and_lower_bound():
    m = _toy_model()
    o = Oracle(m)
    off, ent = o.msg_lists()
    assert list(np.diff(off)) == [2, 1, 1]               # no_messages(): f1 2, f2 1, f3 1
    o.set_reparametrization(M.REPAM_ANISOTROPIC)
    assert list(o.update_order(M.FORWARD)) == [0]        # only f1 sends/receives under schedule left
    om_off, om = o.omega(M.FORWARD, M.REPAM_ANISOTROPIC)
    assert om.shape[0] == 2                              # f1->no_send_messages() == 2
    # no relations => reverse insertion order (topological_sort.hxx:100-144)
    assert list(o.order(M.FORWARD)) == [2, 1, 0]
    o.ComputePass(1000)                                  # default --maxIter 1000
    assert abs(o.LowerBound() - 1.0) <= 1e-8


# ---- test/vector.cpp:62-94: matrix::min1 / min2 ------------------------------------------
MAT56 = np.array([[-2.0, 0.0, 2.0, -0.5, 0.0, 0.5],
                  [-1.0, 0.0, 1.0, -0.5, 0.0, 0.5],
                  [-0.0, -4.0, 0.5, -0.5, 0.0, 0.5],
                  [1.0, 0.0, -1.0, -0.5, 0.0, 0.5],
                  [2.0, 0.0, -2.0, -0.5, 0.0, 0.5]])


def _single_pairwise(T, m1=None, m2=None, potts=None, L=None):
    b = M.ModelBuilder(2, S.mrf_mtypes())
    if potts is None:
        d0, d1 = T.shape
        u0 = b.add_vector_factors(0, np.zeros((1, d0)))
        u1 = b.add_vector_factors(0, np.zeros((1, d1)))
        p = b.add_dense_pairwise(1, T)
    else:
        d0 = d1 = L
        u0 = b.add_vector_factors(0, np.zeros((1, L)))
        u1 = b.add_vector_factors(0, np.zeros((1, L)))
        p = b.add_potts_pairwise(1, L, [potts])
    b.add_messages(0, u0, p)
    b.add_messages(1, u1, p)
    m = b.finish()
    if m1 is not None:
        off = m.dual_offsets()[p[0]]
        m.dual_data[off:off + d0] = m1
        m.dual_data[off + d0:off + d0 + d1] = m2
    return m, int(p[0])


def test_matrix_row_and_column_minima():
    m, p = _single_pairwise(MAT56)
    o = Oracle(m)
    assert list(o.message_value(0, True)) == [-2.0, -1.0, -4.0, -1.0, -2.0]          # min1
    assert list(o.message_value(1, True)) == [-2.0, -4.0, -2.0, -0.5, 0.0, 0.5]      # min2


# ---- test/simplex.cpp:8-12, :52-65 -----------------------------------------------------------
def test_simplex_lower_bounds():
    b = M.ModelBuilder(1, [])
    b.add_vector_factors(0, [[0.1, 0.2, 0.05, 1]])
    o = Oracle(b.finish())
    assert o.factor_lower_bound(0) == 0.05
    T = np.zeros((3, 3))
    for x in range(3):
        T[x, x] = -float(x) - 1.0
    m, p = _single_pairwise(T)
    assert Oracle(m).factor_lower_bound(p) == -3.0


# ---- test/simplex_marginalization.cpp:9-41 -----------------------------------------------------
def test_unary_pairwise_marginalization():
    T = np.array([[0.1, 0.2, 0.05], [0.3, 0.001, 0.2], [-0.3, -0.001, -0.2], [0.3, 0.001, 0.2]])
    m, p = _single_pairwise(T)
    o = Oracle(m)
    # marg starts at 0 and the op does marg -= 1.0 * min_marginal
    assert list(0.0 - o.message_value(0, True)) == [-0.05, -0.001, 0.3, -0.001]
    assert list(0.0 - o.message_value(1, True)) == [0.3, 0.001, 0.2]


# ---- test/potts_factor.cpp:8-72 ----------------------------------------------------------------
@pytest.mark.parametrize("diff", [1.0, -1.0])
@pytest.mark.parametrize("with_msgs", [False, True])
def test_potts_equals_dense(diff, with_msgs):
    L = 3
    T = np.where(np.eye(L) > 0, 0.0, diff)
    m1 = np.array([-0.1, 0.5, 0.8]) if with_msgs else np.zeros(3)
    m2 = np.array([1.5, 1.0, 0.0]) if with_msgs else np.zeros(3)
    md, pd = _single_pairwise(T, m1, m2)
    mp, pp = _single_pairwise(None, m1, m2, potts=diff, L=L)
    od, op = Oracle(md), Oracle(mp)
    assert od.factor_lower_bound(pd) == op.factor_lower_bound(pp)
    assert list(od.message_value(0, True)) == list(op.message_value(0, True))    # min_marginal_1
    assert list(od.message_value(1, True)) == list(op.message_value(1, True))    # min_marginal_2


# ---- test/graphical_model.cpp:90-137 -----------------------------------------------------------
NEG = np.array([[1.0, 0.0], [0.0, 1.0]])
POS = np.array([[0.0, 1.0], [1.0, 0.0]])


def _binary_mrf(n, edges):
    ei = np.array([e[0] for e in edges])
    ej = np.array([e[1] for e in edges])
    tabs = np.stack([e[2] for e in edges])
    return S.mrf_model(n, 2, ei, ej, np.zeros(n * 2), tables=tabs)


@pytest.mark.parametrize("mode", [M.REPAM_ANISOTROPIC, M.REPAM_DAMPED_UNIFORM, M.REPAM_UNIFORM])
def test_chain_and_frustrated_cycle_lower_bound_zero(mode):
    chain = _binary_mrf(5, [(0, 1, NEG), (1, 2, POS), (2, 3, POS), (3, 4, POS)])
    cyc = _binary_mrf(4, [(0, 1, NEG), (1, 2, POS), (2, 3, POS), (0, 3, POS)])
    for m in (chain, cyc):
        o = Oracle(m)
        o.set_reparametrization(mode)
        o.ComputePass(100)                         # --maxIter 100
        assert abs(o.LowerBound() - 0.0) <= 1e-8


# ---- test/multicut.cpp:8-32 and the labeling_message values recorded in SURVEY.md 4 ---------------
def test_labeling_factor_lower_bounds():
    b = S.multicut_builder()
    b.add_vector_factors(0, [[1.0]], implicit_origin=True)
    b.add_vector_factors(0, [[-1.0]], implicit_origin=True)
    b.add_vector_factors(1, [[1.0, 2.0, 3.3, 1.5]], implicit_origin=True)
    b.add_vector_factors(1, [[1.0, -0.5, -0.3, 1.5]], implicit_origin=True)
    o = Oracle(b.finish())
    assert o.factor_lower_bound(0) == 0.0
    assert o.factor_lower_bound(1) == -1.0
    assert o.factor_lower_bound(2) == 0.0
    assert o.factor_lower_bound(3) == -0.5


def test_labeling_message_values():
    b = S.multicut_builder()
    e = b.add_vector_factors(0, np.zeros((3, 1)), implicit_origin=True)
    t = b.add_vector_factors(1, [[1.0, -0.5, -0.3, 1.5]], implicit_origin=True)
    for k in range(3):
        b.add_messages(k, e[k], t[0])
    m = b.finish()
    o = Oracle(m)
    # send_message_to_left on a zero message: msg -= omega * msg_val
    assert 0.0 - o.message_value(0, True, 1.0)[0] == pytest.approx(0.5, abs=1e-15)
    assert 0.0 - o.message_value(1, True, 1.0)[0] == pytest.approx(-0.2, abs=1e-15)
    assert 0.0 - o.message_value(2, True, 0.5)[0] == pytest.approx(0.1, abs=1e-15)
    # RepamRight(index 0, +0.25): a weight-1 send of an edge cost 0.25 through message 0
    m.dual_data[0] = 0.25
    o2 = Oracle(m)
    o2.compute_pass_custom([int(e[0])], [0, 1], [1.0], [0, 1], [0])
    tri = o2.duals()[m.dual_offsets()[t[0]]:][:4]
    assert np.allclose(tri, [1.0, -0.25, -0.05, 1.75], atol=1e-15)
    assert o2.duals()[0] == 0.0


# ---- reference runs recorded in SURVEY.md 8(c), 8(a5) --------------------------------------------
def _survey_grid(H, W, L, costs):
    n = H * W
    return S.grid_model(H, W, L, unaries=costs[:n * L], tables=costs[n * L:])


def test_survey_reference_lower_bounds_small(golden_dir):
    g = np.load(golden_dir + "/survey_grids.npz")
    o = Oracle(_survey_grid(8, 8, 4, g["costs_8x8_L4"]))
    o.set_reparametrization(M.REPAM_ANISOTROPIC)
    exp = g["lb_8x8_L4_pass0_1_4"]
    assert o.LowerBound() == pytest.approx(exp[0], abs=5e-11)
    o.ComputePass(1)
    assert o.LowerBound() == pytest.approx(exp[1], abs=5e-11)
    o.ComputePass(3)
    assert o.LowerBound() == pytest.approx(exp[2], abs=5e-11)
    r, s = o.counters()
    n_msgs = 2 * len(S.grid_edges(8, 8)[0])
    assert r == 4 * n_msgs and s == 4 * n_msgs      # one receive + one send per message per pass


def test_survey_reference_weight_rows_and_modes(golden_dir):
    g = np.load(golden_dir + "/survey_grids.npz")
    m = _survey_grid(16, 16, 4, g["costs_16x16_L4"])
    exp_lb = g["lb_16x16_L4_pass1_aniso_uniform_damped"]
    rows = {
        M.REPAM_ANISOTROPIC: ([.5, .5], [0, 0], [.5, .5, 0, 0], [0, 0, 1, 1], [0, 0], [1, 1]),
        M.REPAM_UNIFORM: ([.5, .5], [1, 1], [.25] * 4, [1] * 4, [.5, .5], [1, 1]),
        M.REPAM_DAMPED_UNIFORM: ([1 / 3] * 2, [1, 1], [.2] * 4, [1] * 4, [1 / 3] * 2, [1, 1]),
    }
    for k, mode in enumerate((M.REPAM_ANISOTROPIC, M.REPAM_UNIFORM, M.REPAM_DAMPED_UNIFORM)):
        o = Oracle(m)
        assert o.LowerBound() == pytest.approx(g["lb_16x16_L4_start"][0], abs=5e-11)
        o.set_reparametrization(mode)
        upd = o.update_order(M.FORWARD)
        assert upd.shape[0] == 256                       # exactly H*W rows
        om_off, om = o.omega(M.FORWARD, mode)
        mk_off, mk = o.mask(M.FORWARD, mode)
        pos = {int(f): i for i, f in enumerate(upd)}
        first, interior, last = pos[0], pos[5 * 16 + 7], pos[255]
        e = rows[mode]
        for row, (eo, em) in ((first, e[0:2]), (interior, e[2:4]), (last, e[4:6])):
            assert np.allclose(om[om_off[row]:om_off[row + 1]], eo, atol=1e-15)
            assert list(mk[mk_off[row]:mk_off[row + 1]]) == list(em)
        o.ComputePass(1)
        assert o.LowerBound() == pytest.approx(exp_lb[k], abs=5e-10)


def test_survey_reference_lower_bounds_large():
    """256x256 L=8 and 128x128 L=32, 11 passes; inputs re-created with
    libstdc++'s mt19937_64 through oracle/gen_mt19937."""
    for (H, L, exp) in ((256, 8, (9279.9435163963, 39300.4786277831, 41683.8104192796)),
                        (128, 32, (526.3889804346, 5013.1486992415, 5396.2923808675))):
        n = H * H
        E = len(S.grid_edges(H, H)[0])
        o = Oracle(_survey_grid(H, H, L, mt19937_u01(12345, n * L + E * L * L)))
        o.set_reparametrization(M.REPAM_ANISOTROPIC)
        assert o.LowerBound() == pytest.approx(exp[0], abs=5e-10)
        o.ComputePass(1)
        assert o.LowerBound() == pytest.approx(exp[1], abs=5e-9)
        o.ComputePass(10)
        assert o.LowerBound() == pytest.approx(exp[2], abs=5e-9)


# ---- structural properties of the restated sweep ---------------------------------------------------
def test_lower_bound_monotone_and_order_is_topological():
    m = S.grid_model(12, 9, 5, seed=3)
    o = Oracle(m)
    for d, rel in ((M.FORWARD, m.rel_fwd), (M.BACKWARD, m.rel_bwd)):
        order = o.order(d)
        pos = np.empty(m.n_factors, np.int64)
        pos[order] = np.arange(m.n_factors)
        assert np.all(pos[rel[:, 0]] < pos[rel[:, 1]])
    for mode in (M.REPAM_ANISOTROPIC, M.REPAM_ANISOTROPIC2, M.REPAM_UNIFORM, M.REPAM_DAMPED_UNIFORM):
        o = Oracle(m)
        o.set_reparametrization(mode)
        for d in (M.FORWARD, M.BACKWARD):
            off, om = o.omega(d, mode)
            sums = np.add.reduceat(om, off[:-1])
            assert om.min() >= 0 and sums.max() <= 1 + 1e-8       # omega_valid, LP_MP.h:1008-1014
        lb = o.LowerBound()
        for _ in range(5):
            o.ComputePass(1)
            nlb = o.LowerBound()
            assert nlb >= lb - 1e-8
            lb = nlb


def test_custom_pass_equals_builtin():
    m = S.grid_model(7, 6, 4, pairwise="potts", seed=5)
    a, b = Oracle(m), Oracle(m)
    a.set_reparametrization(M.REPAM_ANISOTROPIC)
    a.ComputePass(2)
    for _ in range(2):
        for d in (M.FORWARD, M.BACKWARD):
            om_off, om = b.omega(d, M.REPAM_ANISOTROPIC)
            mk_off, mk = b.mask(d, M.REPAM_ANISOTROPIC)
            b.compute_pass_custom(b.update_order(d), om_off, om, mk_off, mk)
    assert np.array_equal(a.duals(), b.duals())


def test_sublist_weights_full_list_equal_builtin():
    m = S.grid_model(6, 6, 3, seed=2)
    o = Oracle(m)
    om_off, om, mk_off, mk = o.anisotropic_weights_sublist(o.order(M.FORWARD))
    r_off, r_om = o.omega(M.FORWARD, M.REPAM_ANISOTROPIC)
    q_off, q_mk = o.mask(M.FORWARD, M.REPAM_ANISOTROPIC)
    assert np.array_equal(om_off, r_off) and np.array_equal(om, r_om)
    assert np.array_equal(mk_off, q_off) and np.array_equal(mk, q_mk)


def test_counter_rng_matches_numpy_restatement():
    assert np.array_equal(synth_u01(1000, 42, 7), S.u01(1000, 42, 7))
    x = S.u01(100000, 1)
    assert 0.0 <= x.min() and x.max() < 1.0 and abs(x.mean() - 0.5) < 0.01


def test_oracle_behaviour_is_frozen(golden_dir):
    """regression vectors of the oracle itself (tests/golden/oracle_regression.npz, produced by
    make_oracle_regression.py — NOT reference outputs): bounds, dual checksums and rounded labels must not move when
    the oracle is edited"""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("mk", os.path.join(golden_dir, "make_oracle_regression.py"))
    mk = importlib.util.module_from_spec(spec); spec.loader.exec_module(mk)
    want = np.load(os.path.join(golden_dir, "oracle_regression.npz"))
    n = 0
    for name, m in mk.cases():
        for k, v in mk.run(m, primal=m.ftype_computes_primal.any()).items():
            ref = want[f"{name}/{k}"]
            if v.dtype.kind == "f":
                assert np.allclose(v, ref, rtol=1e-13, atol=1e-13), (name, k)
            else:
                assert np.array_equal(v, ref), (name, k)
            n += 1
    assert n == len(want.files)
