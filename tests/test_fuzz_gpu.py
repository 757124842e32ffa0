"""Randomised parity: random factor graphs mixing every device kind, every message schedule, random label counts,
random relations, every weight mode, both send rules, and random iterator-range passes — HIP engine vs oracle,
duals bit for bit."""
import numpy as np
import pytest

from lp_mp_amd import engine as E
from lp_mp_amd import model as M
from oracle.binding import Oracle

pytestmark = pytest.mark.gpu
MODES = (M.REPAM_ANISOTROPIC, M.REPAM_ANISOTROPIC2, M.REPAM_UNIFORM, M.REPAM_DAMPED_UNIFORM)
SCHEDS = (M.SCHED_LEFT, M.SCHED_RIGHT, M.SCHED_FULL, M.SCHED_ONLY_SEND, M.SCHED_NONE)


def random_model(rng):
    """unaries of two label counts, dense + Potts pairwise, vector-vector min-norm links, labeling-list factors"""
    La, Lb = int(rng.choice([2, 3, 4, 5, 8, 16])), int(rng.choice([3, 4, 7, 8]))
    s_up = [int(rng.choice(SCHEDS[:3])) for _ in range(4)]
    s_mn = int(rng.choice(SCHEDS))
    s_lab = int(rng.choice(SCHEDS[:4]))
    # factor types: 0 unary(La) 1 unary(Lb) 2 pairwise(La,La) dense/potts 3 pairwise(La,Lb) dense 4 edge-label 5 triplet
    mt = [M.MsgType(0, 2, s_up[0], 0, 1, M.M_UNARY_PAIRWISE, 0), M.MsgType(0, 2, s_up[1], 0, 1, M.M_UNARY_PAIRWISE, 1),
          M.MsgType(0, 3, s_up[2], 0, 1, M.M_UNARY_PAIRWISE, 0), M.MsgType(1, 3, s_up[3], 0, 1, M.M_UNARY_PAIRWISE, 1),
          M.MsgType(0, 0, s_mn, 0, 0, M.M_MINNORM, 0)] + \
         [M.MsgType(4, 5, s_lab, 0, 1, M.M_LABELING, k) for k in range(3)]
    b = M.ModelBuilder(6, mt)
    trip = [(0, 1, 1), (1, 0, 1), (1, 1, 0), (1, 1, 1)]
    for k in range(3):
        b.add_labeling_table([(1,)], trip, (k,))
    na, nb = int(rng.integers(4, 14)), int(rng.integers(2, 8))
    ua = b.add_vector_factors(0, rng.uniform(-1, 1, (na, La)))
    ub = b.add_vector_factors(1, rng.uniform(-1, 1, (nb, Lb)))
    rel = []
    for _ in range(int(rng.integers(3, 2 * na))):
        i, j = sorted(rng.choice(na, 2, replace=False))
        if rng.uniform() < 0.5:
            p = b.add_dense_pairwise(2, rng.uniform(-1, 1, (1, La, La)))[0]
        else:
            p = b.add_potts_pairwise(2, La, [rng.uniform(-1, 1)])[0]
        b.add_messages(0, ua[i], p); b.add_messages(1, ua[j], p)
        rel += [(ua[i], p), (p, ua[j])]
    for _ in range(int(rng.integers(1, na))):
        i, j = int(rng.integers(na)), int(rng.integers(nb))
        p = b.add_dense_pairwise(3, rng.uniform(-1, 1, (1, La, Lb)))[0]
        b.add_messages(2, ua[i], p); b.add_messages(3, ub[j], p)
        rel += [(ua[i], p), (p, ub[j])]
    for _ in range(int(rng.integers(0, na))):
        i, j = rng.choice(na, 2, replace=False)
        b.add_messages(4, ua[i], ua[j])
    ne = int(rng.integers(3, 9))
    e = b.add_vector_factors(4, rng.uniform(-1, 1, (ne, 1)), implicit_origin=bool(rng.integers(2)))
    for _ in range(int(rng.integers(1, 5))):
        t = b.add_vector_factors(5, rng.uniform(-0.5, 0.5, (1, 4)), implicit_origin=bool(rng.integers(2)))[0]
        for k, ei in enumerate(rng.choice(ne, 3, replace=False)):
            b.add_messages(5 + k, e[ei], t)
            rel.append((e[ei], t))
    if rng.uniform() < 0.8 and rel:                       # relations (a DAG by construction), sometimes none at all
        keep = rng.uniform(size=len(rel)) < rng.uniform(0.3, 1.0)
        r = np.array([x for x, k in zip(rel, keep) if k], np.int32).reshape(-1, 2)
        if r.shape[0]:
            b.add_relations(r[:, 0], r[:, 1])
    b.constant = float(rng.uniform(-1, 1))
    return b.finish()


def random_rows(rng, plan_like, oracle, m):
    """a random iterator-range pass: random factor subset in random order, random masks, random weights (row sums <= 1)"""
    off, ent = oracle.msg_lists()
    n = m.n_factors
    f = rng.permutation(n)[: int(rng.integers(1, n + 1))].astype(np.int32)
    caps = {M.SCHED_LEFT: (0, 1, 0, 1), M.SCHED_RIGHT: (1, 0, 1, 0), M.SCHED_FULL: (1, 1, 1, 1),
            M.SCHED_ONLY_SEND: (1, 1, 0, 0), M.SCHED_NONE: (0, 0, 0, 0)}     # to_left, to_right, from_left, from_right
    om, mk, oo, mo = [], [], [0], [0]
    for fi in f:
        ns = nr = 0
        for x in ent[off[fi]:off[fi + 1]]:
            msg, role = int(x) // 2, int(x) % 2
            c = caps[m.mtypes[m.m_type[msg]].schedule]
            ns += c[1] if role == 0 else c[0]
            nr += c[3] if role == 0 else c[2]
        w = rng.uniform(0, 1, ns) * (rng.uniform(size=ns) < 0.7)
        if w.sum() > 0:
            w = w / w.sum() * rng.uniform(0.2, 1.0)
        om += list(w); mk += list((rng.uniform(size=nr) < 0.6).astype(np.uint8))
        oo.append(len(om)); mo.append(len(mk))
    return f, np.array(oo, np.int64), np.array(om, np.float64), np.array(mo, np.int64), np.array(mk, np.uint8)


@pytest.mark.parametrize("seed", range(24))
def test_random_models_all_modes_and_custom_passes(seed):
    rng = np.random.default_rng(1000 + seed)
    m = random_model(rng)
    eng = E.Engine(0)
    try:
        for rtype in (0, 1):
            for mode in MODES:
                o = Oracle(m)
                o.set_reparametrization_type(rtype)
                o.set_reparametrization(mode)
                eng.upload(m)
                eng.set_reparametrization_type(rtype)
                eng.set_reparametrization(mode)
                assert abs(eng.lower_bound() - o.LowerBound()) <= 1e-9 * max(1.0, abs(o.LowerBound()))
                eng.compute_pass(2); o.ComputePass(2)
                eng.forward_pass(); o.ComputeForwardPass()
                assert np.array_equal(eng.download_duals(), o.duals()), (seed, rtype, mode)
                for _ in range(2):
                    rows = random_rows(rng, None, o, m)
                    eng.compute_pass_custom(*rows); o.compute_pass_custom(*rows)
                eng.backward_pass(); o.ComputeBackwardPass()
                eng.compute_pass(1); o.ComputePass(1)
                assert np.array_equal(eng.download_duals(), o.duals()), (seed, rtype, mode)
                lb, lbo = eng.lower_bound(), o.LowerBound()
                assert abs(lb - lbo) <= 1e-9 * max(1.0, abs(lbo))
                flb = eng.factor_lower_bounds()
                ref = np.array([o.factor_lower_bound(f) for f in range(m.n_factors)])
                assert np.max(np.abs(flb - ref)) <= 1e-12
    finally:
        eng.set_reparametrization_type(0)
        eng.close()


def random_mrf(rng, primal=False):
    """pure unary / pairwise MRF on a random graph with ONE label count of a fast kernel class: every unary runs the
    packed dense / Potts kernels, under odd orders (random subset of the relations), with occasional duplicate
    messages and a random mix of dense and Potts edges or all of one kind"""
    from lp_mp_amd import synthetic as S
    L = int(rng.choice([4, 8, 16, 32]))
    n = int(rng.integers(5, 40))
    kind = rng.choice(["dense", "potts", "mixed"])
    b = M.ModelBuilder(2, S.mrf_mtypes() if rng.uniform() < 0.7 else
                       [M.MsgType(0, 1, M.SCHED_LEFT, 0, 0, M.M_UNARY_PAIRWISE, 0), M.MsgType(0, 1, M.SCHED_LEFT, 0, 0, M.M_UNARY_PAIRWISE, 1)],
                       [1, 0] if primal else None)
    u = b.add_vector_factors(0, rng.uniform(0, 1, (n, L)))
    rel = []
    for _ in range(int(rng.integers(n, 3 * n))):
        i, j = sorted(rng.choice(n, 2, replace=False))
        dense = kind == "dense" or (kind == "mixed" and rng.uniform() < 0.5)
        p = b.add_dense_pairwise(1, rng.uniform(0, 1, (1, L, L)))[0] if dense else b.add_potts_pairwise(1, L, [rng.uniform(-0.5, 1)])[0]
        b.add_messages(0, u[i], p); b.add_messages(1, u[j], p)
        if rng.uniform() < 0.08:
            b.add_messages(0, u[i], p)                      # duplicate message into the same vector
        rel += [(u[i], p), (p, u[j])]
    keep = rng.uniform(size=len(rel)) < rng.choice([1.0, 0.9, 0.5, 0.0])
    r = np.array([x for x, k in zip(rel, keep) if k], np.int32).reshape(-1, 2)
    if r.shape[0]:
        b.add_relations(r[:, 0], r[:, 1])
    return b.finish()


@pytest.mark.parametrize("seed", range(30))
def test_random_mrfs_fast_kernels_multi_pass_calls_and_fused_custom_schedules(seed):
    rng = np.random.default_rng(5000 + seed)
    m = random_mrf(rng)
    eng = E.Engine(0)
    try:
        for rtype in (0, 1):
            for mode in MODES:
                o = Oracle(m)
                o.set_reparametrization_type(rtype); o.set_reparametrization(mode)
                eng.upload(m)
                eng.set_reparametrization_type(rtype); eng.set_reparametrization(mode)
                for n in (1, 3, 2):
                    eng.compute_pass(n); o.ComputePass(n)
                    assert np.array_equal(eng.download_duals(), o.duals()), (seed, rtype, mode, n)
                    assert abs(eng.lower_bound() - o.LowerBound()) <= 1e-9 * max(1.0, abs(o.LowerBound()))
                # a fused custom schedule over [forward rows, backward rows, random rows]
                rows = []
                for d in (M.FORWARD, M.BACKWARD):
                    oo, om = o.omega(d, mode); mo, mk = o.mask(d, mode)
                    rows.append((o.update_order(d), oo, om, mo, mk))
                rows.append(random_rows(rng, None, o, m))
                from lp_mp_amd.multi_gpu import _cat_rows
                cat = _cat_rows(*rows)
                sid = eng.schedule_create(*cat, fuse=True)
                for _ in range(2):
                    eng.schedule_run(sid); o.compute_pass_custom(*cat)
                assert np.array_equal(eng.download_duals(), o.duals()), (seed, rtype, mode, "fused custom")
                eng.schedule_destroy(sid)
                flb = eng.factor_lower_bounds()
                ref = np.array([o.factor_lower_bound(f) for f in range(m.n_factors)])
                assert np.max(np.abs(flb - ref)) <= 1e-12
    finally:
        eng.set_reparametrization_type(0)
        eng.close()


def random_mrf_any_labels(rng, primal=False):
    """unary / pairwise MRF where every variable has its own label count (1..130): rectangular dense tables between
    any two variables, Potts between variables of equal count -> the run-time-dims kernel classes of every padded
    width, next to exact classes and (above 32 labels, or for variables with both kinds of edges) the generic one"""
    from lp_mp_amd import synthetic as S
    n = int(rng.integers(6, 30))
    # (round 6: label counts above 32 of every LDS size class of the streaming kernel — 33 ... 64, 65 ... 128, 129 ... 192)
    pool = rng.choice([1, 2, 3, 4, 5, 6, 7, 8, 11, 16, 21, 27, 32, 40, 33, 48, 64, 65, 100, 130], size=int(rng.integers(1, 5)))
    kind = rng.choice(["dense", "potts", "mixed"])
    labels = rng.choice(pool if kind != "potts" else pool[:1], size=n)
    b = M.ModelBuilder(2, S.mrf_mtypes(), [1, 0] if primal else None)
    u = [b.add_vector_factors(0, rng.uniform(0, 1, (1, int(L))))[0] for L in labels]
    rel = []
    for _ in range(int(rng.integers(n, 3 * n))):
        i, j = sorted(rng.choice(n, 2, replace=False))
        Li, Lj = int(labels[i]), int(labels[j])
        potts = Li == Lj and (kind == "potts" or (kind == "mixed" and rng.uniform() < 0.5))
        p = b.add_potts_pairwise(1, Li, [rng.uniform(-0.5, 1)])[0] if potts else b.add_dense_pairwise(1, rng.uniform(0, 1, (1, Li, Lj)))[0]
        b.add_messages(0, u[i], p); b.add_messages(1, u[j], p)
        rel += [(u[i], p), (p, u[j])]
    keep = rng.uniform(size=len(rel)) < rng.choice([1.0, 0.9, 0.5])
    r = np.array([x for x, k in zip(rel, keep) if k], np.int32).reshape(-1, 2)
    if r.shape[0]:
        b.add_relations(r[:, 0], r[:, 1])
    return b.finish()


@pytest.mark.parametrize("seed", range(30))
def test_random_mrfs_any_label_count_runtime_dims_kernels(seed):
    rng = np.random.default_rng(9000 + seed)
    m = random_mrf_any_labels(rng)
    eng = E.Engine(0)
    try:
        for rtype in (0, 1):
            for mode in MODES:
                o = Oracle(m)
                o.set_reparametrization_type(rtype); o.set_reparametrization(mode)
                eng.upload(m)
                eng.set_reparametrization_type(rtype); eng.set_reparametrization(mode)
                for n in (1, 2):
                    eng.compute_pass(n); o.ComputePass(n)
                    assert np.array_equal(eng.download_duals(), o.duals()), (seed, rtype, mode, n)
                    assert abs(eng.lower_bound() - o.LowerBound()) <= 1e-9 * max(1.0, abs(o.LowerBound()))
                rows = random_rows(rng, None, o, m)
                eng.compute_pass_custom(*rows); o.compute_pass_custom(*rows)
                eng.forward_pass(); o.ComputeForwardPass()
                assert np.array_equal(eng.download_duals(), o.duals()), (seed, rtype, mode, "custom")
                flb = eng.factor_lower_bounds()
                ref = np.array([o.factor_lower_bound(f) for f in range(m.n_factors)])
                assert np.max(np.abs(flb - ref)) <= 1e-12
    finally:
        eng.set_reparametrization_type(0)
        eng.close()


@pytest.mark.parametrize("seed", range(24))
def test_random_mrfs_primal_rounding(seed):
    """...AndPrimal passes on random MRFs (duplicate messages, odd orders, every label count): labels, duals and
    the primal cost against the oracle, interleaved with plain passes and random iterator-range passes"""
    rng = np.random.default_rng(13000 + seed)
    m = random_mrf(rng, primal=True) if seed % 2 else random_mrf_any_labels(rng, primal=True)
    _primal_steps(m, rng, seed)


def _primal_steps(m, rng, seed):
    eng = E.Engine(0)
    try:
        for mode in MODES:
            o = Oracle(m); o.set_reparametrization(mode)
            eng.upload(m); eng.set_reparametrization(mode)
            it, last = 0, 0                                 # time stamps must not decrease (reference assert)
            for step in range(6):
                what = int(rng.integers(4))
                if what == 0 and last > 2 * it + 1:
                    what = 1
                if what == 0:
                    eng.forward_pass_and_primal(it); o.ComputeForwardPassAndPrimal(it); last = 2 * it + 1
                elif what == 1:
                    eng.backward_pass_and_primal(it); o.ComputeBackwardPassAndPrimal(it); last = 2 * it + 2
                elif what == 2:
                    if last > 2 * it + 1:
                        it += 1
                    eng.compute_pass_and_primal(it); o.ComputePassAndPrimal(it); last = 2 * it + 2
                else:
                    rows = random_rows(rng, None, o, m)
                    eng.compute_pass_custom(*rows); o.compute_pass_custom(*rows)
                    eng.compute_pass(1); o.ComputePass(1)
                it += int(rng.integers(2))                   # repeated time stamps keep the labels
                assert np.array_equal(eng.download_primal(), o.primal()), (seed, mode, step, what)
                assert np.array_equal(eng.download_duals(), o.duals()), (seed, mode, step, what)
                assert eng.check_primal_consistency() == o.CheckPrimalConsistency()
                c, co = eng.evaluate_primal(), o.EvaluatePrimal()
                assert (c == co) if np.isinf(co) else abs(c - co) <= 1e-9 * max(1.0, abs(co))
    finally:
        eng.close()


def random_mrf_rounding_pairwise(rng):
    """unary / pairwise MRFs whose PAIRWISE type rounds itself (and, half of the time, the unary type too) under `full`,
    `right` or `left` schedules: random graphs, one label count of a packed class or mixed small ones, dense / Potts,
    odd orders, occasional duplicate messages"""
    sched = int(rng.choice([M.SCHED_FULL, M.SCHED_RIGHT, M.SCHED_LEFT]))
    mt = [M.MsgType(0, 1, sched, 0, 1, M.M_UNARY_PAIRWISE, 0), M.MsgType(0, 1, sched, 0, 1, M.M_UNARY_PAIRWISE, 1)]
    b = M.ModelBuilder(2, mt, [int(rng.integers(2)), 1])
    n = int(rng.integers(3, 30))
    if rng.uniform() < 0.5:
        dims = [int(rng.choice([4, 8, 16, 32]))] * n
    else:
        dims = [int(x) for x in rng.integers(2, 12, n)]
    u = np.concatenate([b.add_vector_factors(0, rng.uniform(0, 1, (1, d))) for d in dims])
    rel = []
    for _ in range(int(rng.integers(max(1, n // 2), 3 * n))):
        i, j = sorted(int(x) for x in rng.choice(n, 2, replace=False))
        if dims[i] == dims[j] and rng.uniform() < 0.4:
            p = b.add_potts_pairwise(1, dims[i], [rng.uniform(-0.5, 1)])[0]
        else:
            p = b.add_dense_pairwise(1, rng.uniform(0, 1, (1, dims[i], dims[j])))[0]
        b.add_messages(0, u[i], p)
        if rng.uniform() < 0.9:
            b.add_messages(1, u[j], p)                      # else: a side without a unary
        if rng.uniform() < 0.05:
            b.add_messages(0, u[i], p)                      # duplicate message of the same unary
        rel += [(u[i], p), (p, u[j])]
    keep = rng.uniform(size=len(rel)) < rng.choice([1.0, 0.9, 0.5, 0.0])
    r = np.array([x for x, k in zip(rel, keep) if k], np.int32).reshape(-1, 2)
    if r.shape[0]:
        b.add_relations(r[:, 0], r[:, 1])
    return b.finish()


@pytest.mark.parametrize("seed", range(30))
def test_random_mrfs_pairwise_factors_round_themselves(seed):
    rng = np.random.default_rng(17000 + seed)
    _primal_steps(random_mrf_rounding_pairwise(rng), rng, seed)


def random_bipartite_mrf(rng):
    """2-colourable non-grid graphs in colour-major order (all of side A, then all of side B) with irregular degrees,
    mixed dense / Potts edges, occasional duplicate messages and a random subset of the relations: the shapes for which
    lpmp_compute_pass(n >= 2) may join consecutive passes at their seam (engine.cpp, plan_rotation)"""
    from lp_mp_amd import synthetic as S
    L = int(rng.choice([4, 8, 16, 32]))
    na, nb = int(rng.integers(3, 20)), int(rng.integers(3, 20))
    kind = rng.choice(["dense", "potts", "mixed"])
    b = M.ModelBuilder(2, S.mrf_mtypes() if rng.uniform() < 0.7 else
                       [M.MsgType(0, 1, M.SCHED_LEFT, 0, 0, M.M_UNARY_PAIRWISE, 0), M.MsgType(0, 1, M.SCHED_LEFT, 0, 0, M.M_UNARY_PAIRWISE, 1)])
    u = b.add_vector_factors(0, rng.uniform(0, 1, (na + nb, L)))
    rel = []
    seen = set()
    for _ in range(int(rng.integers(max(na, nb), 3 * (na + nb)))):
        i, j = int(rng.integers(na)), na + int(rng.integers(nb))
        if (i, j) in seen and rng.uniform() < 0.7:
            continue
        seen.add((i, j))
        dense = kind == "dense" or (kind == "mixed" and rng.uniform() < 0.5)
        p = b.add_dense_pairwise(1, rng.uniform(0, 1, (1, L, L)))[0] if dense else b.add_potts_pairwise(1, L, [rng.uniform(-0.5, 1)])[0]
        b.add_messages(0, u[i], p); b.add_messages(1, u[j], p)
        if rng.uniform() < 0.05:
            b.add_messages(1, u[j], p)                      # duplicate message into the same vector
        rel += [(u[i], p), (p, u[j])]
    keep = rng.uniform(size=len(rel)) < rng.choice([1.0, 1.0, 0.9, 0.5])
    r = np.array([x for x, k in zip(rel, keep) if k], np.int32).reshape(-1, 2)
    if r.shape[0]:
        b.add_relations(r[:, 0], r[:, 1])
    return b.finish()


@pytest.mark.parametrize("seed", range(40))
def test_random_bipartite_graphs_multi_pass_calls_equal_single_passes(seed):
    """compute_pass(n >= 2) — fused, and joined across passes where the op-by-op check allows it — against n
    sequential reference passes, on non-grid 2-colourable graphs"""
    rng = np.random.default_rng(17000 + seed)
    m = random_bipartite_mrf(rng)
    eng = E.Engine(0)
    try:
        for mode in MODES:
            o = Oracle(m); o.set_reparametrization(mode)
            eng.upload(m); eng.set_reparametrization(mode)
            for n in (2, 5, 1, 3):
                eng.compute_pass(n); o.ComputePass(n)
                assert np.array_equal(eng.download_duals(), o.duals()), (seed, mode, n, eng.plan.pass_rotates(mode))
            assert abs(eng.lower_bound() - o.LowerBound()) <= 1e-9 * max(1.0, abs(o.LowerBound()))
    finally:
        eng.close()


@pytest.mark.parametrize("seed", range(24))
def test_random_labeling_list_models_in_the_level_loop(seed, monkeypatch):
    """multicut-style models (edge variables + triplet / quadruple labeling-list factors) of random size, density and
    locality with the level loop forced on every deep-enough sweep (LPMP_CHAIN_MIN=2): the one-lane-per-op body
    (kernels.hip, label_ops_body) where a launch qualifies, the generic body between the same barriers where it does not
    (more than 8 receives or sends: uniform weights on dense instances), residual sends (always the generic body)"""
    from lp_mp_amd import synthetic as S
    monkeypatch.setenv("LPMP_CHAIN_MIN", "2")
    rng = np.random.default_rng(23000 + seed)
    if seed % 3 == 0:
        m = S.multicut_triangle_model(int(rng.integers(6, 40)), int(rng.integers(2, 60)), seed=seed)
    else:
        ne = int(rng.integers(30, 500))
        m = S.c5_model(int(rng.integers(4, 12)), int(rng.integers(4, 12)), 8, ne, int(rng.integers(ne // 4, 2 * ne)), int(rng.integers(0, ne)),
                       seed=seed, window=int(rng.choice([6, 16, 64, 100000])), colour_edge_vars=bool(rng.integers(2)))
    eng = E.Engine(0)
    try:
        for rtype in (0, 1):
            for mode in MODES:
                o = Oracle(m)
                o.set_reparametrization_type(rtype); o.set_reparametrization(mode)
                eng.upload(m)
                eng.set_reparametrization_type(rtype); eng.set_reparametrization(mode)
                for step in range(3):
                    what = int(rng.integers(3))
                    if what == 0:
                        eng.compute_pass(2); o.ComputePass(2)
                    elif what == 1:
                        eng.backward_pass(); o.ComputeBackwardPass()
                    else:
                        eng.forward_pass(); o.ComputeForwardPass()
                    assert np.array_equal(eng.download_duals(), o.duals()), (seed, rtype, mode, step, what)
                    assert abs(eng.lower_bound() - o.LowerBound()) <= 1e-9 * max(1.0, abs(o.LowerBound()))
        eng.set_reparametrization_type(0)
    finally:
        eng.close()
