"""Host logic of the product (lp_mp_amd/csrc/plan.cpp, reached through the C ABI) against the oracle:
orderings, per-factor message lists, weights and receive masks for every mode.  No GPU needed."""
import ctypes as C

import numpy as np
import pytest

from lp_mp_amd import model as M
from lp_mp_amd import synthetic as S
from lp_mp_amd import engine as E
from oracle.binding import Oracle

MODES = (M.REPAM_ANISOTROPIC, M.REPAM_ANISOTROPIC2, M.REPAM_UNIFORM, M.REPAM_DAMPED_UNIFORM)


def _toy():
    b = M.ModelBuilder(1, [M.MsgType(0, 0, M.SCHED_LEFT, 0, 0, M.M_MINNORM, 0)])
    f = b.add_vector_factors(0, [[0, 1], [1, 0], [0, 0]])
    b.add_messages(0, f[0], f[1])
    b.add_messages(0, f[0], f[2])
    return b.finish()


def _full_schedule_model():
    """vector-vector messages in every schedule, plus unary/pairwise with schedule full and right."""
    mt = [M.MsgType(0, 0, M.SCHED_FULL, 0, 0, M.M_MINNORM, 0),
          M.MsgType(0, 0, M.SCHED_RIGHT, 0, 0, M.M_MINNORM, 0),
          M.MsgType(0, 0, M.SCHED_ONLY_SEND, 0, 0, M.M_MINNORM, 0),
          M.MsgType(0, 1, M.SCHED_FULL, 0, 1, M.M_UNARY_PAIRWISE, 0),
          M.MsgType(0, 1, M.SCHED_RIGHT, 0, 1, M.M_UNARY_PAIRWISE, 1)]
    b = M.ModelBuilder(2, mt)
    rng = np.random.default_rng(0)
    v = b.add_vector_factors(0, rng.uniform(-1, 1, (12, 3)))
    p = b.add_dense_pairwise(1, rng.uniform(-1, 1, (5, 3, 3)))
    for k in range(11):
        b.add_messages(k % 3, v[k], v[k + 1])
    for k in range(5):
        b.add_messages(3, v[2 * k], p[k])
        b.add_messages(4, v[2 * k + 1], p[k])
        b.add_relations(v[2 * k], p[k])
        b.add_relations(p[k], v[2 * k + 1])
    return b.finish()


MODELS = {
    "toy": _toy,
    "grid_row": lambda: S.grid_model(9, 7, 4, seed=1),
    "grid_colour": lambda: S.grid_model(8, 8, 3, order="colour_major", seed=2),
    "potts_grid": lambda: S.grid_model(6, 11, 5, pairwise="potts", seed=3),
    "chain": lambda: S.chain_model(100, 4),
    "random_graph": lambda: S.random_graph_model(200, 700, 16, seed=4, pairwise="potts"),
    "multicut": lambda: S.multicut_triangle_model(30, 40, seed=5),
    "all_schedules": _full_schedule_model,
}


@pytest.mark.parametrize("name", sorted(MODELS))
def test_plan_matches_oracle(name):
    m = MODELS[name]()
    o, p = Oracle(m), E.Plan(m)
    o_off, o_ent = o.msg_lists()
    p_off, p_ent = p.msg_lists(m.n_messages)
    assert np.array_equal(o_off, p_off) and np.array_equal(o_ent, p_ent)
    for d in (M.FORWARD, M.BACKWARD):
        assert np.array_equal(o.order(d), p.order(d))
        assert np.array_equal(o.update_order(d), p.update_order(d))
        for mode in MODES:
            a, b = o.omega(d, mode), p.omega(d, mode)
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
            a, b = o.mask(d, mode), p.mask(d, mode)
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_sublist_weights_match_oracle():
    """strict-subset rules of ComputeAnisotropicWeights (reference LP_MP.h:1263-1346)"""
    m = S.grid_model(8, 8, 3, seed=6)
    o, p = Oracle(m), E.Plan(m)
    order = o.order(M.FORWARD)
    rng = np.random.default_rng(1)
    for trial in range(6):
        keep = rng.uniform(size=order.shape[0]) < (0.3 + 0.1 * trial)
        sub = order[keep]
        a = o.anisotropic_weights_sublist(sub)
        b = p.anisotropic_weights(sub)
        for x, y in zip(a, b):
            assert np.array_equal(x, y)


def test_level_schedule_shapes():
    # row-major grid: one level per anti-diagonal; colour-major: two levels
    H, W = 10, 14
    p = E.Plan(S.grid_model(H, W, 4, order="row_major"))
    info = p.schedule_info(M.FORWARD, M.REPAM_ANISOTROPIC)
    n_msgs = 2 * (H * (W - 1) + W * (H - 1))
    assert info["n_levels"] == H + W - 1
    assert info["n_receives"] + info["n_sends"] == n_msgs      # each message once per direction
    p = E.Plan(S.grid_model(H, W, 4, order="colour_major"))
    for d in (M.FORWARD, M.BACKWARD):
        info = p.schedule_info(d, M.REPAM_ANISOTROPIC)
        assert info["n_levels"] == 2 and info["n_launches"] == 2
        assert info["n_receives"] + info["n_sends"] == n_msgs
    # algorithmic bytes of a pass follow SURVEY 8(d): 8L^2 + 40L per message + 32L per unary
    L = 4
    tot = sum(p.schedule_info(d, M.REPAM_ANISOTROPIC)["algorithmic_bytes"] for d in (0, 1))
    assert tot == n_msgs * (8 * L * L + 40 * L) + 32 * L * H * W


def test_invalid_models_are_rejected():
    m = S.grid_model(3, 3, 2)
    bad = S.grid_model(3, 3, 2)
    bad.m_left[0] = 10 ** 6
    with pytest.raises(E.EngineError):
        E.Plan(bad)
    bad = S.grid_model(3, 3, 2)
    bad.f_kind[0] = 7
    with pytest.raises(E.EngineError):
        E.Plan(bad)
    p = E.Plan(m)
    with pytest.raises(E.EngineError):
        p.omega(0, M.REPAM_MIXED)
    # labeling match tables index the left factor's labelings: an entry outside [0, n_left] would be an out-of-bounds
    # access in the labeling kernels
    for v in (-1, 2):
        bad = S.multicut_triangle_model(8, 4, seed=1)
        bad.tab_data[1] = v
        with pytest.raises(E.EngineError) as ei:
            E.Plan(bad)
        assert "out of range" in str(ei.value)
    # iterator-range passes: rows with entries need their arrays, offsets start at 0 and do not decrease
    upd = p.update_order(0)
    om_off, om = p.omega(0, M.REPAM_UNIFORM)
    mk_off, mk = p.mask(0, M.REPAM_UNIFORM)
    assert p.custom_schedule_info(upd, om_off, om, mk_off, mk)["n_sends"] == om.shape[0]
    L = E.lib()
    five = [C.c_int64() for _ in range(5)]

    def info(om_off_, om_ptr, mk_off_, mk_ptr):
        return L.lpmp_plan_custom_schedule_info(p.h, upd.shape[0], upd.ctypes.data, om_off_.ctypes.data, om_ptr, mk_off_.ctypes.data,
                                                mk_ptr, 0, *[C.addressof(x) for x in five])
    assert info(om_off, om.ctypes.data, mk_off, mk.ctypes.data) == 0
    assert info(om_off, None, mk_off, mk.ctypes.data) == -1 and b"omega array missing" in L.lpmp_last_error()
    assert info(om_off, om.ctypes.data, mk_off, None) == -1 and b"receive-mask array missing" in L.lpmp_last_error()
    assert info(om_off + 1, om.ctypes.data, mk_off, mk.ctypes.data) == -1
    dec = om_off.copy(); dec[1], dec[2] = dec[2], dec[1] - 1
    assert info(dec, om.ctypes.data, mk_off, mk.ctypes.data) == -1


def test_c_abi_exports_every_declared_symbol():
    import re, os
    hdr = open(os.path.join(os.path.dirname(__file__), "..", "include", "lpmp_engine.h")).read()
    declared = set(re.findall(r"\b(lpmp_[a-z_0-9]+)\s*\(", hdr))
    L = E.lib()
    for name in declared:
        assert hasattr(L, name), name
    assert declared == set(E.EXPORTS)


def test_engine_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(E.EngineError) as ei:
        E.Engine(0)
    assert "no CPU path" in str(ei.value) or ei.value.code == -3


def test_cyclic_relations_and_isolated_factors_match_oracle():
    """The reference's topological sort never detects a cycle (tempMarked is never set,
    topological_sort.hxx:100-144): a cyclic relation set just yields the DFS order.  Isolated factors (no
    messages) are not updated and keep their place in the ordering."""
    m = S.grid_model(4, 4, 3, seed=9)
    b_rel = np.concatenate([m.rel_fwd, np.array([[5, 0], [0, 5], [30, 2]], np.int32)])
    m.rel_fwd = np.ascontiguousarray(b_rel)
    o, p = Oracle(m), E.Plan(m)
    for d in (0, 1):
        assert np.array_equal(o.order(d), p.order(d))
        assert np.array_equal(o.update_order(d), p.update_order(d))
    b = M.ModelBuilder(2, S.mrf_mtypes())
    u = b.add_vector_factors(0, np.arange(12.0).reshape(4, 3))
    pw = b.add_dense_pairwise(1, np.ones((1, 3, 3)))
    b.add_messages(0, u[0], pw[0]); b.add_messages(1, u[2], pw[0])
    b.add_relations(u[0], pw[0]); b.add_relations(pw[0], u[2])
    m2 = b.finish()                                  # u[1], u[3] are isolated
    o, p = Oracle(m2), E.Plan(m2)
    for d in (0, 1):
        assert np.array_equal(o.order(d), p.order(d))
        assert sorted(p.update_order(d)) == [0, 2]
        for mode in MODES:
            assert np.array_equal(o.omega(d, mode)[1], p.omega(d, mode)[1])


def test_kernel_class_of_every_model_shape():
    """which device kernel a sweep's updated factors run on (lpmp_plan_schedule_classes): exact power-of-two classes,
    run-time-dims classes up to 32 labels, the streaming class up to 512 labels and for mixed dense / Potts
    neighbourhoods, the generic kernel for everything else"""
    from lp_mp_amd import engine as E
    def cls(m, d=M.FORWARD, mode=M.REPAM_ANISOTROPIC):
        return E.Plan(m).schedule_classes(d, mode)
    for L, dense, potts in ((4, "dense4", "potts4"), (32, "dense32", "potts32"), (3, "dense_v4", "potts_v4"),
                            (21, "dense_v32", "potts_v32"), (9, "dense_v16", "potts_v16"), (33, "dense_big", "dense_big"),
                            (512, "dense_big", "dense_big"), (513, "generic", "generic")):
        assert cls(S.grid_model(3, 4, L, seed=1)) == {dense: 12}, L
        assert cls(S.grid_model(3, 4, L, pairwise="potts", seed=1)) == {potts: 12}, L
    # updated pairwise factors (`full` schedule): tiny ones one lane each, up to 32 labels the packed pairwise classes
    for L, want in ((2, "small"), (4, "small"), (5, "pairwise8"), (16, "pairwise16"), (21, "pairwise32"), (33, "generic")):
        mt = [M.MsgType(0, 1, M.SCHED_FULL, 0, 1, M.M_UNARY_PAIRWISE, 0), M.MsgType(0, 1, M.SCHED_FULL, 0, 1, M.M_UNARY_PAIRWISE, 1)]
        b = M.ModelBuilder(2, mt)
        u = b.add_vector_factors(0, np.zeros((2, L)))
        p = b.add_dense_pairwise(1, np.zeros((1, L, L)))[0]
        b.add_messages(0, u[0], p); b.add_messages(1, u[1], p)
        b.add_relations(u[0], p); b.add_relations(p, u[1])
        c = cls(b.finish(), mode=M.REPAM_UNIFORM)
        assert c.get(want, 0) >= 1 and ("small" in c) == (want == "small"), (L, c)
    # rectangular tables: the padded width covers the largest dim of the unary's tables
    b = M.ModelBuilder(2, S.mrf_mtypes())
    u = [b.add_vector_factors(0, np.zeros((1, d)))[0] for d in (3, 20, 40)]
    for i, j, d in ((0, 1, (3, 20)), (1, 2, (20, 40))):
        p = b.add_dense_pairwise(1, np.zeros((1,) + d))[0]
        b.add_messages(0, u[i], p); b.add_messages(1, u[j], p)
        b.add_relations(u[i], p); b.add_relations(p, u[j])
    assert cls(b.finish(), mode=M.REPAM_UNIFORM) == {"dense_v32": 1, "dense_big": 2}
    # labeling lists (multicut): tiny factors, one lane each; with more than 8 labelings the wave-per-factor kernel
    assert set(cls(S.multicut_triangle_model(6, 4, seed=1), d=M.BACKWARD)) == {"small"}
    mt = [M.MsgType(0, 1, M.SCHED_LEFT, 0, 1, M.M_LABELING, 0)]
    b = M.ModelBuilder(2, mt)
    labs = [tuple(int(c) for c in np.binary_repr(k, 4)) for k in range(1, 10)]      # 9 labelings of 4 edges
    b.add_labeling_table([(1,)], labs, (0,))
    e = b.add_vector_factors(0, np.zeros((1, 1)), implicit_origin=True)[0]
    q = b.add_vector_factors(1, np.zeros((1, 9)), implicit_origin=True)[0]
    b.add_messages(0, e, q); b.add_relations(e, q)
    assert set(cls(b.finish(), d=M.BACKWARD)) == {"generic"}
    # COMPUTE_PRIMAL factor types keep the record of an update without active messages (isolated unary)
    b = M.ModelBuilder(2, S.mrf_mtypes(), [1, 0])
    b.add_vector_factors(0, np.zeros((2, 5)))
    assert cls(b.finish()) == {"dense_v8": 2}
    b = M.ModelBuilder(2, S.mrf_mtypes())
    b.add_vector_factors(0, np.zeros((2, 5)))
    assert cls(b.finish()) == {}


@pytest.mark.parametrize("block", range(4))
def test_plan_matches_oracle_on_random_models(block):
    """the randomised models of the GPU parity tests (mixed kinds, every schedule, random relations, duplicates): message
    lists, orders, weights and masks of the host analysis against the oracle's independent restatement — runs without
    a GPU, 25 models per block and family"""
    import importlib.util, os
    spec = importlib.util.spec_from_file_location("fuzz_models", os.path.join(os.path.dirname(__file__), "test_fuzz_gpu.py"))
    F = importlib.util.module_from_spec(spec); spec.loader.exec_module(F)
    for seed in range(25 * block, 25 * block + 25):
        for build in (F.random_model, F.random_mrf, F.random_mrf_any_labels):
            m = build(np.random.default_rng(77000 + seed))
            o, p = Oracle(m), E.Plan(m)
            o_off, o_ent = o.msg_lists()
            p_off, p_ent = p.msg_lists(m.n_messages)
            assert np.array_equal(o_off, p_off) and np.array_equal(o_ent, p_ent), (seed, build.__name__)
            for d in (M.FORWARD, M.BACKWARD):
                assert np.array_equal(o.order(d), p.order(d)), (seed, build.__name__)
                assert np.array_equal(o.update_order(d), p.update_order(d)), (seed, build.__name__)
                for mode in MODES:
                    a, b = o.omega(d, mode), p.omega(d, mode)
                    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), (seed, build.__name__, d, mode)
                    om = b[1]
                    a, b = o.mask(d, mode), p.mask(d, mode)
                    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), (seed, build.__name__, d, mode)
                    # and the schedule is well formed: every active receive / send is scheduled exactly once
                    info = p.schedule_info(d, mode)
                    assert info["n_receives"] == int(b[1].sum()) and info["n_sends"] == int((om != 0).sum())


@pytest.mark.parametrize("block", range(3))
def test_custom_pass_schedules_on_random_rows(block):
    """iterator-range passes with random factor subsets in random order, random masks and weights, fused and not: the
    compiled schedule holds exactly the active receives / sends of the rows (host only; also the sanitizer target)"""
    import importlib.util, os
    spec = importlib.util.spec_from_file_location("fuzz_models", os.path.join(os.path.dirname(__file__), "test_fuzz_gpu.py"))
    F = importlib.util.module_from_spec(spec); spec.loader.exec_module(F)
    for seed in range(40 * block, 40 * block + 40):
        rng = np.random.default_rng(88000 + seed)
        m = (F.random_model, F.random_mrf, F.random_mrf_any_labels)[seed % 3](rng)
        o, p = Oracle(m), E.Plan(m)
        for _ in range(3):
            rows = F.random_rows(rng, None, o, m)
            for fuse in (False, True):
                info = p.custom_schedule_info(*rows, fuse=fuse)
                assert info["n_receives"] == int(rows[4].sum()) and info["n_sends"] == int((rows[2] != 0).sum()), (seed, fuse)
        from lp_mp_amd.multi_gpu import _cat_rows
        seq = []
        for d in (M.FORWARD, M.BACKWARD):
            oo, om = o.omega(d, M.REPAM_ANISOTROPIC); mo, mk = o.mask(d, M.REPAM_ANISOTROPIC)
            seq.append((o.update_order(d), oo, om, mo, mk))
        seq.append(F.random_rows(rng, None, o, m))
        cat = _cat_rows(*seq)
        info = p.custom_schedule_info(*cat, fuse=True)
        assert info["n_receives"] == int(cat[4].sum()) and info["n_sends"] == int((cat[2] != 0).sum())


def test_baseline_configs_run_on_the_fast_kernel_classes():
    """BASELINE.json configs[0..4] (at reduced size: the class depends on the shape, not the size): none of them needs
    the generic wave-per-factor kernel"""
    from lp_mp_amd import engine as E
    shapes = {
        "C1 chain, 4-label Potts": (S.chain_model(100, 4), {"potts4"}),
        "C2 grid, 8-label Potts": (S.grid_model(16, 16, 8, pairwise="potts", order="colour_major", seed=2), {"potts8"}),
        "C3 grid, 32-label dense": (S.grid_model(12, 12, 32, order="colour_major", seed=3), {"dense32"}),
        "C4 random sparse graph, 16-label dense": (S.random_graph_model(300, 1500, 16, seed=4), {"dense16"}),
        "C5 grid + labeling-list factors": (S.c5_model(8, 8, 8, 60, 30, 12, seed=5), {"potts8", "small"}),
    }
    for name, (m, want) in shapes.items():
        p = E.Plan(m)
        got = set()
        for d in (M.FORWARD, M.BACKWARD):
            for mode in MODES:
                c = p.schedule_classes(d, mode)
                # the random graph's hubs (more than 32 active messages in the uniform modes, where every message is
                # received AND sent) take the streaming kernel, alone: a few per cent of the records
                if name.startswith("C4") and mode in (M.REPAM_UNIFORM, M.REPAM_DAMPED_UNIFORM):
                    assert c.pop("dense_big", 0) <= 0.05 * sum(c.values()), (name, c)
                got |= set(c)
        assert got == want, (name, got)


def _hub_model(L, n_spokes, pairwise="dense", seed=5):
    """a star: one hub variable in the MIDDLE of the order with n_spokes neighbours, plus a ring over the spokes"""
    rng = np.random.default_rng(seed)
    n = n_spokes + 1
    hub = n_spokes // 2
    others = [v for v in range(n) if v != hub]
    ei = [min(hub, v) for v in others] + [min(others[k], others[(k + 1) % len(others)]) for k in range(len(others))]
    ej = [max(hub, v) for v in others] + [max(others[k], others[(k + 1) % len(others)]) for k in range(len(others))]
    ei, ej = np.array(ei), np.array(ej)
    kw = dict(tables=rng.random((len(ei), L, L))) if pairwise == "dense" else dict(potts=rng.random(len(ei)))
    return S.mrf_model(n, L, ei, ej, rng.random(n * L), **kw)


def test_a_hub_with_more_messages_than_a_packet_slab_holds_gets_a_class_of_its_own():
    """C4's random graph has a few variables with ~30 neighbours among 2 M: such a record exceeds the LDS slab of the packed
    kernels' lane groups (pk_indirect_cap).  It must go to the op-by-op streaming class ALONE — left in the exact class it
    took its whole launch to the unpacked first-version kernel (half of C4's pass time in round 2)."""
    from lp_mp_amd import engine as E
    for pairwise, exact in (("dense", "dense16"), ("potts", "potts16")):
        m = _hub_model(16, 40, pairwise)
        p = E.Plan(m)
        for mode in (M.REPAM_UNIFORM, M.REPAM_ANISOTROPIC):
            for d in (M.FORWARD, M.BACKWARD):
                c = p.schedule_classes(d, mode)
                if mode == M.REPAM_UNIFORM:       # the hub receives 40 and sends 40 messages: 80 ops > 32
                    assert c == {exact: 40, "dense_big": 1}, (pairwise, mode, d, c)
                else:
                    assert set(c) <= {exact, "dense_big"} and c.get("dense_big", 0) <= 1, c
        m8 = _hub_model(16, 12, pairwise)          # 24 ops: fits
        assert set(E.Plan(m8).schedule_classes(M.FORWARD, M.REPAM_UNIFORM)) == {exact}


def test_chain_plans_default_to_the_packed_classes(monkeypatch):
    """deep schedules of the packed dense / Potts classes become chain launches (tickets + flags); many tiny levels of the
    lane-per-factor class become the level loop (one workgroup, no tickets); the ticket form of that class only on
    request (LPMP_CHAIN_ALL=1: measured slower than graph replay, plan.cpp make_schedule)"""
    assert E.Plan(S.grid_model(40, 30, 8, order="row_major")).chain_info(M.FORWARD, M.REPAM_ANISOTROPIC)["n_chains"] == 1
    m = S.c5_model(24, 24, 8, 400, 300, 100, seed=5, window=16)
    ci = E.Plan(m).chain_info(M.BACKWARD, M.REPAM_ANISOTROPIC)
    assert ci["n_chains"] == 1 and ci["n_tickets"] == 0 and ci["n_plain_launches"] == 2          # level loop + the two Potts steps
    monkeypatch.setenv("LPMP_NO_LEVEL_LOOP", "1")
    assert E.Plan(m).chain_info(M.BACKWARD, M.REPAM_ANISOTROPIC)["n_chains"] == 0
    monkeypatch.delenv("LPMP_NO_LEVEL_LOOP")
    monkeypatch.setenv("LPMP_CHAIN_ALL", "1")
    ci = E.Plan(m).chain_info(M.BACKWARD, M.REPAM_ANISOTROPIC)
    assert ci["n_chains"] == 1 and ci["n_plain_launches"] == 2 and ci["n_dependencies"] >= ci["n_tickets"] - 1 > 0


def test_mailbox_covers_the_hand_overs_of_a_deep_dense_chain(monkeypatch):
    """plan.cpp: in a deep chain of an exact dense class a receive polls the mailbox row its neighbour's send writes, and the
    dependency between the two tickets is dropped; what a granule cannot vouch for keeps its flag (the own factor's previous
    update in the fused pass); short schedules and LPMP_NO_MAILBOX=1 keep every flag"""
    monkeypatch.delenv("LPMP_NO_MAILBOX", raising=False)
    m = S.grid_model(40, 30, 8, order="row_major")
    f = E.Plan(m).chain_info(M.FORWARD, M.REPAM_ANISOTROPIC)
    # 40 x 30 grid: every edge is received through exactly once per sweep -> one row per edge
    n_edges = 40 * 29 + 39 * 30
    assert f["n_chains"] == 1 and f["mailbox_rows"] == n_edges and f["mailbox_receives"] == n_edges and f["n_dependencies"] == 0, f
    p = E.Plan(m).chain_info(-1, M.REPAM_ANISOTROPIC)
    assert p["mailbox_rows"] == 2 * n_edges and 0 < p["n_dependencies"] < p["n_tickets"], p      # forward record -> backward record of the same variable
    monkeypatch.setenv("LPMP_NO_MAILBOX", "1")
    g = E.Plan(m).chain_info(M.FORWARD, M.REPAM_ANISOTROPIC)
    assert g["mailbox_rows"] == 0 and g["n_dependencies"] >= g["n_tickets"] - 1, g
    monkeypatch.delenv("LPMP_NO_MAILBOX")
    q = E.Plan(S.grid_model(40, 30, 8, order="row_major", pairwise="potts")).chain_info(M.FORWARD, M.REPAM_ANISOTROPIC)
    assert q["mailbox_rows"] == n_edges and q["n_dependencies"] == 0, q                            # the exact Potts classes too
    for padded in (S.grid_model(40, 30, 7, order="row_major"), S.grid_model(40, 30, 7, order="row_major", pairwise="potts")):
        assert E.Plan(padded).chain_info(M.FORWARD, M.REPAM_ANISOTROPIC)["mailbox_rows"] == n_edges                # ... and the run-time-dims classes
    assert E.Plan(S.grid_model(40, 30, 8, order="colour_major")).chain_info(M.FORWARD, M.REPAM_ANISOTROPIC)["mailbox_rows"] == 0


def test_pass_rotation_is_decided_op_by_op():
    """n passes as H, W, (K, W)^(n-1), T need K = (receives of T, sends of H) and W = (receives of T', sends of H')
    record by record (engine.cpp, plan_rotation): checkerboard grids in colour-major order qualify under anisotropic
    weights, row-major orders and the uniform modes (every message both ways in each sweep) do not"""
    for pw in ("dense", "potts"):
        p = E.Plan(S.grid_model(6, 7, 8, pairwise=pw, order="colour_major"))
        assert p.pass_rotates(M.REPAM_ANISOTROPIC) and p.pass_schedule_info(M.REPAM_ANISOTROPIC)["n_levels"] == 3
        assert not p.pass_rotates(M.REPAM_UNIFORM)
        assert not E.Plan(S.grid_model(6, 7, 8, pairwise=pw, order="row_major")).pass_rotates(M.REPAM_ANISOTROPIC)
    # random bipartite graphs, colour-major: whatever is decided, a joined schedule must consist of the unfused ops
    import sys, os
    sys.path.insert(0, os.path.dirname(__file__))
    from test_fuzz_gpu import random_bipartite_mrf
    n_rot = 0
    for seed in range(60):
        m = random_bipartite_mrf(np.random.default_rng(17000 + seed))
        p = E.Plan(m)
        for mode in (M.REPAM_ANISOTROPIC, M.REPAM_ANISOTROPIC2):
            if p.pass_rotates(mode):
                n_rot += 1
                fb = p.pass_schedule_info(mode)
                one = [p.schedule_info(d, mode) for d in (0, 1)]
                assert fb["n_receives"] == one[0]["n_receives"] + one[1]["n_receives"]
                assert fb["n_sends"] == one[0]["n_sends"] + one[1]["n_sends"]
    assert n_rot > 0


def test_partitions_and_batch_weights_match_oracle():
    """construct_factor_partition (reference LP_MP.h:1717-1760: union-find components in root order, updated factors
    only, insertion order inside) — engine host analysis against the oracle's restatement on random models"""
    import sys, os
    sys.path.insert(0, os.path.dirname(__file__))
    from test_fuzz_gpu import random_model
    for seed in range(40):
        rng = np.random.default_rng(21000 + seed)
        m = random_model(rng)
        n = m.n_factors
        m.part_pairs = rng.integers(0, n, size=(int(rng.integers(0, 2 * n)), 2)).astype(np.int32)
        pe, po = E.Plan(m).partitions(), Oracle(m).partitions()
        assert len(pe) == len(po) and all(np.array_equal(a, b) for a, b in zip(pe, po))
        upd = set(Oracle(m).update_order(0).tolist())
        assert sorted(np.concatenate(pe).tolist()) == sorted(upd)
    bad = S.grid_model(3, 3, 2)
    bad.part_pairs = np.array([[0, 99]], np.int32)
    with pytest.raises(E.EngineError):
        E.Plan(bad)
    bad = S.grid_model(3, 3, 2)
    bad.mtypes[0].flags = 64
    with pytest.raises(E.EngineError):
        E.Plan(bad)


def test_oracle_partition_and_adaptive_rules_are_dual_ascent():
    """properties of the oracle's restatement that need no reference value: every rule only reparametrises (the energy
    of a fixed labeling is unchanged) and never lowers the bound; the improvement op is non-negative and zero for a
    message whose sender has nothing to give"""
    H, W, L = 5, 6, 4
    mts = S.mrf_mtypes()
    for t in mts:
        t.flags = M.MF_IMPROVEMENT
    var = S.grid_variable_order(H, W, "row_major").reshape(-1)
    a, bb = S.grid_edges(H, W)
    b = M.ModelBuilder(2, mts)
    un = S.u01(H * W * L, 7).reshape(-1, L)
    tb = S.u01(len(a) * L * L, 8).reshape(-1, L, L)
    u = b.add_vector_factors(0, un)
    p = b.add_dense_pairwise(1, tb)
    b.add_interleaved_messages(np.tile(np.array([0, 1], np.int32), len(a)), np.stack([u[a], u[bb]], 1).reshape(-1), np.repeat(p, 2))
    b.add_relations(np.stack([u[a], p], 1).reshape(-1), np.stack([p, u[bb]], 1).reshape(-1))
    for k in range(len(a)):
        if (a[k] % W) // 3 == (bb[k] % W) // 3:
            b.put_in_same_partition(u[a[k]], u[bb[k]])
    m = b.finish()
    x = np.random.default_rng(0).integers(0, L, H * W)

    def energy(d):
        th = d[: H * W * L].reshape(-1, L)
        pw = d[H * W * L:].reshape(-1, 2 * L)
        e = th[np.arange(H * W), x].sum()
        for k in range(len(a)):
            e += tb[k, x[a[k]], x[bb[k]]] + pw[k, x[a[k]]] + pw[k, L + x[bb[k]]]
        return e
    for rtype in (M.RTYPE_PARTITION, M.RTYPE_OVERLAPPING_PARTITION, M.RTYPE_ADAPTIVE):
        o = Oracle(m)
        o.set_reparametrization_type(rtype); o.set_reparametrization(M.REPAM_ANISOTROPIC)
        e0, lb = energy(o.duals()), o.LowerBound()
        for _ in range(4):
            o.ComputePass(1)
            assert o.LowerBound() >= lb - 1e-9
            lb = o.LowerBound()
            assert abs(energy(o.duals()) - e0) <= 1e-9
        assert lb <= e0 + 1e-9 and lb > 0


def test_update_levels_without_planning_the_sweep_equal_those_of_the_planned_sweep():
    """lpmp_plan_get_update_levels on a sweep nobody has planned (the global structure of a lock-step run: its sweeps are never
    executed as such) computes the levels alone; the same numbers as read off the full schedule, every mode, both directions"""
    from lp_mp_amd import synthetic as S
    models = [S.grid_model(9, 8, 4, order="row_major", seed=1), S.counter_graph_model(500, 2000, 4, 2),
              S.c5_model(8, 8, 4, 200, 120, 40, seed=3, window=16), S.multicut_triangle_model(12, 15, seed=4)]
    for m in models:
        for mode in (M.REPAM_ANISOTROPIC, M.REPAM_ANISOTROPIC2, M.REPAM_UNIFORM, M.REPAM_DAMPED_UNIFORM):
            p1, p2 = E.Plan(m), E.Plan(m)
            for d in (M.FORWARD, M.BACKWARD):
                alone = p1.update_levels(d, mode)
                p2.schedule_info(d, mode)
                assert np.array_equal(alone, p2.update_levels(d, mode)), (mode, d)
                assert alone.max() <= p2.schedule_info(d, mode)["n_levels"]          # (updates without an active message count no level)
