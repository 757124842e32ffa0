"""The one piece of the reference's hot path that builds here from its own sources (oracle/build_ref.py -> oracle/_ref/): the CSR
container `two_dim_variable_array` behind `weight_array` / `receive_array` (reference include/two_dimensional_variable_array.hxx,
used at include/LP_MP.h:989-992, sized by allocate_omega :1008-1040).  The reference's own test of it must pass as built, and the
layout the real container reports is what lpmp_plan_get_omega / _mask hand out — a check against reference CODE, not
against a restatement.  Skipped where neither /root/reference nor a previously built oracle/_ref is present."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lp_mp_amd import engine as E, model as M, synthetic as S  # noqa: E402
from oracle import build_ref  # noqa: E402


@pytest.fixture(scope="module")
def ref():
    exes = build_ref.build()
    if "ref_two_dim" not in exes:
        pytest.skip("no reference sources and no oracle/_ref build on this box")
    return exes


def _layout(exe, sizes):
    out = subprocess.check_output([exe], input=" ".join(map(str, [len(sizes)] + list(sizes))), text=True, timeout=120)
    rows, flat = [], []
    for line in out.splitlines():
        w = line.split()
        if w[0] == "rows":
            n = int(w[1])
        elif w[0] == "row":
            rows.append((int(w[3]), int(w[5])))
        else:
            flat.append((int(w[1]), int(w[2])))
    assert n == len(rows)
    return rows, flat


def test_the_references_own_container_test_passes_as_built(ref):
    if "ref_test_two_dimensional_variable_array" not in ref:
        pytest.skip("the reference's test was not built")
    assert subprocess.run([ref["ref_test_two_dimensional_variable_array"]], timeout=300).returncode == 0


def test_reference_container_is_a_row_major_csr(ref):
    rng = np.random.default_rng(5)
    for _ in range(20):
        sizes = rng.integers(0, 9, int(rng.integers(1, 40)))
        sizes[rng.integers(0, sizes.shape[0])] = 0                      # rows without entries (factors that send nothing)
        if sizes.sum() == 0:
            continue
        rows, flat = _layout(ref["ref_two_dim"], sizes)
        off = np.concatenate([[0], np.cumsum(sizes)])
        first = int(off[np.nonzero(sizes)[0][0]])
        assert [r[0] for r in rows] == list(map(int, sizes))
        assert all(o == off[i] - first for i, (s, o) in enumerate(rows) if s > 0)
        assert flat == [(i, j) for i, s in enumerate(sizes) for j in range(int(s))]


@pytest.mark.parametrize("mode", [M.REPAM_ANISOTROPIC, M.REPAM_DAMPED_UNIFORM])
def test_plan_weight_and_mask_arrays_have_the_layout_of_the_reference_container(ref, mode):
    """lpmp_plan_get_omega / _mask (what the engine's sweeps and every custom pass index): row i = the i-th updated factor, entries
    in message-list order, rows stored back to back — the reference container given the same row sizes reports the same offsets and
    the same storage order"""
    models = [S.grid_model(4, 5, 3, seed=2), S.counter_graph_model(60, 150, 4, 3), S.multicut_triangle_model(12, 15, seed=4)]
    for gm in models:
        p = E.Plan(gm)
        for d in (M.FORWARD, M.BACKWARD):
            for off, data in (p.omega(d, mode), p.mask(d, mode)):
                sizes = np.diff(off)
                assert off[0] == 0 and data.shape[0] == off[-1]
                if sizes.sum() == 0:
                    continue
                rows, flat = _layout(ref["ref_two_dim"], sizes)
                first = int(off[np.nonzero(sizes)[0][0]])
                assert all(o == off[i] - first for i, (s, o) in enumerate(rows) if s > 0)
                # storage position k of the reference holds (row, column) = what data[k] holds here
                pos = np.repeat(np.arange(sizes.shape[0]), sizes), np.concatenate([np.arange(s) for s in sizes])
                assert flat == list(zip(map(int, pos[0]), map(int, pos[1])))


def test_partitions_are_numbered_as_the_references_union_find_numbers_them(ref):
    """a19: LP::construct_factor_partition (reference LP_MP.h:1724-1745) on the reference's real union_find — merge order, root
    choice and get_contiguous_ids decide which partition is swept first.  lpmp_plan_get_partitions and the C oracle against it, on
    random put_in_same_partition graphs over grid and multicut models (pairs over all factors, also the never-updated ones)"""
    from oracle.binding import Oracle
    if "ref_union_find" not in ref:
        pytest.skip("ref_union_find was not built")
    rng = np.random.default_rng(9)
    for trial in range(12):
        gm = S.grid_model(int(rng.integers(2, 6)), int(rng.integers(2, 6)), 3, seed=trial) if trial % 2 == 0 else S.multicut_triangle_model(10, int(rng.integers(4, 14)), seed=trial)
        nf = gm.n_factors
        gm.part_pairs = rng.integers(0, nf, size=(int(rng.integers(0, 2 * nf)), 2)).astype(np.int32)
        p = E.Plan(gm)
        updated = np.zeros(nf, np.int64); updated[p.update_order(M.FORWARD)] = 1
        text = " ".join(map(str, [nf] + list(updated) + [gm.part_pairs.shape[0]] + list(gm.part_pairs.reshape(-1))))
        out = subprocess.check_output([ref["ref_union_find"]], input=text, text=True, timeout=60).splitlines()
        P = int(out[0].split()[1])
        want = [[] for _ in range(P)]
        for line in out[1:]:
            w = line.split()
            want[int(w[3])].append(int(w[1]))
        got = [list(map(int, a)) for a in p.partitions()]
        assert got == want, trial                                       # same numbering, members in factor order
        o = Oracle(gm)
        assert [list(map(int, a)) for a in o.partitions()] == want, trial


# ---- a13: the scalar two-minimum behind the O(L) Potts message, from the reference's own help_functions.hxx ------------------------
def _ref_two_smallest(exe, vectors):
    def tok(x):
        return "inf" if x == np.inf else "-inf" if x == -np.inf else float(x).hex()
    text = "\n".join(" ".join([str(len(v))] + [tok(x) for x in v]) for v in vectors)
    out = subprocess.check_output([exe], input=text, text=True, timeout=120).split()
    return [(float.fromhex(a) if "inf" not in a else float(a), float.fromhex(b) if "inf" not in b else float(b)) for a, b in zip(out[0::2], out[1::2])]


def _potts_vectors(rng, n_cases, with_inf):
    """message vectors on an exact grid (multiples of 1/8: every sum below is exact whatever the order of the additions), with ties,
    label counts 1, 2 and non-powers of two, optionally +-inf entries"""
    cases = []
    for k in range(n_cases):
        L = [1, 2, 3, 5, 8, 13, 32, 33, 64, 100][k % 10]
        m2 = rng.integers(-40, 40, L) / 8.0
        if k % 3 == 0 and L > 1:
            m2[rng.integers(0, L)] = m2.min()                           # a tie for the minimum
        if k % 4 == 1 and L > 2:
            m2[:] = m2[0]                                               # all equal
        if with_inf and k % 5 == 2 and L > 2:
            m2[rng.integers(0, L)] = np.inf
        if with_inf and k % 7 == 3 and L > 2:
            m2[rng.integers(0, L)] = -np.inf
        cases.append((L, rng.integers(-40, 40, L) / 8.0, m2, float(rng.integers(-16, 24)) / 8.0))
    return cases


def _potts_message_from_reference_two_minimum(m1, m2, d, two):
    s, s2 = two
    other = np.where(m2 == s, s2, s)                                    # the smallest m2[x2] over x2 != x1
    return m1 + np.minimum(m2, d + other)


def test_oracle_potts_message_equals_the_references_two_smallest_elements(ref):
    """the oracle builds the Potts table literally (diff * [a != b], as the reference's test/potts_factor.cpp does) and minimises
    over it; the O(L) form the device uses needs exactly the smallest and second smallest entry — taken here from the reference's
    own two_smallest_elements (help_functions.hxx:106-120) compiled where it lies, incl. ties, L = 1, 2 and +-inf entries"""
    from oracle.binding import Oracle
    from tests.test_oracle_kat import _single_pairwise
    if "ref_two_smallest" not in ref:
        pytest.skip("ref_two_smallest was not built")
    cases = _potts_vectors(np.random.default_rng(21), 60, with_inf=True)
    twos = _ref_two_smallest(ref["ref_two_smallest"], [c[2] for c in cases])
    for (L, m1, m2, d), two in zip(cases, twos):
        srt = np.sort(m2)
        assert two == (srt[0], srt[1] if L > 1 else np.inf)             # what the reference function returns
        m, _ = _single_pairwise(None, m1, m2, potts=d, L=L)
        got = Oracle(m).message_value(0, True)
        assert np.array_equal(got, _potts_message_from_reference_two_minimum(m1, m2, d, two)), (L, d)


@pytest.mark.gpu
def test_device_two_min_equals_the_references_two_smallest_elements(ref):
    """the device's Potts receive (kernels.hip: two-smallest butterfly over the lanes of a unary, +inf in the padding lanes)
    against the reference's two_smallest_elements on the same vectors: ties, L = 1, 2, label counts that fill a wave only partly"""
    from tests.test_oracle_kat import _single_pairwise
    if "ref_two_smallest" not in ref:
        pytest.skip("ref_two_smallest was not built")
    cases = _potts_vectors(np.random.default_rng(22), 50, with_inf=False)
    twos = _ref_two_smallest(ref["ref_two_smallest"], [c[2] for c in cases])
    eng = E.Engine(0)
    for (L, m1, m2, d), two in zip(cases, twos):
        m, p = _single_pairwise(None, m1, m2, potts=d, L=L)
        eng.upload(m)
        # unary 0 receives its message from the pairwise factor (mask 1, no send): theta_0 = 0 + message
        eng.compute_pass_custom(np.array([0], np.int32), [0, 1], [0.0], [0, 1], [1])
        got = eng.download_duals()[:L]
        assert np.array_equal(got, _potts_message_from_reference_two_minimum(m1, m2, d, two)), (L, d)
    eng.close()
