"""The permutation solver of the stitching scan against the package the reference calls.

The reference aligns adjacent segments with ``scipy.optimize.linear_sum_assignment`` (training/losses.py:43; scipy pinned
1.11.4 in requirements.txt, 1.15 in this image -- the same ``rectangular_lsap`` algorithm since 1.6).  Every exact method
finds the same optimum; WHICH optimum on exact ties is a property of that algorithm, and exact ties are the normal case
in a meeting: two speakers silent through a whole overlap make all four costs between them exactly 0.  The oracle
(``css_oracle.lsap``) and the library (``css_pit_scan``: csrc/lsap.hpp, the code the device scan runs too) restate the
algorithm with its tie rules; here both are held to scipy itself on tie-laden matrices.  CPU only."""
import itertools

import numpy as np
import pytest
from scipy.optimize import linear_sum_assignment

import css_oracle as O
from conftest import pkg


def _cases(S, n, seed):
    rs = np.random.RandomState(seed)
    for trial in range(n):
        kind = trial % 5
        if kind == 0:                                  # small integers: ties everywhere
            c = rs.randint(0, 3, (S, S)).astype(np.float64)
        elif kind == 1:                                # 0 / x
            c = rs.randint(0, 2, (S, S)).astype(np.float64) * rs.rand()
        elif kind == 2:                                # duplicated rows and columns: speakers silent on either side
            c = rs.rand(S, S)
            c[rs.randint(0, S)] = c[rs.randint(0, S)]
            c[:, rs.randint(0, S)] = c[:, rs.randint(0, S)]
        elif kind == 3:                                # two speakers silent on BOTH sides: a zero block
            c = rs.rand(S, S)
            i, j = rs.choice(S, 2, replace=False)
            c[i] = c[j]
            c[:, i] = c[:, j]
            c[np.ix_([i, j], [i, j])] = 0.0
        else:                                          # generic float32 costs
            c = rs.rand(S, S).astype(np.float32).astype(np.float64)
        yield c


@pytest.mark.parametrize("S", [2, 3, 4])
def test_oracle_lsap_is_scipys(S):
    for c in _cases(S, 3000, 10 + S):
        assert O.lsap(c) == tuple(linear_sum_assignment(c)[1]), c


def test_lexicographic_search_is_not_scipys_on_ties():
    """why the restatement exists: the first minimum in lexicographic order (what rounds 1-3 shipped) keeps two silent
    speakers in place where scipy swaps them"""
    c = np.array([[0.3, 0.2, 0.2], [0.25, 0.0, 0.0], [0.25, 0.0, 0.0]])
    brute = min(itertools.permutations(range(3)), key=lambda s: sum(c[a, s[a]] for a in range(3)))
    assert brute == (0, 1, 2)
    assert tuple(linear_sum_assignment(c)[1]) == (0, 2, 1) == O.lsap(c)


@pytest.mark.parametrize("S", [2, 3, 4])
def test_library_scan_is_scipys_chained(S):
    """css_pit_scan (host form of the device scan): boundary b's rows are taken in the order of segment b's permutation
    (css.py:283-285 permutes the right segment in place, so it is the next boundary's left), each solved as scipy does"""
    L = pkg("_lib")
    costs = np.stack(list(_cases(S, 2000, 20 + S)))
    perms = L.pit_scan(costs, S)
    assert tuple(perms[0]) == tuple(range(S))
    for b in range(costs.shape[0]):
        want = linear_sum_assignment(costs[b][list(perms[b])])[1]
        assert tuple(perms[b + 1]) == tuple(want), (b, costs[b], perms[b])


def test_pit_wrapper_known_answer():
    """the reference's own known-answer test (losses.py:109-123): exact permutation, zero loss"""
    rs = np.random.RandomState(43236)
    t = rs.rand(100, 257, 4).astype(np.float32)
    p = (3, 0, 2, 1)
    loss, perm, _ = O.pit_perm(t[..., p], t, "mse")
    assert loss == 0.0 and perm == p


def test_non_finite_costs_terminate():
    """a pass that left the split-f16 range hands NaN costs to the scan before the float32 repeat: scipy raises on such a
    matrix ("infeasible"); the library's scan must terminate (it keeps the identity) -- on the device a loop here hangs the GPU"""
    L = pkg("_lib")
    for bad in (np.nan, np.inf, -np.inf):
        costs = np.full((3, 3, 3), bad)
        costs[1] = np.random.RandomState(0).rand(3, 3)
        costs[2, 0, 0] = 1.0
        perms = L.pit_scan(costs, 3)
        assert perms.shape == (4, 3) and all(sorted(p) == [0, 1, 2] for p in perms.tolist())
