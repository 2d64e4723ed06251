"""Shared by tests/test_generator_distributions_*.py: the statistics tests/golden/make_golden.py::distributions() records
over the REFERENCE env's resets, recomputed from arrays (maps, spawns, goals, scripted-target plans) of any generator, and the
comparison against the fixture.

Comparison: two-sample chi-square per histogram, X2 = sum (sqrt(N2/N1) a_i - sqrt(N1/N2) b_i)^2 / (a_i + b_i). The counts are
not independent draws (exactly K cells per Block map, walls of a Maze come in chains), so the statistic's noise scale is taken
from the reference itself: the fixture stores two halves generated on different seeds, and a candidate passes when its X2
against the whole reference stays below max(BOUND x max(X2(half A, half B), degrees of freedom), the chi-square quantile at
p = 1e-4)
(measured on 2 x 16 384 device episodes per case: X2 between 0.3 and 1.8 of that scale; a uniform spawn offset, another density
law or a 1..10 Ram plan length are rejected: test_the_bound_rejects_generators_that_draw_something_else)."""
import os

import numpy as np
from scipy.stats import chi2 as _chi2

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
BOUND = 2.5


def load_reference(case):
    z = np.load(os.path.join(GOLDEN, "distributions.npz"))
    halves = [{k.split("/")[-1]: z[k] for k in z.files if k.startswith("%s/h%d/" % (case, h))} for h in range(2)]
    assert halves[0] and halves[1], case
    return halves


def chi2_two_sample(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    n1, n2 = a.sum(), b.sum()
    m = (a + b) > 0
    if n1 == 0 or n2 == 0:
        return 0.0 if n1 == n2 else float("inf"), int(m.sum())
    k1, k2 = np.sqrt(n2 / n1), np.sqrt(n1 / n2)
    return float((((k1 * a[m] - k2 * b[m]) ** 2) / (a[m] + b[m])).sum()), max(int(m.sum()) - 1, 1)


def collect(map_type, mode, maps, pos, goals, plan_len=None, first_act=None, plan_b=None, redrawn=None, path_len=None):
    """maps bool/u8 [n, S, S]; pos / goals int [n, 2, 2]; Ram: plan_len [n], first_act [n]; Nav: plan_b [n] bool, redrawn [n]
    bool, path_len [n] (cells), first_act [n]. Returns the dict of count arrays the fixture holds."""
    n = len(maps)
    S = 82 if map_type == "Block" else 81
    m = np.asarray(maps)[:, :S, :S] != 0
    st = dict(walls=np.zeros(64, np.int64), offs=np.zeros(4, np.int64), plan_len=np.zeros(32, np.int64),
              first_act=np.zeros(4, np.int64), flags=np.zeros(4, np.int64))
    k = m[:, 1:-1, 1:-1].reshape(n, -1).sum(1)
    st["walls"] = np.bincount(np.minimum(63, k // (15 if map_type == "Block" else 16)), minlength=64).astype(np.int64)
    st["wall_rows"], st["wall_cols"] = m.sum((0, 2)).astype(np.int64), m.sum((0, 1)).astype(np.int64)
    pos, goals = np.asarray(pos), np.asarray(goals)
    d = pos[:, 1] - pos[:, 0]
    assert ((d >= -1) & (d <= 0)).all(), "target spawn outside the 2x2 window up-left of the tracker"
    st["offs"] = np.bincount((d[:, 0] + 1) * 2 + (d[:, 1] + 1), minlength=4).astype(np.int64)
    st["tr_rows"] = np.bincount(pos[:, 0, 0], minlength=S).astype(np.int64)
    st["tr_cols"] = np.bincount(pos[:, 0, 1], minlength=S).astype(np.int64)
    st["g0_rows"] = np.bincount(goals[:, 0, 0], minlength=S).astype(np.int64)
    st["g1_cols"] = np.bincount(goals[:, 1, 1], minlength=S).astype(np.int64)
    if mode == "Ram":
        st["plan_len"] = np.bincount(np.asarray(plan_len), minlength=32).astype(np.int64)
        st["first_act"] = np.bincount(np.asarray(first_act), minlength=4).astype(np.int64)
    if mode == "Nav":
        pb = np.asarray(plan_b, bool)
        st["flags"][0] = int(pb.sum())
        st["flags"][1] = int(np.asarray(redrawn, bool).sum())
        st["plan_len"] = np.bincount(np.minimum(31, np.asarray(path_len)[~pb] // 8), minlength=32).astype(np.int64)
        st["first_act"] = np.bincount(np.asarray(first_act)[~pb], minlength=4).astype(np.int64)
    st["flags"][3] = n
    return st


def compare(case, st, keys=None, report=None):
    """Assert that the candidate's histograms are the reference's within the reference's own half-vs-half noise."""
    h0, h1 = load_reference(case)
    mode = case.split("_")[1]
    keys = keys or (["walls", "wall_rows", "wall_cols", "offs", "tr_rows", "tr_cols", "g0_rows", "g1_cols", "plan_len", "first_act"])
    worst = []
    for k in keys:
        ref = h0[k] + h1[k]
        if ref.sum() == 0:
            continue
        x_self, dof = chi2_two_sample(h0[k], h1[k])
        x, _ = chi2_two_sample(st[k], ref)
        # (small histograms: the chi-square quantile at p = 1e-4 is wider than BOUND x dof — 21.1 at 3 degrees of freedom)
        limit = max(BOUND * max(x_self, dof), float(_chi2.ppf(1.0 - 1e-4, dof)))
        worst.append((k, x, limit, dof))
        assert x <= limit, "%s %s: chi2 %.1f against the reference, limit %.1f (reference half-vs-half %.1f, dof %d)" % (
            case, k, x, limit, x_self, dof)
    if mode == "Nav":          # rare events: rates, not histograms
        n_ref = float(h0["flags"][3] + h1["flags"][3])
        for i, name in ((0, "plan B"), (1, "goal re-drawn")):
            r_ref = float(h0["flags"][i] + h1["flags"][i]) / n_ref
            r = float(st["flags"][i]) / float(st["flags"][3])
            tol = 4.0 * np.sqrt(max(r_ref, 1.0 / n_ref) * (1.0 / n_ref + 1.0 / float(st["flags"][3]))) + 1e-4
            assert abs(r - r_ref) <= tol, "%s %s rate %.5f, reference %.5f (tolerance %.5f)" % (case, name, r, r_ref, tol)
            worst.append((name, r, r_ref, 0))
    if report is not None:
        report.extend(worst)
    return worst
