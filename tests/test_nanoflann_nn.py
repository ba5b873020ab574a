"""B8 (feature_selector.cpp:380-459) against the REFERENCE'S OWN nearest-neighbour search: the vendored, header-only nanoflann is
the one piece of the reference that compiles in the build container.  tests/golden/gen_nanoflann_nn.cpp runs
it - adaptor, tree type, leaf size and search parameters as feature_selector.h:118-146 / feature_selector.cpp:424-455 - on
seeded clouds; tests/golden/nanoflann_nn.npz holds, for 10 500 queries, the index it returned, its squared distance and the depth
findNNDepth hands back.  Empty clouds (the "return 1.0" branch), clouds around the leaf size, queries ON cloud points, and
EXACT TIES (dyadic grid points, queries at midpoints and cell centres: two resp. four points at bit-identical distances).

What is asserted: on EVERY query - the 78 exact ties included - the oracle's and the GPU's search return nanoflann's depth bit for
bit, and the squared distance of the winner is bit-identical.  Which of several equidistant points wins is decided by the order in
which the tree's traversal meets them, so both restate the tree: oracle/fsel.hpp `KdIndex` (divideTree / middleSplit_ / planeSplit /
searchLevel as written in the header), csrc/fsel.hip `fsel_kdtree_kernel` + `kd_nearest` (the same tree built by one wavefront per
frame, searched with an explicit stack).  Depths in the fixture are distinct per cloud, so equal depth = equal index."""
import os

import numpy as np
import pytest

from helpers import synth

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "nanoflann_nn.npz"))
NC = int(GOLD["n_clouds"])
# the degenerate clouds (second set of the generator): duplicated points, clouds of one repeated point, collinear clouds, a complete
# dyadic grid, 300 points, queries far outside the bounding box and with NaN / infinite coordinates
GOLD2 = np.load(os.path.join(os.path.dirname(__file__), "golden", "nanoflann_nn2.npz"))


def _problems(GOLD=GOLD):
    """The fixture's clouds and queries as avm_fsel_batch frames: one frame per cloud, the queries as its candidates."""
    out = []
    for c in range(int(GOLD["n_clouds"])):
        xy, dep, q = GOLD[f"cloud_xy_{c}"], GOLD[f"cloud_depth_{c}"], GOLD[f"query_xy_{c}"]
        p = synth.make_fsel(1, horizon=3, n_cand=len(q), n_used=0, n_cloud=max(len(xy), 1), max_features=4)
        p.a["n_cloud"][0] = len(xy)
        p.a["cloud_xy"][0, :len(xy)] = xy
        p.a["cloud_depth"][0, :len(xy)] = dep
        p.a["cand_xy"][0, :len(q)] = q
        out.append(p)
    return out


def _check(depths_of):
    n_plain = n_tie = n_tie_other = n_tie_not_lowest = 0
    for c, p in enumerate(_problems()):
        xy, dep, q = GOLD[f"cloud_xy_{c}"], GOLD[f"cloud_depth_{c}"], GOLD[f"query_xy_{c}"]
        want, d2 = GOLD[f"nn_depth_{c}"], GOLD[f"nn_dist2_{c}"]
        got = depths_of(p)[0, :len(q)]
        if len(xy) == 0:
            assert (got == 1.0).all() and (want == 1.0).all()      # feature_selector.cpp:444
            n_plain += len(q)
            continue
        D = (q[:, None, 0] - xy[None, :, 0]) ** 2 + (q[:, None, 1] - xy[None, :, 1]) ** 2
        assert np.array_equal(D.min(1), d2)                         # nanoflann's L2_Simple distance, bit for bit
        tied = (D == d2[:, None]).sum(1) > 1
        assert len(np.unique(dep)) == len(dep)                      # (a depth identifies its point)
        assert np.array_equal(dep[GOLD[f"nn_index_{c}"]], want)
        assert np.array_equal(got[~tied], want[~tied]), c           # no tie: the reference's answer exactly
        for i in np.nonzero(tied)[0]:
            assert got[i] in dep[D[i] == d2[i]], (c, i)            # a tie: one of the tied points ...
        n_plain += int((~tied).sum())
        n_tie += int(tied.sum())
        n_tie_other += int((got[tied] != want[tied]).sum())
        n_tie_not_lowest += int((want[tied] != dep[np.argmin(D[tied], axis=1)]).sum()) if tied.any() else 0
    assert n_tie_other == 0                                         # ... and the one nanoflann's traversal meets first
    return n_plain, n_tie, n_tie_not_lowest


def _check2(depths_of):
    """Second set: every answer bit for bit; how many of them a lowest-index rule would have got wrong is reported."""
    n = n_tie = n_not_lowest = 0
    for c, p in enumerate(_problems(GOLD2)):
        xy, dep, q = GOLD2[f"cloud_xy_{c}"], GOLD2[f"cloud_depth_{c}"], GOLD2[f"query_xy_{c}"]
        want, d2, idx = GOLD2[f"nn_depth_{c}"], GOLD2[f"nn_dist2_{c}"], GOLD2[f"nn_index_{c}"]
        got = depths_of(p)[0, :len(q)]
        assert len(np.unique(dep)) == len(dep) and np.array_equal(dep[idx], want)
        fin = np.isfinite(q).all(1)
        with np.errstate(invalid="ignore", over="ignore"):
            D = (q[:, None, 0] - xy[None, :, 0]) ** 2 + (q[:, None, 1] - xy[None, :, 1]) ** 2
        assert np.array_equal(D[fin].min(1), d2[fin])
        assert (idx[~fin] == 0).all()                               # NaN / inf: nothing is "closer", ret_index stays 0
        bad = np.nonzero(got != want)[0]
        assert len(bad) == 0, (c, len(xy), bad[:8], q[bad[:8]])
        tied = fin & ((D == d2[:, None]).sum(1) > 1)
        n += len(q)
        n_tie += int(tied.sum())
        n_not_lowest += int((idx[tied] != np.argmin(D[tied], axis=1)).sum()) if tied.any() else 0
    return n, n_tie, n_not_lowest


def test_the_fixture_is_what_the_docstring_says():
    tot = sum(len(GOLD[f"query_xy_{c}"]) for c in range(NC))
    assert NC == 14 and tot >= 10000
    assert sorted({len(GOLD[f"cloud_xy_{c}"]) for c in range(NC)}) == [0, 1, 9, 10, 11, 64, 150]


def test_oracle_find_nn_depth_against_the_reference_nanoflann(oracle):
    n_plain, n_tie, n_other = _check(oracle.fsel_nn_depth)
    print(f"\n[nanoflann] oracle: {n_plain} queries without a tie identical; {n_tie} exact ties identical as well ({n_other} of them are not the lowest index)")
    assert n_plain >= 10000 and n_tie >= 50


@pytest.mark.gpu
def test_gpu_find_nn_depth_against_the_reference_nanoflann(selector):
    n_plain, n_tie, n_other = _check(selector.nn_depth)
    print(f"\n[nanoflann] gpu: {n_plain} queries without a tie identical; {n_tie} exact ties identical as well ({n_other} of them are not the lowest index)")
    assert n_plain >= 10000 and n_tie >= 50


def test_oracle_on_the_degenerate_clouds(oracle):
    n, n_tie, n_nl = _check2(oracle.fsel_nn_depth)
    print(f"\n[nanoflann] oracle, degenerate clouds: {n} queries identical, {n_tie} of them exact ties ({n_nl} not the lowest index)")
    assert n >= 6000 and n_tie >= 1000


@pytest.mark.gpu
def test_gpu_on_the_degenerate_clouds(selector):
    n, n_tie, n_nl = _check2(selector.nn_depth)
    print(f"\n[nanoflann] gpu, degenerate clouds: {n} queries identical, {n_tie} of them exact ties ({n_nl} not the lowest index)")
    assert n >= 6000 and n_tie >= 1000


def _cloud_problem(xy, dep, q):
    p = synth.make_fsel(1, horizon=3, n_cand=len(q), n_used=0, n_cloud=len(xy), max_features=4)
    p.a["n_cloud"][0] = len(xy)
    p.a["cloud_xy"][0, :len(xy)] = xy
    p.a["cloud_depth"][0, :len(xy)] = dep
    p.a["cand_xy"][0, :len(q)] = q
    return p


def _shapes():
    """Clouds beyond the fixtures, answered by the oracle's restatement (which the fixtures pin): a tree deeper than 64 levels - the device
    search keeps the recursion's state as one bit per level, 64 to a word - a window-sized cloud searched with 2000 queries, and the largest
    cloud the tree builder takes (4096 points: 115 KB of LDS)."""
    rng = np.random.default_rng(20240905)
    out = []
    n = 300                                            # geometric: every split cuts ONE point off -> depth ~ n - 10
    xy = np.stack([0.75 * 0.5 ** np.arange(n), np.zeros(n)], 1)[rng.permutation(n)]
    q = np.concatenate([rng.uniform(-0.1, 0.8, (400, 2)) * [1, 0.02], 10.0 ** rng.uniform(-89, 0, (300, 1)) * [1, 0], xy[:100], (xy[:100] + xy[100:200]) / 2])
    out.append(("deep tree (geometric cloud, 300 points)", xy, q))
    xy = rng.uniform(-0.8, 0.8, (150, 2)) * [1, 0.6]
    out.append(("window-sized cloud, 2000 queries", xy, rng.uniform(-1, 1, (2000, 2))))
    xy = np.round(rng.uniform(-0.8, 0.8, (4096, 2)) * 64) / 64       # 4096 points on a lattice of 103 x 103: hundreds of duplicates and ties
    out.append(("4096 points on a coarse lattice", xy, np.round(rng.uniform(-0.9, 0.9, (1500, 2)) * 128) / 128))
    return out


def test_oracle_tree_depth_of_the_geometric_cloud(oracle):
    """(what the GPU test below relies on: the geometric cloud really gives a tree deeper than 64 levels - checked on a Python statement of
    middleSplit_'s cut, not on the oracle's internals)"""
    name, xy, q = _shapes()[0]
    idx, depth = np.arange(len(xy)), 0
    lo, hi = xy.min(0), xy.max(0)
    while len(idx) > 10:                               # follow the larger child
        span = hi - lo
        d = int(np.argmax([np.ptp(xy[idx, k]) if span[k] > (1 - 1e-5) * span.max() else -1 for k in (0, 1)]))
        cut = min(max((lo[d] + hi[d]) / 2, xy[idx, d].min()), xy[idx, d].max())
        left, right = idx[xy[idx, d] < cut], idx[xy[idx, d] >= cut]
        if len(left) >= len(right):
            idx, hi = left, np.where(np.arange(2) == d, cut, hi)
        else:
            idx, lo = right, np.where(np.arange(2) == d, cut, lo)
        depth += 1
    assert depth > 64, depth


@pytest.mark.gpu
def test_gpu_against_the_oracle_on_deep_and_large_trees(selector, oracle):
    for name, xy, q in _shapes():
        dep = np.random.default_rng(len(xy)).permutation(len(xy)) + 2.0
        p = _cloud_problem(xy, dep, q)
        want = oracle.fsel_nn_depth(p)[0, :len(q)]
        got = selector.nn_depth(p)[0, :len(q)]
        assert not np.isnan(got).any(), name           # (NaN: the three widths of the device search disagreed)
        assert np.array_equal(got, want), (name, int((got != want).sum()))
        D = ((q[:, None, :] - xy[None, :, :]) ** 2).sum(2)
        print(f"\n[kd-tree] {name}: {len(q)} queries identical to the oracle's; {int(((D == D.min(1)[:, None]).sum(1) > 1).sum())} exact ties")


@pytest.mark.gpu
def test_a_cloud_above_the_tree_builder_s_limit_is_refused(selector):
    p = synth.make_fsel(1, horizon=3, n_cand=4, n_used=0, n_cloud=4097, max_features=2)
    with pytest.raises(Exception) as e:
        selector.nn_depth(p)
    assert "4096" in str(e.value)
