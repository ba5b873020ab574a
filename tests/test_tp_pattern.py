"""The compile-time tables of the throughput solve's sparse factorization (window_solve.hip, chol_regs; round 6) against a numpy statement.

The kernel eliminates the speed-bias blocks of frames 10 .. 6 and, beside them, those of frames 4 .. 0 (two pivot chains at a time; each set padded to
three 16-column tiles), then frame 5's block, the poses and the right-hand side, and only ever touches the 16 x 16 tiles its table TPP.nz names.  A tile missing from that table would be a silently dropped part of the factor, so the table is stated a second time
here: the assembled system's SCALAR pattern (estimator.cpp:663-755: an IMU factor couples pose / speed-bias i to pose / speed-bias i + 1; the
prior, marginalization_factor.cpp:333-381, couples speed-bias 0 to every pose; the poses are dense after the Schur complement on the depths), its
scalar symbolic Cholesky factor in the kernel's order, aggregated to tiles.  CPU tier: the library's tables are constants of the build."""
import ctypes as C

import numpy as np

from conftest import mod

NP_, NFR, NF, T = 66, 11, 165, 11


def tables():
    L = mod("lib").lib()
    out = (C.c_int * 512)()
    L.avm_debug_solve_tp_pattern.argtypes = [C.POINTER(C.c_int)]
    n = L.avm_debug_solve_tp_pattern(out)
    a = np.array(out[:n])
    return a[:121].reshape(T, T).astype(bool), a[121:242].reshape(T, T).astype(bool), a[242:253], a[253:]


def scalar_pattern():
    """structural nonzeros of the reduced system, columns poses | speed-biases (the layout S_OFF addresses)"""
    H = np.zeros((NF, NF), bool)
    H[:NP_, :NP_] = True
    for i in range(NFR - 1):  # IMU factor i: pose i, speed-bias i, pose i + 1, speed-bias i + 1
        cols = list(range(6 * i, 6 * i + 12)) + list(range(NP_ + 9 * i, NP_ + 9 * i + 18))
        H[np.ix_(cols, cols)] = True
    H[NP_ : NP_ + 9, :NP_] = H[:NP_, NP_ : NP_ + 9] = True  # the prior: speed-bias 0 x every pose
    return H


def symbolic_cholesky(H):
    Lp = np.tril(H).copy()
    for k in range(len(H)):
        rows = np.nonzero(Lp[k + 1 :, k])[0] + k + 1
        for r in rows:
            Lp[r, rows[rows <= r]] = True
    return Lp


def expected_order():
    """positions of the elimination order -> columns of the assembled system (poses | speed-biases), -1 for padding, NF for the right-hand side"""
    sb = lambda f: list(range(NP_ + 9 * f, NP_ + 9 * f + 9))
    order = sum((sb(f) for f in (10, 9, 8, 7, 6)), []) + [-1] * 3
    order += sum((sb(f) for f in (4, 3, 2, 1, 0)), []) + [-1] * 3
    order += sb(5) + list(range(NP_)) + [NF] + [-1] * 4
    return np.array(order)


def test_elimination_order_two_chains_then_the_separator_and_the_poses():
    _, _, _, perm = tables()
    assert np.array_equal(perm, expected_order())
    assert sorted(perm[perm >= 0]) == list(range(NF + 1))


def test_tile_tables_cover_the_scalar_factor():
    h, nz, owner, perm = tables()
    N = 16 * T
    Ha = np.zeros((N, N), bool)
    H = np.zeros((NF + 1, NF + 1), bool)
    H[:NF, :NF] = scalar_pattern()
    H[NF, :] = H[:, NF] = True  # the right-hand side rides along
    H[NF, NF] = False
    real = perm >= 0
    Ha[np.ix_(real, real)] = H[np.ix_(perm[real], perm[real])]
    Ha[np.arange(N)[~real], np.arange(N)[~real]] = True  # padding positions: rows of the identity
    La = symbolic_cholesky(Ha)
    ht = np.array([[Ha[16 * i : 16 * i + 16, 16 * k : 16 * k + 16].any() for i in range(T)] for k in range(T)])
    lt = np.array([[La[16 * i : 16 * i + 16, 16 * k : 16 * k + 16].any() for i in range(T)] for k in range(T)])
    up = np.triu(np.ones((T, T), bool))
    assert np.array_equal(h, ht & up), "tiles the assembled system reaches"
    assert np.array_equal(nz, lt & up), "tiles of the factor (upper tile (k, i) = L(i, k)^T)"
    # the two chains never meet: tile columns 0..2 (frames 10..6) and 3..5 (frames 4..0) share no tile, so chains t and 3 + t run side by side
    assert not nz[:3, 3:6].any()
    # what the order buys: tile updates U(j, i) -= W(k, j)^T W(k, i) per factorization, against the dense grid's
    upd = sum(n * (n + 1) // 2 for n in (int(nz[k, k + 1 :].sum()) for k in range(T)))
    assert upd == 91 and sum(n * (n + 1) // 2 for n in range(T)) == 220
    assert int(nz.sum()) == 47
    # every wavefront's tiles fit its registers (8 VGPRs per tile, the accumulators of the back substitution beside them)
    per_wave = [int(sum(nz[: i + 1, i].sum() for i in range(T) if owner[i] == w)) for w in range(4)]
    assert max(per_wave) <= 15 and sum(per_wave) == 47
    # the two chains of a step run on two wavefronts
    assert all(owner[t] != owner[t + 3] for t in range(3))


def tables_of(which, T_):
    """avm_debug_solve_pattern: which = 0 throughput, 1 latency, 2 extended build"""
    L = mod("lib").lib()
    out = (C.c_int * 512)()
    L.avm_debug_solve_pattern.argtypes = [C.c_int, C.POINTER(C.c_int)]
    n = L.avm_debug_solve_pattern(which, out)
    a = np.array(out[:n])
    assert n == 2 * T_ * T_ + T_ + 16 * T_
    return a[: T_ * T_].reshape(T_, T_).astype(bool), a[T_ * T_ : 2 * T_ * T_].reshape(T_, T_).astype(bool), a[2 * T_ * T_ : 2 * T_ * T_ + T_], a[2 * T_ * T_ + T_ :]


def test_the_latency_build_factors_with_the_same_tables():
    """The latency form calls the same chol_regs: same order, same pattern (its offsets name places in the packed triangle), same owners."""
    for a, b in zip(tables(), tables_of(1, T)):
        assert np.array_equal(a, b)


def test_the_extended_builds_tables_cover_its_scalar_factor():
    """-DAVM_X: 79 dense columns (poses | relo_Pose | ex_pose | td) in front of the speed-biases, 178 columns, twelve tile columns, nine steps; every
    dense tile column on a wavefront of its own.  Same statement as above: the scalar pattern (the prior couples speed-bias 0 to EVERY dense column,
    ex_pose / td blocks included), its symbolic factor in the kernel's order, aggregated to tiles."""
    NPX, NFX, TX = 79, 178, 12
    h, nz, owner, perm = tables_of(2, TX)
    sb = lambda f: list(range(NPX + 9 * f, NPX + 9 * f + 9))
    order = sum((sb(f) for f in (10, 9, 8, 7, 6)), []) + [-1] * 3 + sum((sb(f) for f in (4, 3, 2, 1, 0)), []) + [-1] * 3
    order += sb(5) + list(range(NPX)) + [NFX] + [-1] * 7
    assert np.array_equal(perm, np.array(order)) and sorted(perm[perm >= 0]) == list(range(NFX + 1))
    H = np.zeros((NFX + 1, NFX + 1), bool)
    H[:NPX, :NPX] = True
    for i in range(NFR - 1):
        cols = list(range(6 * i, 6 * i + 12)) + list(range(NPX + 9 * i, NPX + 9 * i + 18))
        H[np.ix_(cols, cols)] = True
    H[NPX : NPX + 9, :NPX] = H[:NPX, NPX : NPX + 9] = True
    H[NFX, :] = H[:, NFX] = True
    H[NFX, NFX] = False
    N = 16 * TX
    Ha = np.zeros((N, N), bool)
    real = perm >= 0
    Ha[np.ix_(real, real)] = H[np.ix_(perm[real], perm[real])]
    Ha[np.arange(N)[~real], np.arange(N)[~real]] = True
    La = symbolic_cholesky(Ha)
    ht = np.array([[Ha[16 * i : 16 * i + 16, 16 * k : 16 * k + 16].any() for i in range(TX)] for k in range(TX)])
    lt = np.array([[La[16 * i : 16 * i + 16, 16 * k : 16 * k + 16].any() for i in range(TX)] for k in range(TX)])
    up = np.triu(np.ones((TX, TX), bool))
    # (the kernel's table may name a place the system never fills - it reads the packed triangle's zeros there -, never the other way round)
    assert not ((ht & up) & ~h).any() and not ((lt & up) & ~nz).any()
    assert np.array_equal(nz, lt & up), "tiles of the factor"
    assert not nz[:3, 3:6].any()
    per_wave = [int(sum(nz[: i + 1, i].sum() for i in range(TX) if owner[i] == w)) for w in range(8)]
    assert max(per_wave) <= 13 and sum(per_wave) == int(nz.sum())
    assert all(owner[t] != owner[t + 3] for t in range(3)) and len(set(owner[6:])) == 6
