"""The compile-time tables of the throughput solve's sparse factorization (window_solve.hip, chol_regs; round 6) against a numpy statement.

The kernel eliminates the speed-bias blocks last frame first, then the poses, then the right-hand side, and only ever touches the 16 x 16 tiles
its table TPP.nz names.  A tile missing from that table would be a silently dropped part of the factor, so the table is stated a second time
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


def test_elimination_order_is_a_permutation_speed_biases_last_frame_first():
    _, _, _, perm = tables()
    assert sorted(perm[:NF]) == list(range(NF)) and perm[NF] == NF
    assert list(perm[:9]) == list(range(NP_ + 90, NP_ + 99)) and list(perm[90:99]) == list(range(NP_, NP_ + 9))
    assert list(perm[99:NF]) == list(range(NP_))


def test_tile_tables_cover_the_scalar_factor():
    h, nz, owner, perm = tables()
    H = scalar_pattern()[np.ix_(perm[:NF], perm[:NF])]
    Lp = symbolic_cholesky(H)
    Ha = np.zeros((16 * T, 16 * T), bool)
    La = np.zeros((16 * T, 16 * T), bool)
    Ha[:NF, :NF], La[:NF, :NF] = H, Lp
    Ha[NF, : NF + 1] = Ha[: NF + 1, NF] = True  # the right-hand side rides along as position NF
    La[NF, : NF + 1] = True
    ht = np.array([[Ha[16 * i : 16 * i + 16, 16 * k : 16 * k + 16].any() for i in range(T)] for k in range(T)])
    lt = np.array([[La[16 * i : 16 * i + 16, 16 * k : 16 * k + 16].any() for i in range(T)] for k in range(T)])
    up = np.triu(np.ones((T, T), bool))
    assert np.array_equal(h, ht & up), "tiles the assembled system reaches"
    assert np.array_equal(nz, lt & up), "tiles of the factor (upper tile (k, i) = L(i, k)^T)"
    # what the order buys: tile updates U(j, i) -= W(k, j)^T W(k, i) per factorization, against the dense grid's
    upd = sum(n * (n + 1) // 2 for n in (int(nz[k, k + 1 :].sum()) for k in range(T)))
    assert upd == 115 and sum(n * (n + 1) // 2 for n in range(T)) == 220
    assert int(nz.sum()) == 51
    # every wavefront's tiles fit its registers (8 VGPRs per tile, the accumulators of the back substitution beside them)
    per_wave = [int(sum(nz[: i + 1, i].sum() for i in range(T) if owner[i] == w)) for w in range(4)]
    assert max(per_wave) <= 15 and sum(per_wave) == 51
