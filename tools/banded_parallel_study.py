"""Could the banded MINCO solve of an evaluation be done in parallel and still meet the per-evaluation parity bar?  (round-4 review,
item 5: "measure a parallel banded solve ... in the emulator first -- per-evaluation parity vs oracle must stay <= 1e-11 ... If cyclic
reduction loses accuracy on the T^5-scaled rows, say so with the numbers and stop.")

CPU study, numpy only.  The 6N x 6N system of MinJerkOpt<9>::generate (minco.hpp:838-900; the device's minco_generate_mw fills the same
band) is solved three ways for random piece durations and boundary / inner-point data of the benchmark's magnitudes:
  ref   the reference's banded LU WITHOUT pivoting, in its elimination order (banded_system.hpp:66-118), float64 -- what the oracle and
        the device do, bit for bit up to fused multiply-adds;
  bcr   block cyclic reduction over the N 6 x 6 diagonal blocks (rows 6k .. 6k+5 couple pieces k-1, k, k+1: block tridiagonal),
        log2 N levels, the 6 x 6 solves with partial pivoting, float64 -- the parallel algorithm the review names;
  exact the same system in 80-bit extended precision with partial pivoting (numpy longdouble), as the yardstick.
Printed per N: the condition number, |ref - exact| / |exact|, |bcr - exact| / |exact| and |bcr - ref| / |ref| -- the last one is what
a parity test of the evaluation would see, because the oracle solves by `ref`.
"""
import numpy as np


def minco_matrix(T):
    N = len(T)
    n = 6 * N
    A = np.zeros((n, n))
    A[0, 0] = 1.0; A[1, 1] = 1.0; A[2, 2] = 2.0
    for i in range(N - 1):
        T1 = T[i]; T2 = T1 * T1; T3 = T2 * T1; T4 = T2 * T2; T5 = T4 * T1
        r = 6 * i
        A[r + 3, r + 3] = 6.0; A[r + 3, r + 4] = 24.0 * T1; A[r + 3, r + 5] = 60.0 * T2; A[r + 3, r + 9] = -6.0
        A[r + 4, r + 4] = 24.0; A[r + 4, r + 5] = 120.0 * T1; A[r + 4, r + 10] = -24.0
        A[r + 5, r:r + 6] = [1.0, T1, T2, T3, T4, T5]
        A[r + 6, r:r + 6] = [1.0, T1, T2, T3, T4, T5]; A[r + 6, r + 6] = -1.0
        A[r + 7, r + 1:r + 6] = [1.0, 2 * T1, 3 * T2, 4 * T3, 5 * T4]; A[r + 7, r + 7] = -1.0
        A[r + 8, r + 2:r + 6] = [2.0, 6 * T1, 12 * T2, 20 * T3]; A[r + 8, r + 8] = -2.0
    T1 = T[-1]; T2 = T1 * T1; T3 = T2 * T1; T4 = T2 * T2; T5 = T4 * T1
    R0 = n
    A[R0 - 3, R0 - 6:R0] = [1.0, T1, T2, T3, T4, T5]
    A[R0 - 2, R0 - 5:R0] = [1.0, 2 * T1, 3 * T2, 4 * T3, 5 * T4]
    A[R0 - 1, R0 - 4:R0] = [2.0, 6 * T1, 12 * T2, 20 * T3]
    return A


def solve_ref(A, b):
    """banded_system.hpp: factorizeLU without pivoting (lower / upper bandwidth 6) + solve, plain loops in the reference's order"""
    A = A.copy(); x = b.copy(); n = len(b); bw = 6
    for k in range(n - 1):
        iM = min(k + bw, n - 1)
        cVl = A[k, k]
        for i in range(k + 1, iM + 1):
            if A[i, k] != 0.0:
                A[i, k] /= cVl
        jM = min(k + bw, n - 1)
        for j in range(k + 1, jM + 1):
            cVl = A[k, j]
            if cVl != 0.0:
                for i in range(k + 1, iM + 1):
                    if A[i, k] != 0.0:
                        A[i, j] -= A[i, k] * cVl
    for j in range(n):
        iM = min(j + bw, n - 1)
        for i in range(j + 1, iM + 1):
            if A[i, j] != 0.0:
                x[i] -= A[i, j] * x[j]
    for j in range(n - 1, -1, -1):
        x[j] /= A[j, j]
        iM = max(0, j - bw)
        for i in range(iM, j):
            if A[i, j] != 0.0:
                x[i] -= A[i, j] * x[j]
    return x


def solve_bcr(A, b):
    """block cyclic reduction, 6 x 6 blocks, recursive: eliminate the odd-numbered block rows, solve the half-size system, back-substitute"""
    N = len(b) // 6
    D = [A[6 * k:6 * k + 6, 6 * k:6 * k + 6].copy() for k in range(N)]
    L = [A[6 * k:6 * k + 6, 6 * k - 6:6 * k].copy() if k > 0 else None for k in range(N)]
    U = [A[6 * k:6 * k + 6, 6 * k + 6:6 * k + 12].copy() if k < N - 1 else None for k in range(N)]
    f = [b[6 * k:6 * k + 6].copy() for k in range(N)]

    def rec(D, L, U, f):
        n = len(D)
        if n == 1:
            return [np.linalg.solve(D[0], f[0])]
        ev = list(range(0, n, 2)); od = list(range(1, n, 2))
        D2, L2, U2, f2 = [], [], [], []
        inv = {k: np.linalg.inv(D[k]) for k in od}
        for k in ev:
            Dk = D[k].copy(); fk = f[k].copy(); Lk = None; Uk = None
            if k - 1 >= 0:
                G = L[k] @ inv[k - 1]
                Dk -= G @ U[k - 1]; fk -= G @ f[k - 1]
                if k - 2 >= 0: Lk = -G @ L[k - 1]
            if k + 1 < n:
                G = U[k] @ inv[k + 1]
                Dk -= G @ L[k + 1]; fk -= G @ f[k + 1]
                if k + 2 < n: Uk = -G @ U[k + 1]
            D2.append(Dk); L2.append(Lk); U2.append(Uk); f2.append(fk)
        xe = rec(D2, L2, U2, f2)
        x = [None] * n
        for t, k in enumerate(ev): x[k] = xe[t]
        for k in od:
            r = f[k] - L[k] @ x[k - 1]
            if k + 1 < n: r = r - U[k] @ x[k + 1]
            x[k] = inv[k] @ r
        return x
    return np.concatenate(rec(D, L, U, f))


def main():
    rng = np.random.default_rng(1)
    print("  N   cond(A)   |ref-exact|/|exact|  |bcr-exact|/|exact|  |bcr-ref|/|ref|   (worst of 20 systems, 9 right-hand sides each)")
    for N in (5, 9, 16, 32, 64):
        worst = np.zeros(3); cond = 0.0
        for trial in range(20):
            T = np.exp(rng.uniform(np.log(0.4), np.log(4.0), N))      # piece durations as the solver visits them (0.4 .. 4 s)
            A = minco_matrix(T)
            cond = max(cond, np.linalg.cond(A))
            B = np.zeros((6 * N, 9))
            B[0:3] = rng.normal(0, [[3.0], [0.5], [0.2]], (3, 9)); B[-3:] = rng.normal(0, [[3.0], [0.5], [0.2]], (3, 9))
            for i in range(N - 1): B[6 * i + 5] = rng.normal(0, 3.0, 9)
            Al = A.astype(np.longdouble)
            for d in range(9):
                exact = np.linalg.solve(A, B[:, d])        # float64 pivoted start, refined twice in extended precision
                xl = exact.astype(np.longdouble)
                for _ in range(3):
                    r = B[:, d].astype(np.longdouble) - Al @ xl
                    xl = xl + np.linalg.solve(A, r.astype(np.float64)).astype(np.longdouble)
                xr = solve_ref(A, B[:, d]); xb = solve_bcr(A, B[:, d])
                ne = float(np.abs(xl).max())
                worst = np.maximum(worst, [float(np.abs(xr - xl).max()) / ne, float(np.abs(xb - xl).max()) / ne,
                                           float(np.abs(xb - xr).max()) / float(np.abs(xr).max())])
        print("%3d  %8.1e  %18.1e  %18.1e  %16.1e" % (N, cond, worst[0], worst[1], worst[2]))


if __name__ == "__main__":
    main()
