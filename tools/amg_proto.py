"""CPU prototype (SciPy) of multigrid variants for the inner operator B = c L + W_H of the
contraction solve (pyqsm_amd/csrc/amg.hip). Not part of the product: it answers "how many
preconditioned CG iterations for two digits would variant X need" before anything is written
in HIP. Systems are captured from the oracle loop (oracle.extract_skeleton) on a small forest.

    python tools/amg_proto.py [n_points] [steps]
"""
import sys
import time

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as sla

sys.path.insert(0, ".")
import oracle  # noqa: E402
from pyqsm_amd import synth  # noqa: E402

THETA = 0.08
COARSE_MAX = 96
TAIL_SWEEPS = 8


def capture(n, steps, c0=3.0):
    P = synth.forest(n, seed=4)
    out = []

    def solve(cur, L, wl, wh):
        out.append((L.copy().tocsr(), wl.copy(), wh.copy(), cur.copy()))
        # cheap stand-in for spsolve: normal equations by CG to 1e-10 with a Jacobi preconditioner
        c = wl[0]
        A = (c * c) * (L.T @ L) + sp.diags(wh * wh)
        b = (wh * wh)[:, None] * cur
        lu = sla.splu(A.tocsc())
        return lu.solve(b)

    lo, hi = P.min(0) - 1.0, P.max(0) + 1.0
    oracle.extract_skeleton(P, lambda p: oracle.point_cloud_laplacian(p, 20, 1e-6), (lo, hi), max_iter=steps,
                            termination_ratio=0.0, contraction_factor=c0, attraction_factor=3, solve=solve)
    return out


def aggregate(A, rng):
    """MIS-2 roots of the strength graph; the others join the root they reach in 1 or 2 strong steps."""
    n = A.shape[0]
    d = A.diagonal()
    C = A.tocoo()
    off = C.row != C.col
    strong = off & (np.abs(C.data) >= THETA * np.sqrt(d[C.row] * d[C.col]))
    S = sp.csr_matrix((np.abs(C.data[strong]), (C.row[strong], C.col[strong])), shape=(n, n))
    has = np.diff(S.indptr) > 0
    Sb = S.copy()
    Sb.data[:] = 1.0
    S2 = (Sb @ Sb + Sb).tolil()
    S2.setdiag(0)
    S2 = S2.tocsr()
    S2.eliminate_zeros()
    S2.data[:] = 1.0
    pri = rng.permutation(n).astype(np.float64) + 1.0
    state = np.zeros(n, dtype=np.int8)  # 0 undecided 1 root 2 out
    state[~has] = 2
    while (state == 0).any():
        p = np.where(state == 0, pri, 0.0)
        nb = S2.multiply(p[None, :]).tocsr()
        mx = np.asarray(nb.max(axis=1).todense()).ravel() if nb.nnz else np.zeros(n)
        new_root = (state == 0) & (p > mx)
        state[new_root] = 1
        covered = (S2 @ new_root.astype(np.float64)) > 0
        state[(state == 0) & covered] = 2
    roots = np.flatnonzero(state == 1)
    agg = np.full(n, -1, dtype=np.int64)
    agg[roots] = np.arange(len(roots))
    for _ in range(2):  # one strong step at a time, strongest link to an assigned point wins
        un = np.flatnonzero((agg < 0) & has)
        if len(un) == 0:
            break
        sub = S[un].tocoo()
        ok = agg[sub.col] >= 0
        if not ok.any():
            break
        r, cc, v = sub.row[ok], sub.col[ok], sub.data[ok]
        order = np.lexsort((-v, r))
        r, cc = r[order], cc[order]
        first = np.r_[True, r[1:] != r[:-1]]
        agg[un[r[first]]] = agg[cc[first]]
    return agg, len(roots)


class Level:
    pass


def build(B, smooth_p=0.0, rng=None, filt=False):
    rng = rng or np.random.default_rng(0)
    lv = []
    A = B.tocsr()
    while True:
        L = Level()
        L.A = A
        L.n = A.shape[0]
        L.diag = A.diagonal()
        L.l1 = np.asarray(abs(A).sum(axis=1)).ravel()
        lv.append(L)
        if L.n <= COARSE_MAX or len(lv) >= 24:
            break
        agg, nc = aggregate(A, rng)
        if nc == 0 or nc >= 0.9 * L.n:
            break
        keep = agg >= 0
        Pt = sp.csr_matrix((np.ones(keep.sum()), (np.flatnonzero(keep), agg[keep])), shape=(L.n, nc))
        if smooth_p > 0.0:
            Af = A
            if filt:  # drop weak entries into the diagonal
                C = A.tocoo()
                d = L.diag
                weak = (C.row != C.col) & (np.abs(C.data) < THETA * np.sqrt(d[C.row] * d[C.col]))
                Af = sp.csr_matrix((np.where(weak, 0.0, C.data), (C.row, C.col)), shape=A.shape)
                Af = Af + sp.diags(np.asarray(sp.csr_matrix((np.where(weak, C.data, 0.0), (C.row, C.col)),
                                                            shape=A.shape).sum(axis=1)).ravel())
            Dinv = sp.diags(1.0 / Af.diagonal())
            Pm = (Pt - smooth_p * (Dinv @ (Af @ Pt))).tocsr()
        else:
            Pm = Pt
        L.P = Pm
        A = (Pm.T @ A @ Pm).tocsr()
        A.eliminate_zeros()
    last = lv[-1]
    last.inv = np.linalg.inv(last.A.toarray()) if last.n <= COARSE_MAX else None
    return lv


def smooth(L, x, b, kind, sweeps=1):
    if kind == "l1":
        for _ in range(sweeps):
            x = x + (b - L.A @ x) / L.l1[:, None]
    elif kind.startswith("jac"):
        w = float(kind[3:])
        for _ in range(sweeps):
            x = x + w * (b - L.A @ x) / L.diag[:, None]
    elif kind.startswith("cheb"):  # Chebyshev of the given degree on D^-1 A, lambda_max estimate 2 (M-matrix bound)
        deg = int(kind[4:])
        lmax = 2.0
        lmin = lmax / 4.0  # smoothing interval [lmax/4, lmax]
        theta, delta = 0.5 * (lmax + lmin), 0.5 * (lmax - lmin)
        sigma = theta / delta
        rho = 1.0 / sigma
        r = (b - L.A @ x) / L.diag[:, None]
        dvec = r / theta
        for k in range(deg):
            x = x + dvec
            if k == deg - 1:
                break
            r = (b - L.A @ x) / L.diag[:, None]
            rho_new = 1.0 / (2.0 * sigma - rho)
            dvec = rho_new * rho * dvec + 2.0 * rho_new / delta * r
            rho = rho_new
    return x


def cycle(lv, k, b, kind="l1", sweeps=1, kcycle=0, scale=False):
    L = lv[k]
    if k == len(lv) - 1:
        if L.inv is not None:
            return L.inv @ b
        return smooth(L, np.zeros_like(b), b, kind, TAIL_SWEEPS)
    x = smooth(L, np.zeros_like(b), b, kind, sweeps)
    r = b - L.A @ x
    rc = L.P.T @ r
    if kcycle and k + 1 < len(lv) - 1 and k < kcycle:
        # two steps of flexible CG on the coarse system, preconditioned by the cycle below
        Ac = lv[k + 1].A
        xc = np.zeros_like(rc)
        rr = rc.copy()
        pprev = None
        for it in range(2):
            z = cycle(lv, k + 1, rr, kind, sweeps, kcycle, scale)
            if pprev is not None:
                beta = -np.sum(z * qprev, axis=0) / pq
                p = z + beta * pprev
            else:
                p = z
            q = Ac @ p
            pq = np.sum(p * q, axis=0)
            alpha = np.sum(p * rr, axis=0) / np.where(pq > 0, pq, 1.0)
            xc = xc + alpha * p
            rr = rr - alpha * q
            pprev, qprev = p, q
    else:
        xc = cycle(lv, k + 1, rc, kind, sweeps, kcycle, scale)
    e = L.P @ xc
    if scale:  # energy-minimising step length of the coarse correction
        Ae = L.A @ e
        den = np.sum(e * Ae, axis=0)
        e = e * (np.sum(e * r, axis=0) / np.where(den > 0, den, 1.0))
    x = x + e
    # post-smoothing mirrors the pre-smoothing
    x = smooth(L, x, b, kind, sweeps)
    return x


def pcg(B, b, M, rtol=1e-2, max_it=300, flexible=False):
    x = np.zeros_like(b)
    r = b.copy()
    z = M(r)
    p = z.copy()
    rz = np.sum(r * z, axis=0)
    b2 = np.sum(b * b, axis=0)
    for it in range(1, max_it + 1):
        q = B @ p
        alpha = rz / np.sum(p * q, axis=0)
        x += alpha * p
        r_new = r - alpha * q
        if np.sqrt((np.sum(r_new * r_new, axis=0) / b2).max()) <= rtol:
            return it
        z_new = M(r_new)
        if flexible:
            beta = np.sum(z_new * (r_new - r), axis=0) / rz
        else:
            beta = np.sum(r_new * z_new, axis=0) / rz
        rz = np.sum(r_new * z_new, axis=0)
        r, z = r_new, z_new
        p = z + beta * p
    return max_it


def work(lv, passes):
    n0 = lv[0].A.nnz
    return sum(L.A.nnz for L in lv[:-1]) / n0 * passes


VARIANTS = [
    ("plain V(1,1) l1   [now]", dict(), dict(kind="l1"), 3),
    ("plain V(1,1) jac0.7", dict(), dict(kind="jac0.7"), 3),
    ("plain V(1,1) jac0.8", dict(), dict(kind="jac0.8"), 3),
    ("plain V(2,2) l1", dict(), dict(kind="l1", sweeps=2), 5),
    ("plain V(1,1) cheb2", dict(), dict(kind="cheb2"), 5),
    ("plain V(1,1) cheb3", dict(), dict(kind="cheb3"), 7),
    ("plain V l1 scaled corr", dict(), dict(kind="l1", scale=True), 4),
    ("plain K(1 lvl) l1", dict(), dict(kind="l1", kcycle=1), 3),
    ("plain K(2 lvl) l1", dict(), dict(kind="l1", kcycle=2), 3),
    ("plain K(all) l1", dict(), dict(kind="l1", kcycle=99), 3),
    ("SA w=0.67 V(1,1) l1", dict(smooth_p=0.67), dict(kind="l1"), 3),
    ("SA w=0.67 filt V(1,1) l1", dict(smooth_p=0.67, filt=True), dict(kind="l1"), 3),
    ("SA w=0.67 V(1,1) jac0.7", dict(smooth_p=0.67), dict(kind="jac0.7"), 3),
    ("SA w=0.67 V(1,1) cheb2", dict(smooth_p=0.67), dict(kind="cheb2"), 5),
]


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    only = sys.argv[3].split(",") if len(sys.argv) > 3 else None
    t = time.time()
    systems = capture(n, steps)
    print(f"captured {len(systems)} systems of {n} points in {time.time() - t:.1f} s", flush=True)
    rng = np.random.default_rng(1)
    for si in sorted({0, 2, min(5, len(systems) - 1), len(systems) - 1}):
        L, wl, wh, pts = systems[si]
        c = wl[0]
        B = (c * L + sp.diags(wh)).tocsr()
        # right-hand sides as the outer iteration sees them: a residual of the normal equations and B^-1 of it
        A = (c * c) * (L.T @ L) + sp.diags(wh * wh)
        r0 = (wh * wh)[:, None] * pts - A @ pts
        y = sla.splu(B.tocsc()).solve(r0)
        print(f"== step {si}: c = {c:.3g}, W_H {wh.min():.3g}..{wh.max():.3g}, nnz/row {B.nnz / B.shape[0]:.1f}", flush=True)
        cache = {}
        for name, bkw, ckw, passes in VARIANTS:
            if only and not any(o in name for o in only):
                continue
            key = tuple(sorted(bkw.items()))
            if key not in cache:
                cache[key] = build(B, rng=np.random.default_rng(0), **bkw)
            lv = cache[key]
            M = lambda r: cycle(lv, 0, r, **ckw)  # noqa: E731
            flex = "K(" in name or "scaled" in name
            it1 = pcg(B, r0, M, flexible=flex)
            it2 = pcg(B, y, M, flexible=flex)
            sizes = [L_.n for L_ in lv]
            cx = sum(L_.A.nnz for L_ in lv) / lv[0].A.nnz
            print(f"  {name:28s} its {it1:3d} + {it2:3d}   levels {sizes}  op-complexity {cx:.2f}", flush=True)


if __name__ == "__main__":
    main()
