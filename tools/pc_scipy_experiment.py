#!/usr/bin/env python
"""Preconditioner experiments on a dumped system (tools/dump_system.py): block-Jacobi ILU(0) through
the oracle library, optional piecewise-constant coarse space through scipy.  Experiments only."""
import argparse, os, sys, time
import ctypes as C
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import binding as ol

ap = argparse.ArgumentParser()
ap.add_argument("--sys", default="/tmp/exp/system.npz")
ap.add_argument("--agg", type=int, nargs=3, default=None, help="aggregate shape in cells (default: no coarse space)")
ap.add_argument("--mode", default="mult", choices=["add", "mult", "mult2"])
ap.add_argument("--comps", type=int, nargs="*", default=None, help="components in the coarse space (default all)")
ap.add_argument("--levels", type=int, default=1, help="1: exact coarse solve; >1: recursive aggregation x agg with ILU smoothing")
ap.add_argument("--rtol", type=float, default=1e-5)
a = ap.parse_args()
d = np.load(a.sys)
rp, ci, val, f, sub_ptr, ijk, bs = d["rowptr"], d["colidx"], d["val"], d["f"], d["sub_ptr"], d["ijk"], int(d["bs"])
n = rp.size - 1
A = sp.bsr_matrix((val.reshape(-1, bs, bs), ci, rp), shape=(n * bs, n * bs)).tocsr()
L = ol.load(os.path.join(ROOT, "oracle", "liboracle.so"))
fval = np.zeros_like(val); dinv = np.zeros(n * bs * bs)
sp32 = sub_ptr.astype(np.int32)
assert L.wo_bilu0_factor(n, bs, ol.ip(rp), ol.ip(ci), ol.dp(val), sp32.size - 1, ol.ip(sp32), ol.dp(fval), ol.dp(dinv)) == 0

def ilu(r):
    z = np.zeros(n * bs)
    r = np.ascontiguousarray(r)
    L.wo_bilu0_apply(n, bs, ol.ip(rp), ol.ip(ci), ol.dp(fval), ol.dp(dinv), sp32.size - 1, ol.ip(sp32), ol.dp(r), ol.dp(z))
    return z

M = ilu
if a.agg:
    ax, ay, az = a.agg
    dims = d["dims"]
    gx, gy, gz = -(-dims[0] // ax), -(-dims[1] // ay), -(-dims[2] // az)
    aggid = ((ijk[:, 2] // az) * gy + ijk[:, 1] // ay) * gx + ijk[:, 0] // ax
    nc = gx * gy * gz
    comps = a.comps if a.comps is not None else list(range(bs))
    rows, cols = [], []
    for k, c in enumerate(comps):
        rows.append(np.arange(n) * bs + c); cols.append(aggid * len(comps) + k)
    P = sp.csr_matrix((np.ones(n * len(comps)), (np.concatenate(rows), np.concatenate(cols))), shape=(n * bs, nc * len(comps)))
    Ac = (P.T @ A @ P).tocsc()
    t0 = time.time()
    lu = spla.splu(Ac)
    print("coarse: %d aggregates, %d unknowns, nnz %d, LU %.1f s" % (nc, Ac.shape[0], Ac.nnz, time.time() - t0))
    def coarse(r):
        return P @ lu.solve(P.T @ r)
    if a.mode == "add":
        M = lambda r: ilu(r) + coarse(r)
    elif a.mode == "mult":      # coarse correction, then ILU on the updated residual
        def M(r):
            z = coarse(r)
            return z + ilu(r - A @ z)
    else:                        # ILU, then coarse correction of the remaining residual
        def M(r):
            z = ilu(r)
            return z + coarse(r - A @ z)

def bcgs(A, b, M, rtol, maxit=5000):
    x = np.zeros_like(b)
    R = M(b); dp0 = np.linalg.norm(R); RP = R.copy()
    P = np.zeros_like(b); V = np.zeros_like(b)
    rho_old = alpha = omega = 1.0
    for i in range(maxit):
        rho = R @ RP
        beta = (rho / rho_old) * (alpha / omega)
        P = R + beta * (P - omega * V)
        V = M(A @ P)
        alpha = rho / (V @ RP)
        S = R - alpha * V
        T = M(A @ S)
        omega = (S @ T) / (T @ T)
        x += alpha * P + omega * S
        R = S - omega * T
        dp = np.linalg.norm(R)
        rho_old = rho
        if dp <= rtol * dp0:
            return x, i + 1, dp / dp0
    return x, maxit, dp / dp0

t0 = time.time()
x, its, rel = bcgs(A, f, M, a.rtol)
true = np.linalg.norm(f - A @ x) / np.linalg.norm(f)
print("agg %s mode %s comps %s: %d its, prec rel %.2e, true rel %.2e, %.1f s" % (a.agg, a.mode, a.comps, its, rel, true, time.time() - t0))
