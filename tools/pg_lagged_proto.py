#!/usr/bin/env python3
"""CPU prototype (numpy / scipy, no GPU) of the one preconditioner family the earlier rounds had not counted for the pose
graph's damped solve: a LAGGED coarse correction.  The persistent PCG kernel pays two dependent grid exchanges per
iteration -- (1) p and the scalars, (2) the coarse level's gather: restrict r, solve on the coarse matrix, prolong.  If the
coarse half of M^-1 is applied to the PREVIOUS iteration's restricted residual, its gather travels with the first exchange and
nothing waits for it: one exchange per iteration.  The preconditioner then changes from iteration to iteration, so the
recurrence must be the flexible one (Polak-Ribiere beta).  What it costs in iterations, on the bench graph (5 000 keyframes /
24 999 edges) at its optimum with lambda -> 0, relative residual 1e-8:

    python tools/pg_lagged_proto.py [agg_size]
"""
import importlib
import os
import sys
import time

import numpy as np
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import posegraph_oracle as po  # noqa: E402
from pg_precond_proto import bfs_aggregates, rigid_P, block_inverse_op, scalar_idx, pcg  # noqa: E402

synth = importlib.import_module("the-cooper-mapper_amd.synth")


def fpcg(A, b, local, coarse, lag, tol=1e-8, maxit=5000, restart_exact_first=True):
    """Flexible PCG (Polak-Ribiere).  M_k r_k = local(r_k) + coarse(r_{k-lag}) (lag = 0: the ordinary additive preconditioner).
    The first iteration has no older residual: it applies the coarse level to r_0 itself."""
    x = np.zeros_like(b)
    r = b.copy()
    hist = [r.copy()]
    z = local(r) + coarse(r)
    p = z.copy()
    rz = r @ z
    bb = np.sqrt(b @ b)
    for k in range(1, maxit + 1):
        q = A @ p
        al = rz / (p @ q)
        x += al * p
        r_new = r - al * q
        if np.sqrt(r_new @ r_new) <= tol * bb:
            return x, k
        hist.append(r_new.copy())
        r_c = hist[-1 - lag] if len(hist) > lag else hist[0]
        z_new = local(r_new) + coarse(r_c)
        beta = (z_new @ (r_new - r)) / rz  # Polak-Ribiere: tolerates a preconditioner that changes with k
        rz = r_new @ z_new
        p = z_new + beta * p
        r = r_new
        if len(hist) > lag + 2:
            hist.pop(0)
    return x, maxit


def main():
    cap = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    g = synth.make_pose_graph()
    opt = np.load(os.path.join(ROOT, "tests", "golden", "posegraph_bench_optimum.npz"))["poses"]
    H, b, c2 = po.linearize(opt, g["ij"], g["meas"], g["info"])
    H = H.tolil()
    H[:6, :] = 0
    H[:, :6] = 0
    H[:6, :6] = np.eye(6)
    H = H.tocsr()
    rng = np.random.default_rng(1)
    b = H @ rng.normal(size=H.shape[0])  # the gradient vanishes at the optimum: a generic right-hand side of the same smoothness
    b[:6] = 0
    n_v = len(opt)
    lam = 1e-9 * H.diagonal().max()
    A = (H + lam * sp.identity(H.shape[0])).tocsr()
    agg, members = bfs_aggregates(n_v, g["ij"], cap)
    P = rigid_P(opt, members).tolil()
    P[:6, :] = 0
    P = P.tocsr()
    Ac = (P.T @ A @ P).toarray()
    Aci = np.linalg.inv(Ac + 1e-12 * np.trace(Ac) / len(Ac) * np.eye(len(Ac)))
    coarse = lambda r: P @ (Aci @ (P.T @ r))  # noqa: E731
    bj = block_inverse_op(A, [np.arange(6 * v, 6 * v + 6) for v in range(n_v)])
    aggop = block_inverse_op(A, [scalar_idx(m) for m in members])
    print("bench graph at its optimum, lambda = 1e-9 x max diag, %d aggregates (cap %d), relative residual 1e-8" % (len(members), cap))
    for name, local in (("block Jacobi", bj), ("aggregate inverses", aggop)):
        t = time.time()
        it0 = pcg(A, b, lambda r: local(r) + coarse(r))[1]
        print("  %-20s + coarse, PCG                       %5d iterations  (two exchanges each)   (%.0f s)" % (name, it0, time.time() - t), flush=True)
        for lag in (0, 1, 2):
            t = time.time()
            it = fpcg(A, b, local, coarse, lag)[1]
            ex = "two exchanges each" if lag == 0 else "ONE exchange each"
            print("  %-20s + coarse lagged by %d, flexible PCG   %5d iterations  (%s)   (%.0f s)" % (name, lag, it, ex, time.time() - t), flush=True)


if __name__ == "__main__":
    main()
