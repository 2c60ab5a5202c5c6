#!/usr/bin/env python3
"""CPU prototype (numpy / scipy): does REUSE BETWEEN DAMPED SOLVES pay for the bench pose graph (round-4 review, item 5)?

Near the optimum Levenberg-Marquardt takes ~60 iterations in which the system barely changes (lambda -> 0, the poses move by
1e-5), and every damped solve starts from zero with the same two-level preconditioner (6x6 block Jacobi + rigid-body coarse
level on breadth-first aggregates: 83 PCG iterations to 1e-8 at the optimum -- tools/pg_precond_proto.py).  Tried here, on
consecutive systems A_0 x = b_0, A_1 x = b_1, ... made from the committed optimum perturbed the way late LM iterates differ:

  plain        every solve from zero (the device solver today)
  warm         x0 = the previous solve's solution (scaled by the ratio of the right-hand sides' norms)
  init(k)      x0 = W (W^T A W)^-1 W^T b with W = the k lowest Ritz vectors harvested from the FIRST solve's Lanczos basis
               (the normalised preconditioned residuals of PCG); the solve itself unchanged ("init-CG", Erhel & Guyomarc'h)
  defl(k)      the same W appended to the coarse space of the preconditioner (additive), refreshed never

    python tools/pg_recycle_proto.py
"""
import importlib
import os
import sys

import numpy as np
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import posegraph_oracle as po  # noqa: E402
from pg_precond_proto import bfs_aggregates, rigid_P, block_inverse_op  # noqa: E402

synth = importlib.import_module("the-cooper-mapper_amd.synth")


def pcg(A, b, M, x0=None, tol=1e-8, maxit=5000, keep=False, bnorm=None):
    """-> x, iterations, V (the M-orthonormal Lanczos vectors z_j / sqrt(r_j . z_j) when keep)."""
    x = np.zeros_like(b) if x0 is None else x0.copy()
    r = b - A @ x if x0 is not None else b.copy()
    z = M(r)
    p = z.copy()
    rz = r @ z
    bb = np.sqrt(b @ b) if bnorm is None else bnorm
    V = []
    for k in range(1, maxit + 1):
        if keep:
            V.append(z / np.sqrt(rz))
        q = A @ p
        al = rz / (p @ q)
        x += al * p
        r -= al * q
        if np.sqrt(r @ r) <= tol * bb:
            return x, k, (np.array(V).T if keep else None)
        z = M(r)
        rz2 = r @ z
        p = z + (rz2 / rz) * p
        rz = rz2
    return x, maxit, (np.array(V).T if keep else None)


def system(g, poses, lam_rel):
    H, b, c2 = po.linearize(poses, g["ij"], g["meas"], g["info"])
    H = H.tolil()
    H[:6, :] = 0
    H[:, :6] = 0
    H[:6, :6] = np.eye(6)
    H = H.tocsr()
    b = b.copy()
    b[:6] = 0
    lam = lam_rel * H.diagonal().max()
    return (H + lam * sp.identity(H.shape[0])).tocsr(), b, c2


def precond(A, poses, members):
    n_v = len(poses)
    bj = block_inverse_op(A, [np.arange(6 * v, 6 * v + 6) for v in range(n_v)])
    P = rigid_P(poses, members).tolil()
    P[:6, :] = 0
    P = P.tocsr()
    Ac = (P.T @ A @ P).toarray()
    Aci = np.linalg.inv(Ac + 1e-12 * np.trace(Ac) / len(Ac) * np.eye(len(Ac)))
    return lambda r: bj(r) + P @ (Aci @ (P.T @ r))


def main():
    g = synth.make_pose_graph()
    opt = np.load(os.path.join(ROOT, "tests", "golden", "posegraph_bench_optimum.npz"))["poses"]
    n_v = len(opt)
    agg, members = bfs_aggregates(n_v, g["ij"], 64)
    rng = np.random.default_rng(3)

    def perturbed(eps):  # a late LM iterate: the optimum moved by a smooth + rough local update of size eps
        d = rng.normal(size=(n_v, 6)) * eps
        d[0] = 0
        return po.oplus(opt, d.ravel(), 0)

    for lam_rel in (1e-7, 1e-9):
        print("== lambda = %.0e x max diag" % lam_rel, flush=True)
        systems = []
        for j, eps in enumerate((1e-4, 3e-5, 1e-5, 3e-6, 1e-6)):
            poses = perturbed(eps)
            A, b, c2 = system(g, poses, lam_rel * (0.6 ** j))  # lambda keeps shrinking as LM accepts steps
            systems.append((A, b, poses))
        A0, b0, p0 = systems[0]
        M0 = precond(A0, p0, members)
        x_prev, it0, V = pcg(A0, b0, M0, keep=True)
        print("  first solve: %d iterations; Lanczos basis %s" % (it0, V.shape), flush=True)
        T = V.T @ (A0 @ V)
        th, Y = np.linalg.eigh(0.5 * (T + T.T))
        print("  lowest Ritz values of M^-1 A: %s ... largest %.3g" % (np.array2string(th[:6], precision=4), th[-1]))
        for (A, b, poses) in systems[1:]:
            M = precond(A, poses, members)  # the device rebuilds P from the current poses, keeps the inverse while lambda moves < 10x
            row = []
            _, it_plain, _ = pcg(A, b, M)
            row.append("plain %3d" % it_plain)
            s = np.sqrt(b @ b) / max(1e-300, np.sqrt(b0 @ b0))
            _, it_warm, _ = pcg(A, b, M, x0=x_prev * s)
            row.append("warm %3d" % it_warm)
            for k in (8, 16, 32):
                W = V @ Y[:, :k]
                G = W.T @ (A @ W)
                x0 = W @ np.linalg.solve(G, W.T @ b)
                _, it_i, _ = pcg(A, b, M, x0=x0)
                Gi = np.linalg.inv(G)
                _, it_d, _ = pcg(A, b, lambda r, W=W, Gi=Gi: M(r) + W @ (Gi @ (W.T @ r)))
                # both: deflated preconditioner AND the projected start
                _, it_b, _ = pcg(A, b, lambda r, W=W, Gi=Gi: M(r) + W @ (Gi @ (W.T @ r)), x0=x0)
                row.append("k=%2d: init %3d defl %3d both %3d" % (k, it_i, it_d, it_b))
            print("  next system: " + " | ".join(row), flush=True)
            x_prev, b0 = pcg(A, b, M)[0], b


if __name__ == "__main__":
    main()
