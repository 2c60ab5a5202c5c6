#!/usr/bin/env python3
"""Second prototype round (see pg_precond_proto.py): the coarse space is the weak part -- variants of it with 6x6 block Jacobi."""
import importlib, os, sys, time
import numpy as np, scipy.sparse as sp
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import posegraph_oracle as po
from pg_precond_proto import bfs_aggregates, rigid_P, pcg, block_inverse_op, scalar_idx
synth = importlib.import_module("the-cooper-mapper_amd.synth")

g = synth.make_pose_graph()
opt = np.load(os.path.join(ROOT, "tests", "golden", "posegraph_bench_optimum.npz"))["poses"]
poses = opt
H, b, c2 = po.linearize(poses, g["ij"], g["meas"], g["info"])
H = H.tolil(); H[:6, :] = 0; H[:, :6] = 0; H[:6, :6] = np.eye(6); H = H.tocsr()
rng = np.random.default_rng(1)
b = H @ rng.normal(size=H.shape[0]); b[:6] = 0
dmax = H.diagonal().max()
n_v = len(poses)
lam = float(os.environ.get("LAM", "1e-9")) * dmax
A = (H + lam * sp.identity(H.shape[0])).tocsr()
bj = block_inverse_op(A, [np.arange(6 * v, 6 * v + 6) for v in range(n_v)])
# D^-1 as a sparse block-diagonal matrix
blocks = [np.linalg.inv(A[6*v:6*v+6, 6*v:6*v+6].toarray()) for v in range(n_v)]
Dinv = sp.block_diag(blocks, format="csr")

def coarse_op(P):
    Ac = (P.T @ A @ P).toarray()
    Aci = np.linalg.inv(Ac + 1e-13 * np.trace(Ac) / len(Ac) * np.eye(len(Ac)))
    return (lambda r: P @ (Aci @ (P.T @ r))), Ac.shape[0]

def fixP(P):
    P = P.tolil(); P[:6, :] = 0; return P.tocsr()

for cap in [int(a) for a in sys.argv[1:]] or [64]:
    agg, members = bfs_aggregates(n_v, g["ij"], cap)
    P0 = fixP(rigid_P(poses, members))
    c0, nc = coarse_op(P0)
    t = time.time(); it = pcg(A, b, lambda r: bj(r) + c0(r))[1]
    print("cap %3d  %3d aggs  coarse %4d : bj+c %4d its" % (cap, len(members), nc, it), flush=True)
    # smoothed aggregation: P = (I - w D^-1 A) P0
    for w in (0.5, 0.66):
        P1 = fixP((sp.identity(A.shape[0]) - w * (Dinv @ A)) @ P0)
        c1, _ = coarse_op(P1)
        it = pcg(A, b, lambda r: bj(r) + c1(r))[1]
        print("          smoothed aggregation w=%.2f (nnz P %d vs %d): bj+c %4d its" % (w, P1.nnz, P0.nnz, it), flush=True)
    # enriched: rigid modes x (1, s) with s = normalised member index inside the aggregate
    cols = []
    for a, m in enumerate(members):
        s = np.zeros(6 * n_v)
        for k, v in enumerate(m):
            s[6 * v:6 * v + 6] = (k / max(1, len(m) - 1)) - 0.5
        cols.append(s)
    S = np.stack(cols, 1)  # n x n_agg weights
    Pe = sp.hstack([P0, sp.csr_matrix(P0.multiply(np.repeat(S, 6, axis=1)))]).tocsr()
    ce, nce = coarse_op(Pe)
    it = pcg(A, b, lambda r: bj(r) + ce(r))[1]
    print("          enriched (rigid x {1, member index}) coarse %d: bj+c %4d its" % (nce, it), flush=True)
    # deflated PCG (A-DEF2-like): z = bj(r) ; z += c0(r - A z)
    def def2(r):
        z = bj(r)
        return z + c0(r - A @ z)
    it = pcg(A, b, def2)[1]
    print("          bj then coarse on the new residual (non-symmetric, A-DEF2): %4d its" % it, flush=True)
    # two sweeps of block Jacobi (damped) + c
    def bj2(r, w=0.7):
        z = w * bj(r)
        z = z + w * bj(r - A @ z)
        return z
    it = pcg(A, b, lambda r: bj2(r) + c0(r))[1]
    print("          two damped BJ sweeps + c: %4d its" % it, flush=True)
