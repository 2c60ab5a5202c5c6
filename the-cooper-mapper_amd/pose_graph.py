"""Host-side mirror of the reference's pose-graph solver interface over the C ABI.

``PoseGraph`` keeps the method names and argument meaning of ``pose_graph::SolverG2O``
(/root/reference/L_SLAM/src/pose_graph/solver_g2o.h:46-95, solver_g2o.cpp:51-95):
``add_se3_node(pose)`` (the first node is fixed), ``add_se3_edge(v1, v2, relative_pose,
information)``, ``optimize()``.  Poses are 4x4 homogeneous matrices (Eigen::Isometry3d) or
7-vectors {t, q_xyzw}.  The arithmetic runs in the HIP kernels of csrc/lslam_posegraph.hip.
"""
import ctypes as C

import numpy as np

from .capi import ALLGATHERV_FN, ALLREDUCE_FN, LslamError, LslamPgStats, c_double_p, c_int32_p, load_library


def _dp(a):
    return a.ctypes.data_as(c_double_p)


def mat_to_pose7(T):
    """4x4 -> {t, q_xyzw} (Eigen Quaterniond(R), Shepperd's method)."""
    T = np.asarray(T, np.float64)
    R, t = T[:3, :3], T[:3, 3]
    tr = np.trace(R)
    if tr > 0:
        s = np.sqrt(tr + 1.0)
        w = 0.5 * s
        s = 0.5 / s
        q = [(R[2, 1] - R[1, 2]) * s, (R[0, 2] - R[2, 0]) * s, (R[1, 0] - R[0, 1]) * s, w]
    else:
        i = 0
        if R[1, 1] > R[0, 0]:
            i = 1
        if R[2, 2] > R[i, i]:
            i = 2
        j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0)
        q = [0.0, 0.0, 0.0, 0.0]
        q[i] = 0.5 * s
        s = 0.5 / s
        q[3] = (R[k, j] - R[j, k]) * s
        q[j] = (R[j, i] + R[i, j]) * s
        q[k] = (R[k, i] + R[i, k]) * s
    return np.array([t[0], t[1], t[2], *q])


def pose7_to_mat(p):
    x, y, z, w = p[3:]
    T = np.eye(4)
    T[:3, :3] = [[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                 [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                 [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]]
    T[:3, 3] = p[:3]
    return T


class PoseGraph:
    def __init__(self, device=0):
        self.lib = load_library()
        self.device = device
        self._nodes = []
        self._ij = []
        self._meas = []
        self._info = []
        self.h = None
        self._cb = None
        self.last_stats = None

    # ---- SolverG2O interface ------------------------------------------------------
    def add_se3_node(self, pose):
        """solver_g2o.cpp:51-63: returns the vertex id; the first vertex is fixed."""
        p = np.asarray(pose, np.float64)
        self._drop()  # first: it pulls the optimised estimates of the existing vertices back
        self._nodes.append(mat_to_pose7(p) if p.shape == (4, 4) else p.reshape(7).copy())
        return len(self._nodes) - 1

    def add_se3_edge(self, v1, v2, relative_pose, information_matrix):
        """solver_g2o.cpp:65-77."""
        z = np.asarray(relative_pose, np.float64)
        self._drop()
        self._ij.append((int(v1), int(v2)))
        self._meas.append(mat_to_pose7(z) if z.shape == (4, 4) else z.reshape(7).copy())
        self._info.append(np.asarray(information_matrix, np.float64).reshape(6, 6).copy())
        return len(self._ij) - 1

    def optimize(self, max_iterations=1000):
        """solver_g2o.cpp:79-95: graph->optimize(1000).  Returns the iteration count."""
        self._build()
        st = LslamPgStats()
        self._check(self.lib.lslam_pg_optimize(self.h, int(max_iterations), C.byref(st)))
        self.last_stats = st
        return st.iterations

    # ---- bulk construction / access ---------------------------------------------------
    def set_graph(self, poses7, ij, meas7, info):
        self._drop(keep_estimates=False)
        self._nodes = [p for p in np.asarray(poses7, np.float64).reshape(-1, 7)]
        self._ij = [tuple(int(v) for v in e) for e in np.asarray(ij).reshape(-1, 2)]
        self._meas = [m for m in np.asarray(meas7, np.float64).reshape(-1, 7)]
        self._info = [w for w in np.asarray(info, np.float64).reshape(-1, 6, 6)]
        self._drop()

    # ---- g2o text format (solver_g2o.cpp:97-100) -----------------------------------------
    def save(self, path):
        """SolverG2O::save: the graph with its current estimates as a .g2o file."""
        self._build()
        self._check(self.lib.lslam_pg_save_g2o(self.h, str(path).encode()))

    @staticmethod
    def read_g2o(path, lib=None):
        """-> dict(poses (n,7), ij (m,2), meas (m,7), info (m,6,6), fixed) from a .g2o file (host only)."""
        from .capi import load_library
        lib = lib or load_library()
        nv, ne, fx = C.c_int32(0), C.c_int32(0), C.c_int32(-1)
        rc = lib.lslam_g2o_read(str(path).encode(), C.byref(nv), None, C.byref(ne), None, None, None, C.byref(fx))
        if rc < 0:
            raise LslamError(rc, lib.lslam_pg_last_error().decode())
        poses = np.zeros((nv.value, 7))
        ij = np.zeros((ne.value, 2), np.int32)
        meas = np.zeros((ne.value, 7))
        info = np.zeros((ne.value, 6, 6))
        rc = lib.lslam_g2o_read(str(path).encode(), C.byref(nv), _dp(poses), C.byref(ne),
                                ij.ctypes.data_as(c_int32_p), _dp(meas), _dp(info), C.byref(fx))
        if rc < 0:
            raise LslamError(rc, lib.lslam_pg_last_error().decode())
        return dict(poses=poses, ij=ij, meas=meas, info=info, fixed=fx.value)

    def load(self, path):
        """Replace the graph by the one in a .g2o file."""
        g = PoseGraph.read_g2o(path, self.lib)
        self.set_graph(g["poses"], g["ij"], g["meas"], g["info"])
        return g

    def poses(self):
        self._build()
        out = np.zeros((len(self._nodes), 7))
        self._check(self.lib.lslam_pg_get_poses(self.h, _dp(out)))
        return out

    def estimate(self, v):
        return pose7_to_mat(self.poses()[v])

    # ---- multi-GPU ------------------------------------------------------------------------
    def set_shard(self, e_begin, e_end, allreduce=None, system_tensor=None):
        """This rank linearises edges [e_begin, e_end); `allreduce(ptr, count)` must sum the
        `count` doubles at device address `ptr` over all ranks in place (e.g. a
        torch.distributed.all_reduce on `system_tensor`, whose storage the library then
        assembles into)."""
        self._build()
        if allreduce is None:
            self._cb = ALLREDUCE_FN(0)
        else:
            def _tramp(_user, ptr, count):
                allreduce(ptr, count)
            self._cb = ALLREDUCE_FN(_tramp)
        buf = C.c_void_p(system_tensor.data_ptr()) if system_tensor is not None else None
        self._check(self.lib.lslam_pg_set_shard(self.h, int(e_begin), int(e_end), self._cb, None, buf))
        self._sys_tensor = system_tensor

    def set_comm(self, comm, e_begin, e_end):
        """Edge shard [e_begin, e_end) of this rank, the block system all-reduced by the library's own
        RCCL communicator (lslam_pg_set_comm) on the solver's stream."""
        self._build()
        self._cb = ALLREDUCE_FN(0)
        self._check(self.lib.lslam_pg_set_shard(self.h, int(e_begin), int(e_end), self._cb, None, None))
        self._check(self.lib.lslam_pg_set_comm(self.h, comm.h if comm is not None else None))
        self._comm = comm

    def row_shard_range(self, rank, world):
        """lslam_pg_row_shard_range: this rank's vertex rows [v_begin, v_end) of an even partition in whole row blocks."""
        b, e = C.c_int32(), C.c_int32()
        self.lib.lslam_pg_row_shard_range(len(self._nodes), int(rank), int(world), C.byref(b), C.byref(e))
        return b.value, e.value

    def set_row_shard(self, v_begin, v_end):
        """lslam_pg_set_row_shard: the damped solves are shared by the ranks (row-sharded block-Jacobi PCG) instead of replicated;
        (-1, -1) switches back.  Needs set_shard / set_comm first (the transport)."""
        self._build()
        self._check(self.lib.lslam_pg_set_row_shard(self.h, int(v_begin), int(v_end)))

    def set_solve_tolerance(self, rel_tol):
        """lslam_pg_set_solve_tolerance: relative residual at which a damped solve's PCG stops (default 1e-8 = the direct
        solver's answer as far as LM can tell; looser = inexact LM, same optimum, another trajectory)."""
        self._build()
        self._check(self.lib.lslam_pg_set_solve_tolerance(self.h, float(rel_tol)))

    def row_sharded_solves(self):
        return int(self.lib.lslam_pg_row_sharded_solves(self.h)) if self.h else 0

    def set_row_gather(self, allgatherv, rank, world):
        """lslam_pg_set_row_gather: `allgatherv(ptr, offsets, world)` gathers in place -- doubles [offsets[r], offsets[r + 1])
        at device address `ptr` are valid on rank r on entry, on every rank on return.  None removes it.  (With the library's
        RCCL communicator nothing needs registering: the gather is taken by itself.)"""
        self._build()
        if allgatherv is None:
            self._gcb = ALLGATHERV_FN(0)
        else:
            def _tramp(_user, ptr, offs, w):
                allgatherv(ptr, [int(offs[i]) for i in range(w + 1)], int(w))
            self._gcb = ALLGATHERV_FN(_tramp)
        self._check(self.lib.lslam_pg_set_row_gather(self.h, self._gcb, None, int(rank), int(world)))

    def row_gathered_solves(self):
        return int(self.lib.lslam_pg_row_gathered_solves(self.h)) if self.h else 0

    def system_doubles(self):
        self._build()
        return self.lib.lslam_pg_system_doubles(self.h)

    # ---- parity taps --------------------------------------------------------------------------
    def linearize(self):
        self._build()
        n, no = len(self._nodes), self.lib.lslam_pg_num_offdiag(self.h)
        diag = np.zeros((n, 6, 6))
        off = np.zeros((no, 6, 6))
        oij = np.zeros((no, 2), np.int32)
        b = np.zeros(6 * n)
        chi = np.zeros(1)
        self._check(self.lib.lslam_pg_linearize(self.h, _dp(diag), _dp(off), oij.ctypes.data_as(c_int32_p),
                                                _dp(b), _dp(chi)))
        return dict(diag=diag, off=off, off_ij=oij, b=b, chi2=float(chi[0]))

    def solve(self, lam):
        self._build()
        dx = np.zeros(6 * len(self._nodes))
        it = C.c_int32()
        self._check(self.lib.lslam_pg_solve(self.h, float(lam), _dp(dx), C.byref(it)))
        return dx, it.value

    # ---- internals ---------------------------------------------------------------------------------
    def _check(self, rc):
        if rc < 0:
            raise LslamError(rc, self.lib.lslam_pg_last_error().decode())
        return rc

    def build(self):
        """Upload the graph and build the device-side structures now (otherwise the first
        optimize()/linearize() does it); the counterpart of g2o's initializeOptimization()."""
        self._build()

    def _drop(self, keep_estimates=True):
        """Release the device graph.  The optimised estimates are pulled back first (g2o keeps its
        vertices' estimates when vertices or edges are added after an optimize())."""
        if self.h:
            if keep_estimates and len(self._nodes):
                out = np.zeros((len(self._nodes), 7))
                if self.lib.lslam_pg_get_poses(self.h, _dp(out)) == 0:
                    self._nodes = [p for p in out]
            self.lib.lslam_pg_destroy(self.h)
            self.h = None

    def _build(self):
        if self.h:
            return
        poses = np.ascontiguousarray(np.stack(self._nodes), np.float64)
        ne = len(self._ij)
        ij = np.ascontiguousarray(np.array(self._ij, np.int32).reshape(ne, 2))
        meas = np.ascontiguousarray(np.stack(self._meas) if ne else np.zeros((0, 7)), np.float64)
        info = np.ascontiguousarray(np.stack(self._info) if ne else np.zeros((0, 6, 6)), np.float64)
        h = C.c_void_p()
        rc = self.lib.lslam_pg_create(self.device, len(poses), _dp(poses), ne, ij.ctypes.data_as(c_int32_p),
                                      _dp(meas), _dp(info), 0, C.byref(h))
        if rc != 0:
            raise LslamError(rc, self.lib.lslam_pg_last_error().decode())
        self.h = h

    def close(self):
        self._drop(keep_estimates=False)

    def __del__(self):
        try:
            self._drop()
        except Exception:
            pass
