// lslam_comm.hip -- RCCL inside the library (include/lslam_c.h, "collectives").
//
// One process per GPU; every rank creates an lslam_comm from the same 128-byte id (made by rank 0
// with lslam_comm_unique_id and handed to the others by whatever the host program has: MPI, a
// file, torch.distributed ...).  The data-path collectives of the two sharded paths -- the 32
// fp64 normal-equation sums of a Gauss-Newton iteration and the block system
// [H | b | chi2] of a pose-graph linearisation (row 3) -- are then ncclAllReduce calls enqueued on
// the library's own stream between the kernels that produce and consume them: no host round trip.
//
// librccl is dlopen'ed on first use ("librccl.so.1": the copy already loaded in the process if there
// is one -- PyTorch ships its own -- else ROCm's), so a single-GPU user has no dependency on it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstring>
#include <mutex>

#include "../../include/lslam_c.h"
#include "lslam_internal.hpp"

static_assert(sizeof(ncclUniqueId) == LSLAM_COMM_ID_BYTES, "lslam_comm id size");

namespace {

struct Rccl {
  void *handle = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclBroadcast) Broadcast = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  decltype(&ncclCommCount) CommCount = nullptr;
  decltype(&ncclCommUserRank) CommUserRank = nullptr;
  decltype(&ncclGetVersion) GetVersion = nullptr;
  bool ok = false;
};

Rccl g_rccl;
std::mutex g_rccl_mu;

bool load_rccl() {
  std::lock_guard<std::mutex> lk(g_rccl_mu);
  if (g_rccl.ok) return true;
  const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char *n : names) {
    g_rccl.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (g_rccl.handle) break;
  }
  if (!g_rccl.handle) {
    lslam::set_error((std::string("cannot load librccl: ") + dlerror()).c_str());
    return false;
  }
#define SYM(f)                                                              \
  g_rccl.f = reinterpret_cast<decltype(g_rccl.f)>(dlsym(g_rccl.handle, "nccl" #f)); \
  if (!g_rccl.f) { lslam::set_error("librccl lacks nccl" #f); return false; }
  SYM(GetUniqueId) SYM(CommInitRank) SYM(CommDestroy) SYM(AllReduce) SYM(GetErrorString) SYM(CommCount) SYM(CommUserRank)
  SYM(Broadcast) SYM(GroupStart) SYM(GroupEnd) SYM(GetVersion)
#undef SYM
  g_rccl.ok = true;
  return true;
}

int nccl_fail(const char *what, ncclResult_t r) {
  char buf[256];
  snprintf(buf, sizeof(buf), "%s failed: %s", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?");
  lslam::set_error(buf);
  return LSLAM_ERR_COMM;
}

}  // namespace

struct lslam_comm {
  ncclComm_t comm = nullptr;
  int device = 0, rank = 0, world = 1;
};

namespace lslam {
hipError_t comm_allreduce_f64(lslam_comm *c, double *buf, size_t count, hipStream_t s) {
  if (!c || !c->comm) return hipErrorInvalidValue;
  const ncclResult_t r = g_rccl.AllReduce(buf, buf, count, ncclDouble, ncclSum, c->comm, s);
  if (r != ncclSuccess) {
    nccl_fail("ncclAllReduce", r);
    return hipErrorUnknown;
  }
  return hipSuccess;
}
// In-place all-gather of UNEQUAL segments (ncclAllGather wants equal ones): one ncclBroadcast per segment, rooted at its owner,
// all inside ONE group -- RCCL fuses a group into a single launch, so the latency is that of one collective.  n_lists buffers
// share the group (the row-sharded PCG gathers the vector z and, behind it, two partial sums per rank).
hipError_t comm_allgatherv_f64(lslam_comm *c, int n_lists, double *const *bufs, const int64_t *const *offs, hipStream_t s) {
  if (!c || !c->comm) return hipErrorInvalidValue;
  ncclResult_t r = g_rccl.GroupStart();
  if (r != ncclSuccess) { nccl_fail("ncclGroupStart", r); return hipErrorUnknown; }
  for (int l = 0; l < n_lists && r == ncclSuccess; ++l)
    for (int root = 0; root < c->world && r == ncclSuccess; ++root) {
      const int64_t len = offs[l][root + 1] - offs[l][root];
      if (len <= 0) continue;
      double *seg = bufs[l] + offs[l][root];
      r = g_rccl.Broadcast(seg, seg, (size_t)len, ncclDouble, root, c->comm, s);
    }
  const ncclResult_t r2 = g_rccl.GroupEnd();
  if (r != ncclSuccess || r2 != ncclSuccess) {
    nccl_fail("ncclBroadcast (grouped all-gather)", r != ncclSuccess ? r : r2);
    return hipErrorUnknown;
  }
  return hipSuccess;
}
int comm_world(const lslam_comm *c) { return c ? c->world : 1; }
int comm_rank(const lslam_comm *c) { return c ? c->rank : 0; }
}  // namespace lslam

extern "C" {

int lslam_comm_unique_id(uint8_t id[LSLAM_COMM_ID_BYTES]) {
  if (!id) return LSLAM_ERR_INVALID;
  if (!load_rccl()) return LSLAM_ERR_COMM;
  ncclUniqueId u;
  const ncclResult_t r = g_rccl.GetUniqueId(&u);
  if (r != ncclSuccess) return nccl_fail("ncclGetUniqueId", r);
  std::memcpy(id, &u, LSLAM_COMM_ID_BYTES);
  return LSLAM_OK;
}

int lslam_comm_create(int device, const uint8_t id[LSLAM_COMM_ID_BYTES], int32_t rank, int32_t world, lslam_comm **out) {
  if (!out || !id || world < 1 || rank < 0 || rank >= world) {
    lslam::set_error("bad communicator arguments");
    return LSLAM_ERR_INVALID;
  }
  *out = nullptr;
  if (!load_rccl()) return LSLAM_ERR_COMM;
  if (hipSetDevice(device) != hipSuccess) {
    lslam::set_error("hipSetDevice failed");
    return LSLAM_ERR_HIP;
  }
  ncclUniqueId u;
  std::memcpy(&u, id, LSLAM_COMM_ID_BYTES);
  lslam_comm *c = new lslam_comm();
  c->device = device;
  c->rank = rank;
  c->world = world;
  const ncclResult_t r = g_rccl.CommInitRank(&c->comm, world, u, rank);
  if (r != ncclSuccess) {
    delete c;
    return nccl_fail("ncclCommInitRank", r);
  }
  *out = c;
  return LSLAM_OK;
}

void lslam_comm_destroy(lslam_comm *c) {
  if (!c) return;
  if (c->comm && g_rccl.ok) {
    (void)hipSetDevice(c->device);
    (void)g_rccl.CommDestroy(c->comm);
  }
  delete c;
}

// what the COMMUNICATOR says (ncclCommUserRank / ncclCommCount), not what it was asked to be: the rank count `bench.py --gpus N`
// prints as `rccl_ranks`
int lslam_comm_version(int32_t *version) {
  if (!version) {
    lslam::set_error("bad version argument");
    return LSLAM_ERR_INVALID;
  }
  if (!load_rccl()) return LSLAM_ERR_COMM;
  int v = 0;
  const ncclResult_t r = g_rccl.GetVersion(&v);
  if (r != ncclSuccess) return nccl_fail("ncclGetVersion", r);
  *version = (int32_t)v;
  return LSLAM_OK;
}

int lslam_comm_info(const lslam_comm *c, int32_t *rank, int32_t *world) {
  if (!c || !c->comm || !g_rccl.ok) return LSLAM_ERR_INVALID;
  int r = -1, w = -1;
  ncclResult_t e = g_rccl.CommUserRank(c->comm, &r);
  if (e != ncclSuccess) return nccl_fail("ncclCommUserRank", e);
  e = g_rccl.CommCount(c->comm, &w);
  if (e != ncclSuccess) return nccl_fail("ncclCommCount", e);
  if (rank) *rank = r;
  if (world) *world = w;
  return LSLAM_OK;
}

int lslam_comm_allreduce_f64(lslam_comm *c, double *device_buf, size_t count, void *hip_stream) {
  if (!c || !device_buf) return LSLAM_ERR_INVALID;
  if (hipSetDevice(c->device) != hipSuccess) return LSLAM_ERR_HIP;
  return lslam::comm_allreduce_f64(c, device_buf, count, (hipStream_t)hip_stream) == hipSuccess ? LSLAM_OK : LSLAM_ERR_COMM;
}

}  // extern "C"
