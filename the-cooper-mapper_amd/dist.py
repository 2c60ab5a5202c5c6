"""Multi-GPU plumbing for the scan-match path: one process per GPU, independent scans
sharded across ranks, no data-path collective.

torch.distributed (backend "nccl" = RCCL on ROCm, "gloo" in the CPU tests) is used only
for the rendezvous, the barrier that brackets the timed region and the reduction of the
per-rank counters (sum) and wall time (max)."""
import os


def env_rank():
    """(rank, device index, world).  LSLAM_FORCE_DEVICE pins the device index (pre-flight runs of the
    multi-rank code paths on a box with fewer GPUs than ranks, together with LSLAM_DIST_BACKEND=gloo)."""
    dev = os.environ.get("LSLAM_FORCE_DEVICE")
    return (int(os.environ.get("RANK", "0")), int(dev) if dev is not None else int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def shard_range(n_items, rank, world):
    """Contiguous, balanced [begin, end) of `n_items` independent scans for `rank`."""
    base, extra = divmod(n_items, world)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def row_shard_range(n_vertices, rank, world, block=21):
    """This rank's vertex rows [begin, end) of a row-sharded pose-graph solve: an even partition in whole row blocks of the
    PCG kernels (21 vertices = 126 rows) -- the Python statement of lslam_pg_row_shard_range (include/lslam_c.h)."""
    nblk = (n_vertices + block - 1) // block
    b0, b1 = nblk * rank // world, nblk * (rank + 1) // world
    return min(n_vertices, b0 * block), min(n_vertices, b1 * block)


def init(backend=None, device=None):
    """Initialise the process group from the torchrun environment (no-op for 1 rank)."""
    rank, local_rank, world = env_rank()
    if world == 1:
        return None
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    backend = os.environ.get("LSLAM_DIST_BACKEND", backend)
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if torch.cuda.is_available():
        torch.cuda.set_device(local_rank)
    kw = {}
    if backend == "nccl":
        torch.cuda.set_device(local_rank)
        kw["device_id"] = torch.device("cuda", local_rank)
    dist.init_process_group(backend, **kw)
    return dist


def barrier(dist):
    import torch
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    if torch.cuda.is_available():
        torch.cuda.synchronize()


def aggregate(dist, sums, elapsed):
    """All ranks: element-wise SUM of `sums` (list of floats) and MAX of `elapsed`."""
    import torch
    dev = "cuda" if (torch.cuda.is_available() and (dist is None or dist.get_backend() == "nccl")) else "cpu"
    t_sum = torch.tensor([float(x) for x in sums], dtype=torch.float64, device=dev)
    t_max = torch.tensor([float(elapsed)], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(t_sum, op=dist.ReduceOp.SUM)
        dist.all_reduce(t_max, op=dist.ReduceOp.MAX)
    return t_sum.tolist(), float(t_max.item())
