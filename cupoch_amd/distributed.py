"""Multi-GPU ICP: one process per GPU (torch.distributed for rendezvous, RCCL for
the per-iteration all-reduce, issued natively by libmi_icp.so on its own stream).

The reference is single-GPU; this is the partitioning BASELINE.json asks for:
every rank holds the full target (+ LBVH) and a spatially contiguous shard of
the source; each iteration all-reduces the 32 fp64 values of the reduced system
(256 bytes) and every rank solves the same 6x6 on its host, so no broadcast is
needed and all ranks take identical steps.
"""
import numpy as np


def shard_bounds(n, rank, world):
    """Contiguous [lo, hi) of rank's share of n items (sizes differ by at most 1)."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    base, rem = divmod(int(n), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _spread3(v):
    v = v.astype(np.uint64) & np.uint64(0x1fffff)
    v = (v | (v << np.uint64(32))) & np.uint64(0x1f00000000ffff)
    v = (v | (v << np.uint64(16))) & np.uint64(0x1f0000ff0000ff)
    v = (v | (v << np.uint64(8))) & np.uint64(0x100f00f00f00f00f)
    v = (v | (v << np.uint64(4))) & np.uint64(0x10c30c30c30c30c3)
    v = (v | (v << np.uint64(2))) & np.uint64(0x1249249249249249)
    return v


def morton_order(points, bits=10):
    """Host-side Morton order of an (n,3) cloud (stable): the order in which the
    source is cut into per-rank shards, so that each GPU walks one compact region
    of the target tree."""
    p = np.asarray(points, np.float32).reshape(-1, 3)
    if len(p) == 0:
        return np.zeros(0, np.int64)
    mn = p.min(0)
    ext = float((p.max(0) - mn).max())
    scale = (float(1 << bits) / ext) if ext > 0 else 0.0
    q = np.clip((p - mn) * np.float32(scale), 0, (1 << bits) - 1).astype(np.uint32)
    key = (_spread3(q[:, 0]) << np.uint64(2)) | (_spread3(q[:, 1]) << np.uint64(1)) | _spread3(q[:, 2])
    return np.argsort(key, kind="stable")


def shard_source(points, rank, world, order=None):
    """Indices (into the original source) of rank's spatial shard."""
    order = morton_order(points) if order is None else order
    lo, hi = shard_bounds(len(order), rank, world)
    return np.sort(order[lo:hi])


def exchange_unique_id(make_id, rank, device=None):
    """Rank 0 creates the 128-byte RCCL unique id, everyone receives it through
    the already-initialised torch.distributed group (gloo or nccl)."""
    import torch
    import torch.distributed as dist
    buf = torch.zeros(128, dtype=torch.uint8)
    if rank == 0:
        buf = torch.frombuffer(bytearray(make_id()), dtype=torch.uint8).clone()
    if dist.get_backend() == "nccl":
        buf = buf.to(device if device is not None else torch.device("cuda", torch.cuda.current_device()))
    dist.broadcast(buf, src=0)
    return bytes(buf.cpu().numpy().tobytes())


def init_engine_comm(engine, n_source_total):
    """Attach an Engine to the job's ranks: RCCL communicator + global source size
    (the fitness denominator, registration.cu:76)."""
    import torch.distributed as dist
    from .engine import comm_unique_id
    rank, world = dist.get_rank(), dist.get_world_size()
    uid = exchange_unique_id(comm_unique_id, rank)
    engine.comm_init(uid, world, rank)
    engine.set_global_source_count(n_source_total)
    return rank, world


def init_engine_comm_local(engine, n_source_total):
    """The node-local communicator alone (shared-memory mailbox, no RCCL): rank 0 picks the job's
    name, everyone receives it through the already-initialised torch.distributed group."""
    import os
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(), dist.get_world_size()
    tag = torch.zeros(2, dtype=torch.int64)
    if rank == 0:
        tag = torch.tensor([os.getpid(), int.from_bytes(os.urandom(6), "little")], dtype=torch.int64)
    if dist.get_backend() == "nccl":
        tag = tag.to(torch.device("cuda", torch.cuda.current_device()))
    dist.broadcast(tag, src=0)
    tag = tag.cpu()
    engine.comm_init_local("job_%d_%x" % (int(tag[0]), int(tag[1])), world, rank)
    engine.set_global_source_count(n_source_total)
    return rank, world


def gather_correspondences(local_pairs, shard_indices):
    """The job's correspondence set from the ranks' shard-local ones: source indices are
    mapped back through the shard (shard_source's result), everything is all-gathered and
    ordered ascending in source index, as registration.cu:62-69 leaves a single-GPU set.
    Every rank returns the same (n, 2) int32 array."""
    import torch
    import torch.distributed as dist
    pairs = np.asarray(local_pairs, np.int32).reshape(-1, 2).copy()
    if len(pairs):
        pairs[:, 0] = np.asarray(shard_indices, np.int64)[pairs[:, 0]].astype(np.int32)
    device = None
    if dist.get_backend() == "nccl":
        device = torch.device("cuda", torch.cuda.current_device())
    count = torch.tensor([len(pairs)], dtype=torch.int64, device=device)
    counts = [torch.zeros_like(count) for _ in range(dist.get_world_size())]
    dist.all_gather(counts, count)
    cap = int(max(int(c.item()) for c in counts))
    buf = torch.full((max(cap, 1), 2), -1, dtype=torch.int32, device=device)
    if len(pairs):
        buf[: len(pairs)] = torch.from_numpy(pairs).to(buf.device)
    parts = [torch.empty_like(buf) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, buf)
    out = np.concatenate([p.cpu().numpy()[: int(c.item())] for p, c in zip(parts, counts)], axis=0)
    return out[np.argsort(out[:, 0], kind="stable")] if len(out) else out.reshape(0, 2)


def allreduce_system(sys32):
    """Host-side equivalent of the in-library all-reduce, for any backend
    (used by the CPU/gloo tests and by Python-level estimators)."""
    import torch
    import torch.distributed as dist
    t = torch.from_numpy(np.ascontiguousarray(sys32, np.float64).copy())
    if dist.get_backend() == "nccl":   # RCCL reduces device tensors only
        t = t.to(torch.device("cuda", torch.cuda.current_device()))
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy()


class HostDrivenLoop:
    """The registration loop driven from the host, one rank of a sharded run: search + reduction on
    this rank's shard (HIP kernels), torch.distributed all-reduce of the 32 doubles (any backend:
    gloo between processes that share one GPU, nccl = RCCL between GPUs), the 6x6 solve on the
    host, identical on every rank.  One host round trip per iteration -- the fallback for when the
    in-library RCCL communicator (init_engine_comm: all-reduce and step stay on the device) is
    not available, and the form the two-processes-on-one-GPU test drives.

    Same stepping interface as Engine.icp_begin / icp_iterate."""

    def __init__(self, engine, est, max_distance, n_source_total, det_thresh=-1.0, init=None, world=None):
        import torch.distributed as dist
        self.eng, self.est, self.max_distance = engine, int(est), float(max_distance)
        self.det_thresh = float(det_thresh)
        self.n_total = int(n_source_total)
        self.world = (dist.get_world_size() if dist.is_initialized() else 1) if world is None else int(world)
        self.T = np.eye(4, dtype=np.float32) if init is None else np.asarray(init, np.float32).copy()
        self.fitness = self.inlier_rmse = 0.0
        self.iterations = 0
        engine.set_global_source_count(self.n_total)

    def _evaluate(self):
        """correspondences under self.T; the job's fitness / rmse from the summed statistics"""
        self.eng.evaluate_registration(self.max_distance, self.T)
        sys32 = self.eng.compute_system(self.est, self.T)
        if self.world > 1:
            sys32 = allreduce_system(sys32)
        cnt = sys32[29]
        self.fitness = float(np.float32(cnt) / np.float32(self.n_total)) if cnt > 0 and self.n_total > 0 else 0.0
        self.inlier_rmse = float(np.sqrt(np.float32(sys32[28]) / np.float32(cnt))) if cnt > 0 else 0.0
        return sys32

    def begin(self):
        self.sys32 = self._evaluate()
        return self

    def iterate(self, k=1):
        from .engine import kabsch_from_sums, solve_system
        for _ in range(int(k)):
            if self.est == 1:                                  # point-to-point: Kabsch from the sums
                upd = kabsch_from_sums(self.sys32, self.n_total)
            else:
                ok, upd = solve_system(self.sys32, self.det_thresh)
                if self.est == 3 and ok:                        # symmetric: the half rotation applied twice
                    full = np.eye(4, dtype=np.float32)
                    full[:3, :3] = (upd[:3, :3].astype(np.float64) @ upd[:3, :3].astype(np.float64)).astype(np.float32)
                    full[:3, 3] = upd[:3, 3]
                    upd = full
            self.T = (upd @ self.T).astype(np.float32)
            self.iterations += 1
            self.sys32 = self._evaluate()
        return self


def device_shard_source(engine, points, rank, world):
    """Original indices (ascending) of rank's spatial shard, cut from the Morton order the ENGINE
    computes on the device (no host sort of the whole cloud on every rank).  points: (n, 3)
    numpy array or torch tensor on the engine's device; returns an int64 numpy array."""
    order = engine.morton_order(points)
    lo, hi = shard_bounds(len(order), rank, world)
    return np.sort(order[lo:hi].astype(np.int64))
