"""Multi-GPU particle-filter sensor update: one process per GPU, particles block-partitioned, mesh + BVH
replicated, ONE collective per update -- an all-gather of the per-particle likelihood means over
RCCL/xGMI.  The reference has no distributed code at all; this is new.

The all-gather payload is 4 B x N (1 M particles: 4 MB total, 0.5 MB per rank on 8 GPUs): with 7
point-to-point xGMI links per GPU this is latency-bound (tens of microseconds), so it is issued as a
single collective on equal-sized (padded) shards, never bucketed or split.

Works with backend "nccl" (= RCCL) on GPUs and "gloo" on CPU (tests).
"""
import numpy as np


def shard_bounds(n, rank, world):
    """contiguous block partition of [0, n): the first n % world ranks own one extra particle."""
    base, rem = divmod(int(n), int(world))
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def shard_capacity(n, world):
    """padded per-rank shard length (equal on all ranks, as all_gather requires)."""
    return (int(n) + int(world) - 1) // int(world)


class _GatherBuffers:
    """send / receive / dense buffers of one (n_total, world, device, dtype) all-gather, allocated ONCE: a 4 MB
    latency-bound collective must not pay three allocations and a torch.cat per step."""

    _cache = {}

    def __init__(self, n_total, world, device, dtype, rec):
        import torch
        self.cap = shard_capacity(n_total, world)
        shape = (self.cap,) if rec == 0 else (self.cap, rec)
        self.send = torch.zeros(shape, dtype=dtype, device=device)
        self.recv = torch.empty((world * self.cap,) + shape[1:], dtype=dtype, device=device)
        self.ragged = (n_total % world) != 0
        self.dense = torch.empty((n_total,) + shape[1:], dtype=dtype, device=device) if self.ragged else None
        self.bounds = [shard_bounds(n_total, r, world) for r in range(world)]

    @classmethod
    def get(cls, n_total, world, device, dtype, rec=0):
        key = (int(n_total), int(world), str(device), dtype, int(rec))
        b = cls._cache.get(key)
        if b is None:
            b = cls._cache[key] = cls(int(n_total), int(world), device, dtype, int(rec))
        return b

    def gather(self, local, group):
        import torch.distributed as dist
        self.send[: local.shape[0]].copy_(local)
        dist.all_gather_into_tensor(self.recv, self.send, group=group)
        if not self.ragged:
            return self.recv          # equal shards: the receive buffer IS the dense vector (no copy)
        for r, (lo, hi) in enumerate(self.bounds):
            self.dense[lo:hi].copy_(self.recv[r * self.cap: r * self.cap + (hi - lo)])
        return self.dense


def allgather_weights(local_weights, n_total, group=None):
    """local_weights: 1-D float32 torch tensor holding this rank's shard (device tensor for RCCL).
    Returns the dense [n_total] weight vector, identical on every rank.  ONE all_gather_into_tensor on equal-sized
    (padded) shards into preallocated buffers; the result aliases a cached buffer that the next call overwrites."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    return _GatherBuffers.get(n_total, world, local_weights.device, torch.float32).gather(local_weights, group)


def gathered_sum_max(weights_all, stats_fn=None):
    """global {sum, max} of the weights WITHOUT a second collective (round 6): after the all-gather every rank holds the whole
    vector and reduces it itself, in a fixed order -- every rank gets the same bits, and they do not depend on the order in which a
    collective library would have added per-rank partials (the value feeds size_t(L / sum * N) in the residual resampler,
    resampling.cu:41-92, ResidualResamplerCPU.cpp:112-118).  stats_fn(weights_all, n) -> {"sum", "max"}: the device kernel
    (GladiatorResamplerHip.compute_stats_weights: the single-GPU statistics bit for bit); None: numpy on the host (CPU tests)."""
    n = int(weights_all.shape[0])
    if stats_fn is not None:
        st = stats_fn(weights_all, n)
        return float(st["sum"]), float(st["max"])
    w = weights_all.detach().cpu().numpy() if hasattr(weights_all, "detach") else np.asarray(weights_all)
    return float(np.float32(w.astype(np.float64).sum())), float(max(np.float32(0.0), w.max())) if n else (0.0, 0.0)


def allreduce_sum_max(local_weights, group=None):
    """(rounds 2-5, kept for A/B) global {sum, max} as two tiny all-reduces of per-rank partials: the sum then depends on the
    collective library's order of summation -- gathered_sum_max does not.
    The distributed form of the reference's simple_stats_kernel (rmcl_ros/src/rmcl/resampling.cu:41-92)."""
    import torch
    import torch.distributed as dist
    s = local_weights.to(torch.float64).sum().reshape(1)
    m = local_weights.max().reshape(1).clone() if local_weights.numel() else torch.full((1,), -np.inf, device=local_weights.device)
    dist.all_reduce(s, op=dist.ReduceOp.SUM, group=group)
    dist.all_reduce(m, op=dist.ReduceOp.MAX, group=group)
    return float(s.item()), float(m.item())


class ShardedSensorUpdate:
    """rank-local PCDSensorUpdaterHip over this rank's block of the particle cloud + the weight all-gather."""

    def __init__(self, updater, n_total, rank, world):
        self.updater = updater
        self.n_total, self.rank, self.world = int(n_total), int(rank), int(world)
        self.lo, self.hi = shard_bounds(n_total, rank, world)

    @property
    def n_local(self):
        return self.hi - self.lo

    def update(self, poses_dev, attrs_dev, weights_local):
        """poses_dev / attrs_dev: this rank's shard on the device; weights_local: float32 torch CUDA tensor
        of n_local elements that receives likelihood.mean.  Returns the gathered [n_total] tensor."""
        self.updater.update(poses_dev, attrs_dev, n_particles=self.n_local)
        self.updater.extract_weights(attrs_dev, self.n_local, weights_local.data_ptr())
        return allgather_weights(weights_local, self.n_total)


def allgather_records(local_records, n_total, group=None):
    """All-gather of fixed-size records (poses 32 B, attributes 36 B): local_records is a [n_local, rec_bytes]
    uint8 torch tensor holding this rank's shard; returns the dense [n_total, rec_bytes] tensor, identical on every
    rank.  One collective on equal-sized padded shards (C5: 68 MB over 8 ranks, 8.5 MB per rank contribution) into
    preallocated buffers (the result aliases a cached buffer that the next call with the same shape overwrites)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    return _GatherBuffers.get(n_total, world, local_records.device, torch.uint8, local_records.shape[1]).gather(local_records, group)


class ShardedResample:
    """Distributed gladiator tournament: every rank owns the champions of its block but the
    enemy may live anywhere, so the cloud (68 B per particle) is all-gathered once per resampling step and each
    rank then resamples its own block against the gathered copy.  Because the random stream is a function of the
    GLOBAL champion index (Philox counter), the result is identical to the single-GPU tournament.

    resample_fn(poses_all, attrs_all, n_total, first, count) -> (poses_new_local, attrs_new_local) as
    [count, 32] / [count, 36] uint8 tensors (GladiatorResamplerHip on GPUs, the oracle in the CPU tests)."""

    def __init__(self, resample_fn, n_total, rank, world):
        self.resample_fn = resample_fn
        self.n_total, self.rank, self.world = int(n_total), int(rank), int(world)
        self.lo, self.hi = shard_bounds(n_total, rank, world)

    def update(self, poses_local, attrs_local, group=None):
        poses_all = allgather_records(poses_local, self.n_total, group)
        attrs_all = allgather_records(attrs_local, self.n_total, group)
        return self.resample_fn(poses_all, attrs_all, self.n_total, self.lo, self.hi - self.lo)


class ShardedBatchCorrector:
    """MICP pose batches over ranks (north_star: pose-corrections/s at 1/2/4/8 GPUs; SURVEY.md 8(e): "shard poses, no exchange").
    Every rank holds the same operator configuration (map replicated, same model / dataset / parameters) and the same pose list;
    rank r corrects the block shard_bounds(nposes, r, world) with its own `correct_fn(Tbm_block) -> (Tdelta_block, stats_block)`
    (RCCHip*.correct_batch on GPUs, the oracle in the CPU tests).  The data path has NO collective: a pose's correction does not
    depend on any other pose.  `gather=True` adds ONE all-gather of the 32-B deltas (1000 poses: 32 kB) for callers that want the
    whole vector on every rank (the v1 benchmark's T_curr = multNxN(T_curr, Tdelta) on one host)."""

    def __init__(self, correct_fn, rank, world):
        self.correct_fn = correct_fn
        self.rank, self.world = int(rank), int(world)

    def correct(self, Tbm_all, gather=False, group=None):
        n = len(Tbm_all)
        lo, hi = shard_bounds(n, self.rank, self.world)
        Td, st = self.correct_fn(Tbm_all[lo:hi]) if hi > lo else (Tbm_all[:0].copy(), None)
        if not gather:
            return lo, hi, Td, st
        import torch
        rec = np.ascontiguousarray(Td).view(np.uint8).reshape(hi - lo, 32)
        dense = allgather_records(torch.from_numpy(rec.copy()), n, group)
        return lo, hi, dense.numpy().reshape(-1).view(Td.dtype).copy(), st
