"""Data-parallel pieces of the hot path: one process per GPU, torch.distributed (backend "nccl" is
RCCL on ROCm; "gloo" in the CPU tests).

  * training: ONE all-reduce(SUM) per step over the flat buffer [gradients | presence map | loss] instead of the
    reference's per-tensor loop (pyprob/nn/inference_network.py:296-333), then every rank divides by world size;
  * minibatch partition: the bucketed sampler of pyprob/nn/dataset.py:328-400 (all ranks walk the same bucket at the
    same time, each rank takes a disjoint stride of its minibatches);
  * importance sampling: particles are independent -> contiguous shards per rank with disjoint Philox counters, no
    collective on the data path (pyprob/model.py:339-406 shards the same way over processes).
"""
import math

import numpy as np


def allreduce_flat_(buf):
    """all-reduce(SUM) of the flat [grads | presence | loss] buffer, in place."""
    import torch.distributed as dist
    dist.all_reduce(buf)
    return buf


def allreduce_pieces_(views):
    """all-reduce(SUM) of several views of one flat buffer, in place, as one launch where the backend can coalesce
    (RCCL: ncclGroupStart / ncclGroupEnd around the calls), otherwise one collective per view (gloo in the CPU tests)."""
    import torch.distributed as dist
    if len(views) == 1:
        dist.all_reduce(views[0])
        return
    if _coalescing_works(views[0].device):
        with dist._coalescing_manager(device=views[0].device):
            for v in views:
                dist.all_reduce(v)
        return
    for v in views:
        dist.all_reduce(v)


_COALESCE = {}


def _coalescing_works(device):
    """Probed once per process on two scratch tensors (never on the gradients: a half-issued group must not be retried)."""
    import torch
    import torch.distributed as dist
    key = (dist.get_backend(), str(device))
    if key not in _COALESCE:
        ok = False
        if dist.get_backend() == 'nccl' and hasattr(dist, '_coalescing_manager'):
            try:
                a, b = torch.ones(4, device=device), torch.ones(4, device=device)
                with dist._coalescing_manager(device=device):
                    dist.all_reduce(a)
                    dist.all_reduce(b)
                ok = bool(torch.allclose(a, torch.full_like(a, float(dist.get_world_size()))) and torch.equal(a, b))
            except (RuntimeError, TypeError, ValueError):
                ok = False
        _COALESCE[key] = ok
    return _COALESCE[key]


_native_poisoned = [False]      # a native communicator init failed / timed out in this process: never retried


def init_native_comm(device, lib=None):
    """This library's own RCCL communicator over the ranks of the default torch.distributed group (csrc/dp.hip): rank 0
    creates the ncclUniqueId, a broadcast hands it to every rank, all ranks call ncclCommInitRank on their device; the
    outcome is agreed on with a MIN all-reduce, so either every rank uses the native exchange or none does. Returns True
    when the communicator is up (pp_dp_world() == world size)."""
    import ctypes as C
    import os
    import torch
    import torch.distributed as dist
    from . import lib as L
    lib = lib or L.load()
    world, rank = dist.get_world_size(), dist.get_rank()
    # the early outs are agreed on by ALL ranks (a rank whose earlier init timed out must not return alone while its peers
    # enter the collective init below): poisoned anywhere -> nobody retries; up everywhere -> everybody reuses
    st = torch.tensor([0 if _native_poisoned[0] else 1, 1 if lib.pp_dp_world() == world else 0], dtype=torch.int32, device=device)
    dist.all_reduce(st, op=dist.ReduceOp.MIN)
    healthy, up = (int(v) for v in st.tolist())
    if not healthy:
        return False
    if up:
        return True
    path = os.environ.get('PP_RCCL_PATH') or os.path.join(os.path.dirname(torch.__file__), 'lib', 'librccl.so')
    ident = torch.zeros(128, dtype=torch.uint8)
    ok = os.path.exists(path)
    if ok and rank == 0:
        raw = (C.c_char * 128)()
        ok = lib.pp_dp_unique_id(path.encode(), raw) == 0
        ident = torch.frombuffer(bytearray(raw.raw), dtype=torch.uint8).clone()
    ident = ident.to(device)
    dist.broadcast(ident, 0)
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)        # rank 0 could not even create the id: nobody enters the collective init
    if int(flag.item()) == 0:
        return False
    torch.cuda.set_device(device)
    raw = ident.cpu().numpy().tobytes()
    # ncclCommInitRank is a collective: in a thread with a deadline, so that a rank whose peers never arrive falls back to
    # the torch.distributed exchange instead of hanging the job (ctypes releases the GIL during the call)
    import threading
    box = {}

    def _init():
        torch.cuda.set_device(device)
        box['rc'] = lib.pp_dp_init(path.encode(), raw, rank, world)
    th = threading.Thread(target=_init, daemon=True)
    th.start()
    th.join(float(os.environ.get('PP_DP_INIT_TIMEOUT', '90')))
    ok = (not th.is_alive()) and box.get('rc') == 0
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if int(flag.item()) == 0:
        # Some rank failed or timed out. Its init thread may still be inside ncclCommInitRank and could install a
        # communicator later; peers must not destroy theirs while it may still be in the collective. The library's native
        # exchange is never tried again in this process (ICEngine.native_dp stays False: torch.distributed exchanges).
        _native_poisoned[0] = True
        return False
    return True


def finish_reduce(buf, n_params, n_tensors, world_size):
    """After the all-reduce: averaged gradients (inference_network.py:324-325), merged presence map (:300-315: a tensor
    is updated if ANY rank produced a gradient) and mean loss (:327-333). Returns views (grads, active, loss)."""
    grads = buf[:n_params]
    grads /= float(world_size)
    active = buf[n_params:n_params + n_tensors]
    loss = buf[n_params + n_tensors:] / float(world_size)
    return grads, active, loss


def shard_range(n, rank, world):
    """Contiguous particle shard of this rank: (offset, count), counts differ by at most one."""
    base, rem = divmod(n, world)
    count = base + (1 if rank < rem else 0)
    offset = rank * base + min(rank, rem)
    return offset, count


def gather_particles(values, log_weights, num_traces):
    """Every rank contributes its particle shard (shard_range order) and receives all `num_traces` (values, log-weights):
    what ParallelModel does with per-process files and a concatenated Empirical (pyprob/model.py:395-404), as ONE
    all-gather of [value | log-weight] pairs (shards differ by at most one particle: padded to the longest)."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(), dist.get_rank()
    counts = [shard_range(num_traces, r, world)[1] for r in range(world)]
    if values.numel() != counts[rank] or log_weights.numel() != counts[rank]:
        raise ValueError('rank %d holds %d particles, its shard has %d' % (rank, values.numel(), counts[rank]))
    width = max(counts)
    mine = torch.zeros(2, width, dtype=torch.float32, device=values.device)
    mine[0, :counts[rank]] = values.reshape(-1)
    mine[1, :counts[rank]] = log_weights.reshape(-1)
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine)
    v = torch.cat([p[0, :c] for p, c in zip(parts, counts)])
    lw = torch.cat([p[1, :c] for p, c in zip(parts, counts)])
    return v, lw


def _seed0_drop_positions(n, drop):
    """Positions (in the original list of n items) that `random.seed(0); for _ in range(drop): del l[random.randrange(len(l))]`
    deletes - util.drop_items as DistributedTraceBatchSampler calls it - without the O(n) list deletions: the k-th item of
    the current list is the smallest original position p with p - #(deleted <= p) == k."""
    import bisect
    import random
    rnd = random.Random(0)
    removed = []
    for i in range(drop):
        k = rnd.randrange(n - i)
        p = k
        while True:
            q = k + bisect.bisect_right(removed, p)
            if q == p:
                break
            p = q
        bisect.insort(removed, p)
    return np.asarray(removed, np.int64)


class DistributedTraceBatchSampler:
    """The minibatch partition of pyprob/nn/dataset.py:328-400 as array views: the sorted trace indices become ONE int64
    table [n_batches, batch_size]; buckets are row ranges of it, a rank's share of a bucket is a strided row slice.
    Same minibatches as the reference: the reference's seed-0 random subset of traces is dropped so that the number of minibatches is a
    multiple of the world size, the last short bucket is merged into its predecessor, all ranks walk the buckets in the
    same (epoch-seeded) order and take `floor(len / world)` minibatches of each, rank r the rows r, r + world, ..."""

    def __init__(self, sorted_indices, batch_size, rank, world_size, num_buckets=None, shuffle_batches=True,
                 shuffle_buckets=True):
        self._world_size, self._rank = int(world_size), int(rank)
        idx = np.asarray(sorted_indices, np.int64).reshape(-1)
        drop = ((len(idx) // batch_size) % self._world_size) * batch_size
        if drop:       # every rank drops the same traces: the reference's seed-0 choice (dataset.py:337-343, util.py:426-432)
            idx = np.delete(idx, _seed0_drop_positions(len(idx), drop))
        n_batches = len(idx) // batch_size           # a short last minibatch is dropped (dataset.py:345-346)
        if n_batches == 0:
            raise RuntimeError('dataset too small for batch_size:{} and world_size:{}'.format(batch_size, world_size))
        self._table = idx[:n_batches * batch_size].reshape(n_batches, batch_size)
        self._num_buckets = n_batches / self._world_size if num_buckets is None else num_buckets
        self._bucket_size = math.ceil(n_batches / self._num_buckets)
        if self._bucket_size < self._world_size:
            raise RuntimeError('batch_size:{} and num_buckets:{} imply a bucket_size:{} smaller than world_size:{}'.format(
                batch_size, self._num_buckets, self._bucket_size, world_size))
        starts = np.arange(0, n_batches, self._bucket_size)
        ends = np.minimum(starts + self._bucket_size, n_batches)
        if ends[-1] - starts[-1] < self._bucket_size and n_batches % self._bucket_size:
            if len(starts) < 2:
                raise RuntimeError('dataset too small for given batch_size:{} and num_buckets:{}'.format(batch_size, num_buckets))
            starts, ends = starts[:-1], np.concatenate([ends[:-2], ends[-1:]])      # the short tail joins its predecessor
        self._bounds = np.stack([starts, ends], 1)                                   # [n_buckets, 2] row ranges of the table
        self._shuffle_batches, self._shuffle_buckets = shuffle_batches, shuffle_buckets
        self._epoch = 0
        self._current_bucket_id = 0

    @property
    def _batches(self):
        return list(self._table)

    @property
    def _buckets(self):
        return [list(self._table[a:b]) for a, b in self._bounds]

    def __iter__(self):
        self._epoch += 1
        order = np.arange(len(self._bounds))
        if self._shuffle_buckets:
            np.random.RandomState(self._epoch).shuffle(order)      # same order on every rank
        for bucket_id in order:
            a, b = self._bounds[bucket_id]
            self._current_bucket_id = int(bucket_id)
            mine = self._table[a + self._rank:b:self._world_size][:(b - a) // self._world_size]
            if self._shuffle_batches:
                mine = mine[np.random.permutation(len(mine))]
            yield from mine

    def __len__(self):
        return len(self._table)
