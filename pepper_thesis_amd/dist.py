"""Multi-GPU plumbing: region sharding and the single exchange step (gather of predictions).

The reference distributes by process fan-out only: interval i goes to worker i % threads
(pepper_variant/modules/python/ImageGenerationUI.py:211) and files to callers i % callers
(RunInference.py:101-106); every caller writes its own prediction file and no gather exists. Here one
process drives one GPU, regions are dealt the same round-robin way, and the per-window predictions
are gathered to ONE rank (`dst`) with a gather, not an all-gather: the other ranks receive nothing
(RCCL send/recv when the backend is nccl; gloo in CPU tests). The same exchange is exported from the
C-ABI as pv_gather (include/pepper_hip.h) for hosts that do not carry torch.distributed.
"""
from typing import List, Optional, Sequence

import torch
import torch.distributed as dist


def shard_regions(n_items: int, rank: int, world: int) -> List[int]:
    """indices of the intervals this rank owns: i % world == rank (ImageGenerationUI.py:211)"""
    return [i for i in range(n_items) if i % world == rank]


def _gather_padded(t: torch.Tensor, m: int, dst: int, world: int, rank: int):
    """pad `t` (rows along dim 0) to m rows and gather to dst; returns the list of per-rank padded tensors on dst"""
    pad = torch.zeros((m,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    pad[: t.shape[0]] = t
    bufs = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
    dist.gather(pad, bufs, dst=dst)
    return bufs


class GatherCapacityError(RuntimeError):
    """more rows arrive than the destination has room for; raised on EVERY rank (the decision is collective)"""

    def __init__(self, total, capacity, counts):
        super().__init__("gather: %d rows arrive, the destination holds %d" % (total, capacity))
        self.total, self.capacity, self.counts = total, capacity, counts


def gather_predictions(local: torch.Tensor, dst: int = 0, keys: Optional[torch.Tensor] = None,
                       capacity_rows: Optional[int] = None):
    """Gather [n_r, C] float rows (and optional int64 keys [n_r]) from every rank to rank `dst`.

    Ranks may hold different row counts: the counts are all-gathered (16 B per rank, every rank needs the
    padded size), the payloads are padded to the maximum and moved with ONE gather to `dst`
    (~12-20 B per window; only `dst` allocates receive buffers).
    capacity_rows (meaningful on `dst`; None = unbounded) travels WITH the counts, exactly as in pv_gather
    (csrc/pv_comm.hip): when the rows do not fit, every rank raises GatherCapacityError before any payload
    collective is entered, so no rank is left inside a gather its partner never joins.
    Returns (rows [sum n_r, C], keys or None, counts) on dst, None elsewhere."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        if capacity_rows is not None and int(local.shape[0]) > int(capacity_rows):
            raise GatherCapacityError(int(local.shape[0]), int(capacity_rows), [int(local.shape[0])])
        return local, keys, [int(local.shape[0])]
    world, rank = dist.get_world_size(), dist.get_rank()
    if dist.get_backend() != "nccl" and local.is_cuda:  # gloo rehearsal with device tensors: collective on the host
        local = local.cpu()
        keys = None if keys is None else keys.cpu()
    dev = local.device
    unbounded = (1 << 62)
    cap = (unbounded if capacity_rows is None else int(capacity_rows)) if rank == dst else -1
    n = torch.tensor([local.shape[0], cap], dtype=torch.int64, device=dev)
    pairs = torch.zeros(2 * world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(pairs, n)
    pairs_l = [int(c) for c in pairs.tolist()]
    counts_l = pairs_l[0::2]
    if sum(counts_l) > pairs_l[2 * dst + 1]:  # same verdict on every rank
        raise GatherCapacityError(sum(counts_l), pairs_l[2 * dst + 1], counts_l)
    m = max(max(counts_l), 1)
    row_bufs = _gather_padded(local, m, dst, world, rank)
    key_bufs = None if keys is None else _gather_padded(keys.to(torch.int64), m, dst, world, rank)
    if rank != dst:
        return None
    rows = torch.cat([row_bufs[r][: counts_l[r]] for r in range(world)])
    ks = None if key_bufs is None else torch.cat([key_bufs[r][: counts_l[r]] for r in range(world)])
    return rows, ks, counts_l


def merge_sharded(rows_per_rank: Sequence[torch.Tensor], idx_per_rank: Sequence[Sequence[int]], n_items: int):
    """undo shard_regions for one-row-per-item payloads (used by tests)"""
    out = [None] * n_items
    for rows, idx in zip(rows_per_rank, idx_per_rank):
        for k, i in enumerate(idx):
            out[i] = rows[k]
    return out


class CabiGather:
    """The same exchange through the C-ABI (pv_comm_* / pv_gather, csrc/pv_comm.hip): RCCL driven from the library itself, for
    hosts that do not carry torch.distributed. The 128-byte communicator id is created on rank 0 and distributed by
    `exchange_id(bytes_or_None) -> bytes` (default: torch.distributed.broadcast_object_list when a process group exists)."""

    def __init__(self, ctx, rank: int, world: int, exchange_id=None):
        import ctypes as C
        from . import _ffi
        self.ctx, self.rank, self.world = ctx, int(rank), int(world)
        self.lib = _ffi.load()
        buf = C.create_string_buffer(128)
        if self.rank == 0:
            _ffi.check(self.lib.pv_comm_unique_id(ctx.handle, buf))
        uid = buf.raw if self.rank == 0 else None
        if exchange_id is not None:
            uid = exchange_id(uid)
        elif self.world > 1:
            box = [uid]
            dist.broadcast_object_list(box, src=0)
            uid = box[0]
        self.handle = C.c_void_p()
        _ffi.check(self.lib.pv_comm_create(ctx.handle, uid, self.rank, self.world, C.byref(self.handle)))

    def gather(self, local: torch.Tensor, dst: int = 0, capacity_rows: Optional[int] = None, stream: int = 0):
        """local: contiguous DEVICE tensor [n_r, ...]; returns (rows [sum n_r, ...], counts) on dst, (None, counts) elsewhere.
        capacity_rows: rows the destination provides room for. None = size the buffer from the counts (pv_gather_counts: one
        more 8-byte all-gather), which is right for ragged ranks; a caller with a bound passes it and gets PV_ERR_CAPACITY
        on every rank when the rows do not fit."""
        import ctypes as C
        from . import _ffi
        assert local.is_cuda and local.is_contiguous()
        n = int(local.shape[0])
        row_bytes = int(np_prod(local.shape[1:]) * local.element_size())
        if capacity_rows is None:
            pre = (C.c_int64 * self.world)()
            _ffi.check(self.lib.pv_gather_counts(self.ctx.handle, self.handle, n, pre, stream or None))
            cap = sum(int(c) for c in pre)
        else:
            cap = int(capacity_rows)
        recv = None
        if self.rank == dst:
            recv = torch.empty((max(cap, 1),) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        counts = (C.c_int64 * self.world)()
        _ffi.check(self.lib.pv_gather(self.ctx.handle, self.handle, local.data_ptr() if n else None, n, row_bytes,
                                      recv.data_ptr() if recv is not None else None, cap, counts, int(dst), stream or None))
        cl = [int(c) for c in counts]
        if self.rank != dst:
            return None, cl
        self.ctx.synchronize() if not stream else torch.cuda.synchronize()
        return recv[: sum(cl)], cl

    def close(self):
        if getattr(self, "handle", None):
            self.lib.pv_comm_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def np_prod(shape) -> int:
    out = 1
    for s in shape:
        out *= int(s)
    return out
