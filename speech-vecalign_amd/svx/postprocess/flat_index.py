"""Exact ("Flat") embedding database in HBM -- what the margin-scoring step searches.

The reference trains and populates a faiss index (svecalign/postprocess/prep_index.py:153-185) and searches
it on the GPU with gpu_type "fp16-shard" (svecalign/postprocess/score_align.py:48-50,197-214).  For corpora
that fit a Flat index faiss keeps the unit-norm rows in fp16 and does a brute-force search; `FlatIndex`
is that database as one fp16 (or bf16) matrix on the device, searched by svx_knn_mean_sim (a GEMM fused
with the per-row top-k, csrc/svx_margin.hip).  The file format is faiss' own IndexFlat serialisation
("IxF2"/"IxFI" + fp32 rows), read and written with `struct` -- no faiss needed, nothing is unpickled.

Sharded corpora: every rank adds the rows of its own alignments.  Either `all_gather_rows` (RCCL all-gather)
assembles the global database on each GPU, or -- `ring_shards` -- the shards travel round the ring of ranks
(point-to-point over xGMI) while every rank merges the top-k of its own queries shard by shard
(svx_knn_topk_merge): a GPU then holds two shards instead of the corpus, and the transfer of the next shard
overlaps the search of the current one.  Row order is irrelevant to a mean over nearest neighbours.
"""
import ctypes
import struct
from typing import Optional

import numpy as np

from .. import _lib

_HEADER = struct.Struct("<4siqqqBi")  # fourcc, d, ntotal, 2 x dummy, is_trained, metric_type
_FOURCC_L2, _FOURCC_IP = b"IxF2", b"IxFI"


def read_faiss_flat(path) -> np.ndarray:
    """rows [ntotal, d] float32 of a faiss IndexFlatL2 / IndexFlatIP file (memory-mapped, read-only)."""
    with open(path, "rb") as f:
        head = f.read(_HEADER.size)
        if len(head) < _HEADER.size:
            raise ValueError(f"{path}: not a faiss index file")
        fourcc, d, ntotal, _, _, _, metric = _HEADER.unpack(head)
        if fourcc not in (_FOURCC_L2, _FOURCC_IP):
            raise NotImplementedError(
                f"{path}: faiss index type {fourcc!r}; only Flat indexes are searched exactly here "
                "(IVF / PQ indexes are approximate structures of the reference's faiss path)")
        off = _HEADER.size
        if metric > 1:  # metric_arg
            off += 4
            f.seek(off)
        (count,) = struct.unpack("<Q", f.read(8))
        off += 8
    if count != ntotal * d:
        raise ValueError(f"{path}: {count} stored values for {ntotal} x {d}")
    if ntotal == 0:
        return np.zeros((0, d), dtype=np.float32)
    return np.memmap(path, dtype="<f4", mode="r", offset=off, shape=(ntotal, d))


def write_faiss_flat(path, rows: np.ndarray) -> None:
    """IndexFlatL2 file (metric L2, as the reference's Flat.populate.idx) with fp32 rows."""
    rows = np.ascontiguousarray(rows, dtype="<f4")
    n, d = rows.shape
    with open(path, "wb") as f:
        f.write(_HEADER.pack(_FOURCC_L2, d, n, 1 << 20, 1 << 20, 1, 1))
        f.write(struct.pack("<Q", n * d))
        f.write(rows.tobytes())


def _torch_dtype_code(t, dtype):
    return {t.float32: _lib.SVX_F32, t.float16: _lib.SVX_F16, t.bfloat16: _lib.SVX_BF16}[dtype]


def to_device_rows(ctx, a):
    """numpy / torch [n, d] -> contiguous device tensor in its own storage type (fp32 / fp16 / bf16)."""
    t = ctx.torch
    if not hasattr(a, "data_ptr"):
        a = np.asarray(a)
        if a.dtype not in (np.float32, np.float16):
            a = a.astype(np.float32)
        a = t.from_numpy(np.array(a))  # (copies: file readers hand out read-only maps)
    if a.dtype not in (t.float32, t.float16, t.bfloat16):
        a = a.float()
    return a.to(ctx.tdev).contiguous()


class FlatIndex:
    """Unit-norm rows [ntotal, d] in fp16 (default) or bf16 on one GPU."""

    def __init__(self, d: int = 1024, storage: str = "fp16", device=None):
        self.ctx = _lib.context(device)
        t = self.ctx.torch
        if storage not in ("fp16", "bf16"):
            raise ValueError(f"storage {storage!r}: the database is kept in fp16 or bf16")
        self.d = int(d)
        self.storage = storage
        self.tdtype = t.float16 if storage == "fp16" else t.bfloat16
        self.code = _lib.SVX_F16 if storage == "fp16" else _lib.SVX_BF16
        self._chunks = []
        self._rows = t.empty((0, self.d), dtype=self.tdtype, device=self.ctx.tdev)

    # -- population (prep_index.py:153-185: normalize_L2 + index.add per embedding file)
    def add(self, embed) -> None:
        ctx = self.ctx
        x = to_device_rows(ctx, embed)
        if x.ndim != 2 or x.shape[1] != self.d:
            raise ValueError(f"expected [n, {self.d}] rows, got {tuple(x.shape)}")
        if x.shape[0] == 0:
            return
        out = ctx.torch.empty(x.shape, dtype=self.tdtype, device=ctx.tdev)
        ctx.check(ctx.lib.svx_unit_rows(ctx.h, ctypes.c_void_p(x.data_ptr()), _torch_dtype_code(ctx.torch, x.dtype),
                                        int(x.shape[0]), self.d, ctypes.c_void_p(out.data_ptr()), self.code))
        self._chunks.append(out)

    def add_unit_rows(self, rows) -> None:
        """Rows that are already unit norm (a populated index file, or another rank's shard)."""
        x = to_device_rows(self.ctx, rows)
        if x.shape[0]:
            self._chunks.append(x.to(self.tdtype))

    @property
    def rows(self):
        if self._chunks:
            self._rows = self.ctx.torch.cat([self._rows] + self._chunks, dim=0).contiguous()
            self._chunks = []
        return self._rows

    @property
    def ntotal(self) -> int:
        return int(self._rows.shape[0] + sum(c.shape[0] for c in self._chunks))

    # -- search (score_align.py:137-148): mean cosine to the k nearest rows
    def mean_sim(self, queries, k: int):
        ctx = self.ctx
        q = to_device_rows(ctx, queries)
        if q.ndim != 2 or q.shape[1] != self.d:
            raise ValueError(f"expected [n, {self.d}] queries, got {tuple(q.shape)}")
        db = self.rows
        out = ctx.torch.empty((q.shape[0],), dtype=ctx.torch.float32, device=ctx.tdev)
        ctx.check(ctx.lib.svx_knn_mean_sim(ctx.h, ctypes.c_void_p(q.data_ptr()), _torch_dtype_code(ctx.torch, q.dtype),
                                           int(q.shape[0]), ctypes.c_void_p(db.data_ptr()), self.code, int(db.shape[0]),
                                           self.d, int(k), ctypes.c_void_p(out.data_ptr())))
        return out

    def merge_topk(self, queries, k: int, topk=None, want_mean: bool = False):
        """One shard of a sharded search: merge this index's rows into the running top-k lists `topk` [n, k]
        (None: start them) -> (topk, mean_sim or None).  The shard may hold fewer than k rows, or none."""
        ctx = self.ctx
        t = ctx.torch
        q = to_device_rows(ctx, queries)
        if q.ndim != 2 or q.shape[1] != self.d:
            raise ValueError(f"expected [n, {self.d}] queries, got {tuple(q.shape)}")
        first = topk is None
        if first:
            topk = t.empty((q.shape[0], int(k)), dtype=t.float32, device=ctx.tdev)
        elif tuple(topk.shape) != (q.shape[0], int(k)) or topk.dtype != t.float32 or not topk.is_contiguous():
            raise ValueError(f"topk must be a contiguous float32 [{q.shape[0]}, {k}] tensor")
        db = self.rows
        mean = t.empty((q.shape[0],), dtype=t.float32, device=ctx.tdev) if want_mean else None
        ctx.check(ctx.lib.svx_knn_topk_merge(ctx.h, ctypes.c_void_p(q.data_ptr()), _torch_dtype_code(t, q.dtype), int(q.shape[0]),
                                             ctypes.c_void_p(db.data_ptr() if db.shape[0] else None), self.code, int(db.shape[0]),
                                             self.d, int(k), ctypes.c_void_p(topk.data_ptr()), int(first),
                                             ctypes.c_void_p(mean.data_ptr()) if want_mean else None))
        return topk, mean

    # -- files
    @classmethod
    def read(cls, path, storage: str = "fp16", device=None) -> "FlatIndex":
        rows = read_faiss_flat(path)
        idx = cls(d=rows.shape[1], storage=storage, device=device)
        idx.add_unit_rows(rows)
        return idx

    def write(self, path) -> None:
        write_faiss_flat(path, self.rows.float().cpu().numpy())


def all_gather_rows(local, group=None):
    """Concatenate every rank's [n_r, d] rows (rank order) on every rank: one size exchange and one padded
    all-gather (RCCL over xGMI on GPUs; gloo in the CPU tests).  Works on any torch tensor."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    n = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    sizes = [int(s.item()) for s in sizes]
    cap = max(sizes)
    if cap == 0:
        return local
    padded = torch.zeros((cap,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    padded[:local.shape[0]] = local
    # (fp16 / bf16 travel as raw bytes: an all-gather is a copy and every backend moves uint8)
    wire = padded.view(torch.uint8) if padded.dtype in (torch.float16, torch.bfloat16) else padded
    parts = [torch.empty_like(wire) for _ in range(world)]
    dist.all_gather(parts, wire.contiguous(), group=group)
    parts = [p.view(local.dtype)[:s] for p, s in zip(parts, sizes)]
    return torch.cat(parts, dim=0)


def ring_shards(local, group=None):
    """Generator over every rank's [n_r, d] rows, own shard first, then the shards of ranks r-1, r-2, ... as they
    arrive round the ring: before shard s is handed out, its forwarding to rank r+1 (and the receive of shard
    s+1 from rank r-1) is already posted, so the caller's work on shard s overlaps the transfer.  Point-to-point
    isend/irecv (RCCL over xGMI on GPUs; gloo in the CPU tests); fp16 / bf16 travel as raw bytes.  Yields
    (owner_rank, rows); a rank holds its own shard and two wire buffers of the largest shard, never the corpus."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        yield (dist.get_rank(group) if dist.is_available() and dist.is_initialized() else 0), local
        return
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    n = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    sizes = [int(v.item()) for v in sizes]
    cap = max(sizes)
    if cap == 0:
        for s in range(world):
            yield (rank - s) % world, local
        return
    nxt = dist.get_global_rank(group, (rank + 1) % world) if group is not None else (rank + 1) % world
    prv = dist.get_global_rank(group, (rank - 1) % world) if group is not None else (rank - 1) % world
    as_wire = (lambda x: x.view(torch.uint8)) if local.dtype in (torch.float16, torch.bfloat16) else (lambda x: x)
    row_shape = tuple(local.shape[1:])
    bufs = [torch.zeros((cap,) + row_shape, dtype=local.dtype, device=local.device) for _ in range(2)]
    bufs[0][:local.shape[0]] = local
    cur = 0
    for s in range(world):
        owner = (rank - s) % world
        reqs = []
        if s + 1 < world:
            ops = [dist.P2POp(dist.isend, as_wire(bufs[cur]), nxt, group), dist.P2POp(dist.irecv, as_wire(bufs[1 - cur]), prv, group)]
            reqs = dist.batch_isend_irecv(ops)
        yield owner, (local if s == 0 else bufs[cur][:sizes[owner]])
        for r in reqs:
            r.wait()
        cur = 1 - cur
