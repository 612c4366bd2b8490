"""Row-range partition of the Neumann iteration across the GPUs of one node.

One process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI, "gloo" in the CPU tests).
Rank p owns the contiguous rows [lo_p, hi_p) of A (column ids stay global) and the matching
slices of dinv / x / t.  Per iteration there is exactly one exchange step — the freshly computed
slice of the term vector t has to reach every rank that gathers from it — plus an 8-byte
all-reduce of the squared term norm:

  * AllGather     uniform column structure: every rank needs all of t  ->  all_gather_into_tensor
  * Halo          banded structure (|i - j| <= w): only the w entries either side of a slice
                  boundary are exchanged, point to point with the two neighbours.

Overlap (banded case): the rows within w of a slice boundary are two small matrices of their own, launched on a
side stream concurrently with the interior launch (all other rows); their halo transfer is enqueued behind them
on RCCL's stream; the step only waits for the transfer at its end.  The norm all-reduce of a step is asynchronous too
and overlaps the next step.

Precedent for the partition itself: simd_ops::parallel_matrix_vector_multiply row chunks
(src/simd_ops.rs:201-239).  The exchange is a copy, so every rank sees bit-identical t and the
partitioned iteration reproduces the single-GPU one bit for bit.

The local step is a callable so that the CPU tests can drive the same host logic with a stand-in;
the product wiring (`hip_local_step`, `hip_local_ops`, `HipSplitStep`) calls libsublinear_hip on device memory.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def _staged(t: torch.Tensor, group=None) -> bool:
    """gloo moves host memory only: device tensors are staged through the host (test mode — two ranks sharing one
    GPU exercise the partitioned driver with the real kernels; the production backend is "nccl" = RCCL)."""
    return t.is_cuda and dist.get_backend(group) == "gloo"


def all_reduce_scalar(t: torch.Tensor, op, group=None, async_op: bool = False):
    if _staged(t, group):
        h = t.cpu()
        dist.all_reduce(h, op=op, group=group)
        t.copy_(h)
        return None
    return dist.all_reduce(t, op=op, group=group, async_op=async_op)


@dataclass
class RowPartition:
    """Contiguous row ranges, one per rank.  Default: equal row counts (the last ranks may be short or empty; vectors are
    padded to world * rows_per_rank so the all-gather moves equal chunks).  With `bounds` (world + 1 ascending row
    indices, e.g. from nnz_balanced_bounds) the ranges are arbitrary and vectors have exactly n_global entries."""
    n_global: int
    world: int
    rank: int
    bounds: Optional[Sequence[int]] = None

    def __post_init__(self):
        if self.bounds is not None:
            b = [int(v) for v in self.bounds]
            if len(b) != self.world + 1 or b[0] != 0 or b[-1] != self.n_global or any(b[i] > b[i + 1] for i in range(self.world)):
                raise ValueError("bounds must be world + 1 ascending row indices from 0 to n_global")
            self.bounds = b
            self.rows_per_rank = max(b[i + 1] - b[i] for i in range(self.world))
            self.n_padded = self.n_global
            self.lo, self.hi = b[self.rank], b[self.rank + 1]
            return
        self.rows_per_rank = -(-self.n_global // self.world)          # ceil
        self.n_padded = self.rows_per_rank * self.world
        self.lo = min(self.rank * self.rows_per_rank, self.n_global)
        self.hi = min(self.lo + self.rows_per_rank, self.n_global)

    @property
    def n_local(self) -> int:
        return self.hi - self.lo

    @property
    def uniform(self) -> bool:
        return self.bounds is None

    def range_of(self, r: int):
        if self.bounds is not None:
            return self.bounds[r], self.bounds[r + 1]
        lo = min(r * self.rows_per_rank, self.n_global)
        return lo, min(lo + self.rows_per_rank, self.n_global)


def nnz_balanced_bounds(row_ptr, world: int):
    """Row bounds that give every rank about nnz / world stored entries (prefix sum of row_ptr): rank r starts at
    the first row whose prefix reaches r * nnz / world.  Every rank derives the same bounds from the same global row_ptr."""
    import numpy as np
    rp = np.asarray(row_ptr, dtype=np.int64)
    n, nnz = len(rp) - 1, int(rp[-1])
    cuts = [int(np.searchsorted(rp, (r * nnz) // world, side="left")) for r in range(1, world)]
    b = [0] + [min(max(c, 0), n) for c in cuts] + [n]
    for i in range(1, len(b)):
        b[i] = max(b[i], b[i - 1])
    return b


class _WorkList:
    """several outstanding collectives waited for as one (plus host buffers to land in the staged test mode)"""

    def __init__(self, works, landing=()):
        self.works, self.landing = works, landing

    def wait(self):
        for w in self.works:
            w.wait()
        for buf, view in self.landing:
            view.copy_(buf)


class AllGatherExchange:
    """Every rank contributes its slice; result is the full (padded) vector on every rank."""
    name = "allgather"

    def __init__(self, part: RowPartition, group=None):
        self.part, self.group = part, group

    def start(self, t_full: torch.Tensor):
        p = self.part
        if p.world == 1:
            return None
        if not p.uniform:                       # unequal ranges: one broadcast per rank (all_gather needs equal chunks on gloo)
            staged = _staged(t_full, self.group)
            works, landing = [], []
            for r in range(p.world):
                lo, hi = p.range_of(r)
                if hi == lo:
                    continue
                src = dist.get_global_rank(self.group, r) if self.group is not None else r
                view = t_full[lo:hi]
                if staged:
                    buf = view.cpu() if r == p.rank else torch.empty(hi - lo, dtype=t_full.dtype)
                    works.append(dist.broadcast(buf, src=src, group=self.group, async_op=True))
                    if r != p.rank:
                        landing.append((buf, view))
                else:
                    works.append(dist.broadcast(view, src=src, group=self.group, async_op=True))
            return _WorkList(works, landing)
        mine = t_full[p.rank * p.rows_per_rank:(p.rank + 1) * p.rows_per_rank]
        if _staged(t_full, self.group):
            host = torch.empty(t_full.numel(), dtype=t_full.dtype)
            dist.all_gather_into_tensor(host, mine.cpu(), group=self.group)
            t_full.copy_(host)
            return None
        return dist.all_gather_into_tensor(t_full, mine, group=self.group, async_op=True)

    def finish(self, handle) -> None:
        if handle is not None:
            handle.wait()

    def __call__(self, t_full: torch.Tensor) -> None:
        self.finish(self.start(t_full))

    def bytes_sent_per_step(self) -> int:
        return 8 * (self.part.n_local if not self.part.uniform else self.part.rows_per_rank) * (self.part.world - 1)


def _check_halo_partition(part: "RowPartition", half_bandwidth: int) -> None:
    """A neighbour halo pairs rank r with r - 1 / r + 1: every rank must own at least w rows.  A rank with NO rows would leave
    its neighbours' sends unmatched (hang) or ship stale strips (explicit bounds with an empty middle rank), so such
    partitions are rejected — e.g. the uniform partition of n = 9 over 4 ranks, whose last rank is [9, 9)."""
    if part.world <= 1:
        return
    sizes = [part.range_of(r)[1] - part.range_of(r)[0] for r in range(part.world)]
    if min(sizes) <= 0:
        raise ValueError(f"halo exchange needs rows on every rank; row counts per rank: {sizes}")
    if min(sizes) < half_bandwidth:
        raise ValueError("halo exchange needs at least w rows on every rank")


class HaloExchange:
    """Banded systems: only w entries on each side of every slice boundary travel (neighbours only)."""
    name = "halo"
    needs_only_boundary = True      # may start as soon as the boundary rows of the new term are written

    def __init__(self, part: RowPartition, half_bandwidth: int, group=None, loopback: bool = False):
        """loopback (MEASUREMENT mode, world size 1 with an initialised RCCL group): the rank sends both boundary strips to
        ITSELF into scratch strips — same op list, same enqueue cost and same transfer kernels as a rank with two neighbours,
        so the per-step host and device cost of the exchange can be measured on a one-GPU box."""
        _check_halo_partition(part, half_bandwidth)
        self.part, self.w, self.group = part, int(half_bandwidth), group
        self.loopback = bool(loopback) and part.world == 1
        self._ops = {}
        self._scratch = None

    def start(self, t_full: torch.Tensor):
        p, w = self.part, self.w
        if self.loopback and w > 0:
            ops = self._ops.get(t_full.data_ptr())
            if ops is None:
                if self._scratch is None:
                    self._scratch = torch.empty(2, w, dtype=t_full.dtype, device=t_full.device)
                ops = [dist.P2POp(dist.isend, t_full[p.lo:p.lo + w], p.rank, group=self.group),
                       dist.P2POp(dist.irecv, self._scratch[0], p.rank, group=self.group),
                       dist.P2POp(dist.isend, t_full[p.hi - w:p.hi], p.rank, group=self.group),
                       dist.P2POp(dist.irecv, self._scratch[1], p.rank, group=self.group)]
                self._ops[t_full.data_ptr()] = ops
            return dist.batch_isend_irecv(ops), ()
        if p.world == 1 or w == 0:
            return None
        staged = _staged(t_full, self.group)
        if not staged:                        # device tensors over RCCL: the op list of a buffer is built once (two ping-pong buffers)
            key = t_full.data_ptr()
            ops = self._ops.get(key)
            if ops is None:
                ops = []
                if p.rank > 0:                # left neighbour: send my first w, receive its last w
                    ops.append(dist.P2POp(dist.isend, t_full[p.lo:p.lo + w], p.rank - 1, group=self.group))
                    ops.append(dist.P2POp(dist.irecv, t_full[p.lo - w:p.lo], p.rank - 1, group=self.group))
                if p.rank < p.world - 1 and p.hi < p.n_global:
                    ops.append(dist.P2POp(dist.isend, t_full[p.hi - w:p.hi], p.rank + 1, group=self.group))
                    ops.append(dist.P2POp(dist.irecv, t_full[p.hi:p.hi + w], p.rank + 1, group=self.group))
                self._ops[key] = ops
            return (dist.batch_isend_irecv(ops), ()) if ops else None
        ops, landing = [], []                 # gloo test mode: staged through the host

        def pair(send_view, recv_view, peer):
            buf = torch.empty(recv_view.numel(), dtype=recv_view.dtype)
            landing.append((buf, recv_view))
            ops.append(dist.P2POp(dist.isend, send_view.cpu(), peer, group=self.group))
            ops.append(dist.P2POp(dist.irecv, buf, peer, group=self.group))

        if p.rank > 0:
            pair(t_full[p.lo:p.lo + w], t_full[p.lo - w:p.lo], p.rank - 1)
        if p.rank < p.world - 1 and p.hi < p.n_global:
            pair(t_full[p.hi - w:p.hi], t_full[p.hi:p.hi + w], p.rank + 1)
        if not ops:
            return None
        return dist.batch_isend_irecv(ops), landing

    def finish(self, handle) -> None:
        if handle:
            reqs, landing = handle
            for r in reqs:
                r.wait()
            for buf, view in landing:
                view.copy_(buf)

    def __call__(self, t_full: torch.Tensor) -> None:
        self.finish(self.start(t_full))

    def bytes_sent_per_step(self) -> int:
        inner = (1 if self.part.rank > 0 else 0) + (1 if self.part.rank < self.part.world - 1 else 0)
        return 8 * self.w * inner


class HaloAllReduceExchange:
    """The same halo as HaloExchange, moved the way BASELINE's north_star words it: ONE all-reduce (sum) over a compact, zero-filled
    buffer of all boundary strips — boundary k (between ranks k and k + 1) owns two slots of w entries, [last w of rank k | first
    w of rank k + 1]; every slot is written by exactly one rank and zero everywhere else, so the sum is a copy (x + 0 = x; only a
    -0.0 would come back as +0.0).  Every rank receives every strip (2 w (P - 1) entries): more bytes and more glue launches than
    the neighbour sends, which stay the default; selectable for comparison (bench.py --exchange allreduce)."""
    name = "halo_allreduce"
    needs_only_boundary = True

    def __init__(self, part: RowPartition, half_bandwidth: int, group=None):
        _check_halo_partition(part, half_bandwidth)
        self.part, self.w, self.group = part, int(half_bandwidth), group
        self.compact = None

    def start(self, t_full: torch.Tensor):
        p, w = self.part, self.w
        if p.world == 1 or w == 0:
            return None
        if self.compact is None or self.compact.device != t_full.device:
            self.compact = torch.zeros(p.world - 1, 2, w, dtype=t_full.dtype, device=t_full.device)
        c = self.compact
        c.zero_()
        if p.rank > 0 and p.hi > p.lo:
            c[p.rank - 1, 1].copy_(t_full[p.lo:p.lo + w])
        if p.rank < p.world - 1 and p.hi < p.n_global:
            c[p.rank, 0].copy_(t_full[p.hi - w:p.hi])
        return all_reduce_scalar(c, dist.ReduceOp.SUM, self.group, async_op=True), t_full

    def finish(self, handle) -> None:
        if not handle:
            return
        work, t_full = handle
        if work is not None:
            work.wait()
        p, w, c = self.part, self.w, self.compact
        if p.rank > 0 and p.hi > p.lo:
            t_full[p.lo - w:p.lo].copy_(c[p.rank - 1, 0])
        if p.rank < p.world - 1 and p.hi < p.n_global:
            t_full[p.hi:p.hi + w].copy_(c[p.rank, 1])

    def __call__(self, t_full: torch.Tensor) -> None:
        self.finish(self.start(t_full))

    def bytes_sent_per_step(self) -> int:
        return 8 * 2 * self.w * max(self.part.world - 1, 0)


# local_step(t_in_full, t_out_local, x_local, norm2_out) -> None : one fused Neumann step on the local rows
LocalStep = Callable[[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor], None]


class SplitStep:
    """A local step cut into row pieces [(lo, hi, step)], local row coordinates: `boundary` pieces are launched
    first, then the exchange starts, then the `interior` pieces run while it is in flight."""

    def __init__(self, boundary: Sequence[Tuple[int, int, LocalStep]], interior: Sequence[Tuple[int, int, LocalStep]], device):
        self.boundary, self.interior = list(boundary), list(interior)
        self.parts = torch.zeros(len(self.boundary) + len(self.interior), 2, dtype=torch.float64, device=device)

    def _run(self, pieces, first, t_in, t_out_local, x_local):
        for k, (lo, hi, step) in enumerate(pieces):
            if hi > lo:
                step(t_in, t_out_local[lo:hi], x_local[lo:hi], self.parts[first + k])

    def run_boundary(self, t_in, t_out_local, x_local):
        self.parts.zero_()
        self._run(self.boundary, 0, t_in, t_out_local, x_local)

    def run_interior(self, t_in, t_out_local, x_local, norm2_out):
        self._run(self.interior, len(self.boundary), t_in, t_out_local, x_local)
        torch.sum(self.parts[:, 0], dim=0, out=norm2_out[0])          # fixed order: pieces in list order


class HipSplitStep:
    """SplitStep on libsublinear_hip: every piece is its own row-slice matrix; the pieces leave their per-block partial sums
    side by side in one device buffer (sl_neumann_step_partials) and ONE fixed-order reduction closes the step
    (sl_reduce_partials) — boundary pieces, interior piece, one reduce: four launches per step, no glue kernels."""

    def __init__(self, boundary: Sequence[Tuple[int, int, int]], interior: Sequence[Tuple[int, int, int]], dinv_local: torch.Tensor, order: int = 0):
        """pieces: (lo, hi, matrix handle) in local row coordinates"""
        import ctypes as C

        from . import _lib as L
        self._L, self._lib, self._C = L, L.load(), C
        self.boundary = [(lo, hi, h) for lo, hi, h in boundary if hi > lo]
        self.interior = [(lo, hi, h) for lo, hi, h in interior if hi > lo]
        self.dinv, self.order = dinv_local, order
        cap = 0
        for _, _, h in self.boundary + self.interior:
            c = L.u64(0)
            L.check(self._lib.sl_matrix_partials_capacity(h, C.byref(c)))
            cap += int(c.value)
        self.partials = torch.zeros(max(cap, 1), dtype=torch.float64, device=dinv_local.device)
        self._used = 0
        self.concurrent = True                       # boundary pieces on a side stream (run_overlapped)

    def _run(self, pieces, t_in, t_out_local, x_local):
        L, C = self._L, self._C
        d0, t0, o0, x0, p0 = self.dinv.data_ptr(), t_in.data_ptr(), t_out_local.data_ptr(), x_local.data_ptr(), self.partials.data_ptr()
        for lo, hi, h in pieces:                     # raw addresses: no tensor views on the per-step path
            got = C.c_uint32(0)
            L.check(self._lib.sl_neumann_step_partials(h, d0 + 8 * lo, t0, o0 + 8 * lo, x0 + 8 * lo, p0 + 8 * self._used, C.byref(got), self.order))
            self._used += got.value

    def _use_stream(self, stream):
        self._L.check(self._lib.sl_set_stream(self._C.c_void_p(stream.cuda_stream)))

    def run_boundary(self, t_in, t_out_local, x_local):
        self._used = 0
        self._run(self.boundary, t_in, t_out_local, x_local)

    def run_interior(self, t_in, t_out_local, x_local, norm2_out):
        self._run(self.interior, t_in, t_out_local, x_local)
        self._L.check(self._lib.sl_reduce_partials(self.partials.data_ptr(), self._used, norm2_out.data_ptr()))

    def run_overlapped(self, t_in, t_out, t_out_local, x_local, norm2_out, exchange):
        """One step with the boundary pieces on a side stream: they run CONCURRENTLY with the interior piece (both only
        read t_in), the halo transfer is enqueued behind them, and the main stream joins both before the reduction.
        Critical path of a step = the interior kernel; the small launches and the transfer hide under it."""
        main = torch.cuda.current_stream(self.dinv.device)
        if self.side is None:
            self.side = torch.cuda.Stream(device=self.dinv.device, priority=-1)
        side = self.side
        side.wait_stream(main)                       # t_in complete: the previous step's interior rows are in place
        with torch.cuda.stream(side):
            self._use_stream(side)
            if self._halo_pending is not None:       # the strips received during the previous step: only boundary rows read them
                exchange.finish(self._halo_pending)
                self._halo_pending = None
            self.run_boundary(t_in, t_out_local, x_local)
            handle = exchange.start(t_out)           # rides behind the boundary kernels
        self._use_stream(main)
        self._run(self.interior, t_in, t_out_local, x_local)
        main.wait_stream(side)                       # boundary rows and their partial sums are final
        self._L.check(self._lib.sl_reduce_partials(self.partials.data_ptr(), self._used, norm2_out.data_ptr()))
        if self.halo_wait_on_side:
            self._halo_pending = handle              # interior rows never read the strips: the main stream does not wait for them
        else:
            exchange.finish(handle)

    def flush(self, exchange) -> None:
        """current stream waits for the strips still in flight (call before anything but the next step reads the term vector)"""
        if self._halo_pending is not None:
            exchange.finish(self._halo_pending)
            self._halo_pending = None

    halo_wait_on_side = True
    _halo_pending = None
    side = None


class PartitionedNeumann:
    """Ping-pong driver of the partitioned iteration: local fused step -> exchange -> norm all-reduce.

    Every step leaves its local ||t||^2 in the next row of a small device log; the log is all-reduced asynchronously
    (it overlaps the following kernels) once per `reduce_every` steps — 1 = one 8-byte all-reduce per step, B > 1 = one
    all-reduce of B rows per B steps, the batch of the speculative solve loop (the convergence replay
    lags the device by a batch anyway).  A pending all-reduce is only waited for when a norm is read (term_norm) or
    when its log is about to be reused two batches later.  With a SplitStep the term exchange overlaps the interior rows."""

    def __init__(self, part: RowPartition, local_step, exchange, t0_full: torch.Tensor,
                 x_local: torch.Tensor, group=None, reduce_every: int = 1):
        self.part, self.local_step, self.exchange, self.group = part, local_step, exchange, group
        self.t = [t0_full, torch.zeros_like(t0_full)]
        self.x = x_local
        self.cur = 0
        self.reduce_every = max(1, int(reduce_every))
        self._norm = [torch.zeros(self.reduce_every, 2, dtype=torch.float64, device=t0_full.device) for _ in range(2)]
        self._pending = [None, None]
        self._log, self._fill = 0, 0                   # log being filled, rows written
        self._last = (0, 0)                            # (log, row) of the most recent step
        self.steps_done = 0
        self.reduce_always = False                     # measurement mode: issue the norm all-reduce even at world size 1

    def restart(self, t0_full: torch.Tensor, x_local: torch.Tensor) -> None:
        """back to the start of the series (current term = t0, solution = x_local, no step taken): pending all-reduces are waited
        for, the logs start over.  bench.py's parity gate runs two steps, reads them, and restarts."""
        for k in (0, 1):
            if self._pending[k] is not None:
                self._pending[k].wait()
                self._pending[k] = None
        self.t[0].copy_(t0_full)
        self.t[1].zero_()
        self.x.copy_(x_local)
        self.cur = 0
        self._log, self._fill, self._last, self.steps_done = 0, 0, (0, 0), 0

    @property
    def norm2(self) -> torch.Tensor:
        """log row of the most recent step (globally summed after term_norm())"""
        return self._norm[self._last[0]][self._last[1]]

    def _close_log(self, reduce_norm: bool) -> None:
        if self._fill == 0:
            return
        if reduce_norm and (self.part.world > 1 or self.reduce_always):
            self._pending[self._log] = all_reduce_scalar(self._norm[self._log][:self._fill], dist.ReduceOp.SUM, self.group, async_op=True)
        self._log, self._fill = 1 - self._log, 0

    def step(self, reduce_norm: bool = True) -> None:
        p = self.part
        if self._fill == 0 and self._pending[self._log] is not None:   # the all-reduce that last used this log must be done
            self._pending[self._log].wait()
            self._pending[self._log] = None
        norm_out = self._norm[self._log][self._fill]
        t_in, t_out = self.t[self.cur], self.t[1 - self.cur]
        t_out_local = t_out[p.lo:p.hi]
        if isinstance(self.local_step, HipSplitStep) and t_in.is_cuda and getattr(self.exchange, "needs_only_boundary", False) \
                and self.local_step.concurrent:
            self.local_step.run_overlapped(t_in, t_out, t_out_local, self.x, norm_out, self.exchange)
        elif isinstance(self.local_step, (SplitStep, HipSplitStep)):
            self.local_step.run_boundary(t_in, t_out_local, self.x)
            early = getattr(self.exchange, "needs_only_boundary", False)
            handle = self.exchange.start(t_out) if early else None   # boundary rows are final: ship them while the interior computes
            self.local_step.run_interior(t_in, t_out_local, self.x, norm_out)
            if not early:
                handle = self.exchange.start(t_out)                  # an all-gather needs the whole slice
            self.exchange.finish(handle)
        else:
            self.local_step(t_in, t_out_local, self.x, norm_out)
            self.exchange(t_out)
        self._last = (self._log, self._fill)
        self._fill += 1
        if self._fill == self.reduce_every:
            self._close_log(reduce_norm)
        self.cur = 1 - self.cur
        self.steps_done += 1

    def flush(self) -> None:
        """strips of the last step that are still in flight (HipSplitStep.run_overlapped) land before the caller reads the term"""
        if isinstance(self.local_step, HipSplitStep):
            self.local_step.flush(self.exchange)

    @property
    def term(self) -> torch.Tensor:
        self.flush()
        return self.t[self.cur]

    def term_norm(self) -> float:
        """global ||t|| of the most recent step (closes a partly filled log, waits for its all-reduce)"""
        self.flush()
        self._close_log(True)
        log, row = self._last
        if self._pending[log] is not None:
            self._pending[log].wait()
            self._pending[log] = None
        return float(self._norm[log][row][0].item()) ** 0.5

    def term_norms_of_batch(self):
        """global ||t|| of every step of the batch the most recent step belongs to, oldest first (what the convergence replay reads)"""
        self.term_norm()
        log, row = self._last
        return [float(v) ** 0.5 for v in self._norm[log][:row + 1, 0].tolist()]


def hip_local_step(matrix_handle: int, dinv_local: torch.Tensor, order: int = 0) -> LocalStep:
    """Product wiring: the local step is sl_neumann_step on this rank's row slice (device pointers)."""
    from . import _lib as L
    lib = L.load()

    def step(t_in_full, t_out_local, x_local, norm2):
        L.check(lib.sl_neumann_step(matrix_handle, dinv_local.data_ptr(), t_in_full.data_ptr(), t_out_local.data_ptr(),
                                    x_local.data_ptr(), norm2.data_ptr(), order))

    return step


def split_bounds(n_local: int, w: int, has_left: bool, has_right: bool) -> Tuple[List[Tuple[int, int]], List[Tuple[int, int]]]:
    """local row ranges: boundary pieces (what a neighbour's halo needs) and the interior"""
    top = (0, min(w, n_local)) if has_left else (0, 0)
    bot = (max(n_local - w, top[1]), n_local) if has_right else (n_local, n_local)
    return [top, bot], [(top[1], bot[0])]


# ------------------------------------------------------------------------------------------------------
# Partitioned NeumannSolver::solve: the control flow of neumann.rs:469-555 with the vectors split by rows
# ------------------------------------------------------------------------------------------------------
@dataclass
class LocalOps:
    """what one rank must be able to do on its row slice; `hip_local_ops` wires these to the C ABI, the CPU
    tests provide stand-ins.  All tensors live on the rank's device."""
    step: LocalStep                                                    # fused a8 + a9 on the local rows
    residual_norm2: Callable[[torch.Tensor, torch.Tensor, torch.Tensor], None]   # (x_full, rhs_local, norm2_out): a10
    axpy: Callable[[float, torch.Tensor, torch.Tensor], None]           # y += alpha * x (local vectors)
    sumsq: Callable[[torch.Tensor, torch.Tensor], None]                 # (v_local, out[0]) = sum v_i^2


@dataclass
class PartitionedResult:
    solution_local: torch.Tensor
    residual_norm: float
    iterations: int
    converged: bool
    terms_computed: int
    term_norms: List[float]


class PartitionedNeumannSolver:
    """NeumannSolver::solve (src/solver/neumann.rs:469-555) over a row partition: same loop, same stop rules;
    every norm is the all-reduced sum of the ranks' partial sums, every gathered vector (the term each iteration,
    the solution every 5th iteration for update_residual) goes through the exchange.  Per-row values are the same
    bits as on one GPU; norms agree to rounding (partial sums are combined in rank order)."""

    def __init__(self, part: RowPartition, ops: LocalOps, exchange, max_terms: int = 50, series_tolerance: float = 1e-8,
                 reference_scaled_residual: bool = False, group=None):
        self.part, self.ops, self.exchange, self.group = part, ops, exchange, group
        self.max_terms, self.series_tolerance = max_terms, series_tolerance
        self.scaled = reference_scaled_residual

    def _allsum(self, scalar: torch.Tensor) -> float:
        if self.part.world > 1:
            all_reduce_scalar(scalar, dist.ReduceOp.SUM, self.group)
        return float(scalar[0].item())

    def solve(self, b_local: torch.Tensor, dinv_local: torch.Tensor, tolerance: float = 1e-6, max_iterations: int = 1000,
              initial_guess_local: Optional[torch.Tensor] = None, reference_default_start: bool = False) -> PartitionedResult:
        p = self.part
        dev, dt = b_local.device, torch.float64
        rhs = b_local * dinv_local                                       # neumann.rs:191-194
        t = [torch.zeros(p.n_padded, dtype=dt, device=dev), torch.zeros(p.n_padded, dtype=dt, device=dev)]
        t[0][p.lo:p.hi] = rhs                                            # current_term = rhs (:211)
        self.exchange(t[0])
        if initial_guess_local is not None:
            x = initial_guess_local.clone()
        elif reference_default_start:
            x = rhs.clone()                                              # :197-208
        else:
            x = torch.zeros_like(rhs)
        x_full = torch.zeros(p.n_padded, dtype=dt, device=dev)
        res_rhs = rhs if self.scaled else b_local
        scal = torch.zeros(2, dtype=dt, device=dev)
        cur, terms, it = 0, 0, 0
        resn, series_conv = float("inf"), False
        norms: List[float] = []

        def is_converged():
            return resn <= tolerance or (series_conv and not terms >= self.max_terms)

        def update_residual():                                           # :302-318
            x_full[p.lo:p.hi] = x
            self.exchange(x_full)
            self.ops.residual_norm2(x_full, res_rhs, scal)
            return self._allsum(scal[:1]) ** 0.5

        while not is_converged() and it < max_iterations:
            if terms < self.max_terms:                                   # compute_next_term :252-277
                if terms > 0:
                    self.ops.step(t[cur], t[1 - cur][p.lo:p.hi], x, scal)
                    self.exchange(t[1 - cur])
                    cur = 1 - cur
                else:
                    self.ops.axpy(1.0, t[cur][p.lo:p.hi], x)             # x += term (k = 0)
                    self.ops.sumsq(t[cur][p.lo:p.hi], scal)
                tn = self._allsum(scal[:1]) ** 0.5
                norms.append(tn)
                terms += 1
                if tn < self.series_tolerance:
                    series_conv = True
            if it % 5 == 0:
                resn = update_residual()
            it += 1
            if resn != resn or resn in (float("inf"), float("-inf")):
                raise FloatingPointError(f"Non-finite residual norm at iteration {it}")     # NumericalInstability :501-507
            if series_conv:
                break
        resn = update_residual()                                         # :516
        return PartitionedResult(x, resn, it, is_converged(), terms, norms)


def hip_local_ops(matrix_handle: int, dinv_local: torch.Tensor, order: int = 0) -> LocalOps:
    """Product wiring of LocalOps: every operation is a launch of libsublinear_hip on device pointers."""
    import ctypes as C

    from . import _lib as L
    lib = L.load()
    if dinv_local.is_cuda:       # torch's copies / RCCL calls and the library's launches must share one stream
        L.check(lib.sl_set_device(dinv_local.device.index))
        L.check(lib.sl_set_stream(C.c_void_p(torch.cuda.current_stream(dinv_local.device).cuda_stream)))
    step = hip_local_step(matrix_handle, dinv_local, order)

    def residual_norm2(x_full, rhs_local, norm2):
        L.check(lib.sl_residual_norm2(matrix_handle, x_full.data_ptr(), rhs_local.data_ptr(), None, norm2.data_ptr(), order))

    def axpy(alpha, xv, yv):
        L.check(lib.sl_axpy(xv.numel(), alpha, xv.data_ptr(), yv.data_ptr(), L.SL_MEM_DEVICE))

    def sumsq(v, out):
        h = C.c_double(0)
        L.check(lib.sl_dot(v.numel(), v.data_ptr(), v.data_ptr(), C.byref(h), L.SL_MEM_DEVICE))
        out[0] = h.value

    return LocalOps(step, residual_norm2, axpy, sumsq)


class AbiPartitionedNeumannSolver:
    """The partitioned solve through the library's OWN multi-GPU path (C ABI: sl_comm_*, sl_neumann_state_create_partitioned, _run,
    _solution) — this class only passes pointers; PartitionedNeumannSolver above is the same loop over torch.distributed / RCCL.
    One instance per rank; every rank of the job passes the same `name`."""

    def __init__(self, rank: int, world: int, name: str):
        from .solver import Communicator
        self.comm = Communicator(rank, world, name)

    def solve(self, matrix_handle, b_local: torch.Tensor, tolerance: float = 1e-6, max_iterations: int = 1000, max_terms: int = 50,
              series_tolerance: float = 1e-8, order: int = 0) -> dict:
        """matrix_handle: this rank's row slice (sl_matrix_create_csr with row_offset = first row, n_cols = n_global);
        b_local: its rows of the right-hand side (device tensor).  Collective."""
        import ctypes as C

        from . import _lib as L
        lib = L.load()
        o = L.NeumannOptions()
        lib.sl_neumann_options_default(C.byref(o))
        o.tolerance, o.max_iterations, o.max_terms, o.series_tolerance = tolerance, max_iterations, max_terms, series_tolerance
        o.order, o.mem = order, L.SL_MEM_DEVICE
        st = C.c_void_p()
        L.check(lib.sl_neumann_state_create_partitioned(self.comm._h, matrix_handle, b_local.data_ptr(), None, C.byref(o), C.byref(st)))
        try:
            res = L.NeumannResult()
            status = lib.sl_neumann_state_run(st, None, C.byref(res))
            x = torch.empty_like(b_local)
            L.check(lib.sl_neumann_state_solution(st, x.data_ptr(), L.SL_MEM_DEVICE))
            if status not in (0, 3):
                L.check(status)
            return {"solution_local": x, "iterations": int(res.iterations), "terms": int(res.terms_computed), "converged": bool(res.converged),
                    "residual_norm": res.residual_norm, "last_term_norm": res.last_term_norm, "device_time_ms": res.device_time_ms}
        finally:
            lib.sl_neumann_state_destroy(st)

    def close(self) -> None:
        self.comm.close()
