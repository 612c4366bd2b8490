"""Row-range partition of the Neumann iteration across the GPUs of one node.

One process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI, "gloo" in the CPU tests).
Rank p owns the contiguous rows [lo_p, hi_p) of A (column ids stay global) and the matching
slices of dinv / x / t.  Per iteration there is exactly one exchange step — the freshly computed
slice of the term vector t has to reach every rank that gathers from it — plus an 8-byte
all-reduce of the squared term norm:

  * AllGather     uniform column structure: every rank needs all of t  ->  all_gather_into_tensor
  * Halo          banded structure (|i - j| <= w): only the w entries either side of a slice
                  boundary are exchanged, point to point with the two neighbours.

Overlap (banded case): the rows within w of a slice boundary are computed FIRST by two small launches;
their halo transfer then runs on RCCL's stream while the interior launch (all other rows) computes;
the step only waits for the transfer at its end.  The norm all-reduce of a step is asynchronous too
and overlaps the next step.

Precedent for the partition itself: simd_ops::parallel_matrix_vector_multiply row chunks
(src/simd_ops.rs:201-239).  The exchange is a copy, so every rank sees bit-identical t and the
partitioned iteration reproduces the single-GPU one bit for bit.

The local step is a callable so that the CPU tests can drive the same host logic with a stand-in;
the product wiring (`hip_local_step`, `hip_split_step`) calls sl_neumann_step on device memory.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


@dataclass
class RowPartition:
    n_global: int
    world: int
    rank: int

    def __post_init__(self):
        self.rows_per_rank = -(-self.n_global // self.world)          # ceil
        self.n_padded = self.rows_per_rank * self.world
        self.lo = min(self.rank * self.rows_per_rank, self.n_global)
        self.hi = min(self.lo + self.rows_per_rank, self.n_global)

    @property
    def n_local(self) -> int:
        return self.hi - self.lo

    def bounds(self, r: int):
        lo = min(r * self.rows_per_rank, self.n_global)
        return lo, min(lo + self.rows_per_rank, self.n_global)


class AllGatherExchange:
    """Every rank contributes its slice; result is the full (padded) vector on every rank."""
    name = "allgather"

    def __init__(self, part: RowPartition, group=None):
        self.part, self.group = part, group

    def start(self, t_full: torch.Tensor):
        p = self.part
        if p.world == 1:
            return None
        mine = t_full[p.rank * p.rows_per_rank:(p.rank + 1) * p.rows_per_rank]
        return dist.all_gather_into_tensor(t_full, mine, group=self.group, async_op=True)

    def finish(self, handle) -> None:
        if handle is not None:
            handle.wait()

    def __call__(self, t_full: torch.Tensor) -> None:
        self.finish(self.start(t_full))

    def bytes_sent_per_step(self) -> int:
        return 8 * self.part.rows_per_rank * (self.part.world - 1)


class HaloExchange:
    """Banded systems: only w entries on each side of every slice boundary travel (neighbours only)."""
    name = "halo"
    needs_only_boundary = True      # may start as soon as the boundary rows of the new term are written

    def __init__(self, part: RowPartition, half_bandwidth: int, group=None):
        if part.world > 1 and part.rows_per_rank < half_bandwidth:
            raise ValueError("halo exchange needs rows_per_rank >= w")
        self.part, self.w, self.group = part, int(half_bandwidth), group

    def start(self, t_full: torch.Tensor):
        p, w = self.part, self.w
        if p.world == 1 or w == 0:
            return None
        ops = []
        if p.rank > 0:                       # left neighbour: send my first w, receive its last w
            ops.append(dist.P2POp(dist.isend, t_full[p.lo:p.lo + w], p.rank - 1, group=self.group))
            ops.append(dist.P2POp(dist.irecv, t_full[p.lo - w:p.lo], p.rank - 1, group=self.group))
        if p.rank < p.world - 1 and p.hi < p.n_global:
            ops.append(dist.P2POp(dist.isend, t_full[p.hi - w:p.hi], p.rank + 1, group=self.group))
            ops.append(dist.P2POp(dist.irecv, t_full[p.hi:p.hi + w], p.rank + 1, group=self.group))
        return dist.batch_isend_irecv(ops) if ops else None

    def finish(self, handle) -> None:
        if handle:
            for r in handle:
                r.wait()

    def __call__(self, t_full: torch.Tensor) -> None:
        self.finish(self.start(t_full))

    def bytes_sent_per_step(self) -> int:
        inner = (1 if self.part.rank > 0 else 0) + (1 if self.part.rank < self.part.world - 1 else 0)
        return 8 * self.w * inner


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


class PartitionedNeumann:
    """Ping-pong driver of the partitioned iteration: local fused step -> exchange -> norm all-reduce.

    The 8-byte all-reduce of ||t||^2 is issued asynchronously into one of two result slots, so it
    overlaps the next step's kernel; it is only waited for when the norm is read (term_norm) or when
    its slot is about to be reused two steps later.  With a SplitStep the term exchange itself
    overlaps the interior rows."""

    def __init__(self, part: RowPartition, local_step, exchange, t0_full: torch.Tensor,
                 x_local: torch.Tensor, group=None):
        self.part, self.local_step, self.exchange, self.group = part, local_step, exchange, group
        self.t = [t0_full, torch.zeros_like(t0_full)]
        self.x = x_local
        self.cur = 0
        self._norm = [torch.zeros(2, dtype=torch.float64, device=t0_full.device) for _ in range(2)]
        self._pending = [None, None]
        self._slot = 0
        self.steps_done = 0

    @property
    def norm2(self) -> torch.Tensor:
        """result slot of the most recent step (valid after term_norm() / a wait)"""
        return self._norm[self._slot]

    def step(self, reduce_norm: bool = True) -> None:
        p = self.part
        slot = 1 - self._slot
        if self._pending[slot] is not None:            # the all-reduce that last used this slot must be done
            self._pending[slot].wait()
            self._pending[slot] = None
        t_in, t_out = self.t[self.cur], self.t[1 - self.cur]
        t_out_local = t_out[p.lo:p.hi]
        if isinstance(self.local_step, SplitStep):
            self.local_step.run_boundary(t_in, t_out_local, self.x)
            early = getattr(self.exchange, "needs_only_boundary", False)
            handle = self.exchange.start(t_out) if early else None   # boundary rows are final: ship them while the interior computes
            self.local_step.run_interior(t_in, t_out_local, self.x, self._norm[slot])
            if not early:
                handle = self.exchange.start(t_out)                  # an all-gather needs the whole slice
            self.exchange.finish(handle)
        else:
            self.local_step(t_in, t_out_local, self.x, self._norm[slot])
            self.exchange(t_out)
        if reduce_norm and p.world > 1:
            self._pending[slot] = dist.all_reduce(self._norm[slot][:1], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._slot = slot
        self.cur = 1 - self.cur
        self.steps_done += 1

    @property
    def term(self) -> torch.Tensor:
        return self.t[self.cur]

    def term_norm(self) -> float:
        if self._pending[self._slot] is not None:
            self._pending[self._slot].wait()
            self._pending[self._slot] = None
        return float(self._norm[self._slot][0].item()) ** 0.5


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
