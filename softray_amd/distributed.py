"""Row-tiled multi-GPU rendering: one process per GPU, the frame split into interleaved strips of rows,
one RCCL gather over xGMI to rank 0.  torch.distributed is plumbing only; the pixels are
produced by libsoftray_hip.so on each rank's GPU.

The reference already parallelises by row blocks (Renderer.cs:1659-1670) and exposes
rayTraceStartRow/EndRow (:135-136); strips are interleaved rather than contiguous because background
rows are ~100x cheaper than shadowed rows.
"""
import torch
import torch.distributed as dist


def owned_rows(height, strip_rows, world, rank, start_row=0, end_row=None):
    """Image rows rank `rank` renders: r in [start_row, end_row] with (r // strip_rows) % world == rank
    (identical to the predicate of sr_frame.strip_* in include/softray.h)."""
    end_row = height - 1 if end_row is None else end_row
    a, b = min(max(0, start_row), height - 1), min(max(0, end_row), height - 1)
    return [r for r in range(a, b + 1) if (r // strip_rows) % world == rank]


class StripGather:
    """Pre-allocated buffers + the one exchange step.  `local` is each rank's compact strip buffer
    (owned rows in order, int32 [rows_r * width]), padded to the largest rank so that every rank sends the
    same amount."""

    def __init__(self, width, height, strip_rows, world, rank, device, start_row=0, end_row=None):
        self.width, self.height, self.world, self.rank = width, height, world, rank
        self.rows = [owned_rows(height, strip_rows, world, r, start_row, end_row) for r in range(world)]
        self.counts = [len(r) * width for r in self.rows]
        self.maxc = max(self.counts) if self.counts else 0
        self.local = torch.empty(max(1, self.maxc), dtype=torch.int32, device=device)
        self.full = None
        self.gather_list = None
        if rank == 0:
            self.full = torch.zeros((height, width), dtype=torch.int32, device=device)
            # one receive buffer, one row per rank: when every rank owns the same number of rows (the usual case) the
            # de-interleave is a single indexed copy instead of one per rank
            self.gbuf = torch.empty((world, max(1, self.maxc)), dtype=torch.int32, device=device)
            self.gather_list = [self.gbuf[k] for k in range(world)]
            self.row_idx = [torch.tensor(r, dtype=torch.long, device=device) for r in self.rows]
            self.uniform = self.maxc > 0 and all(c == self.maxc for c in self.counts)
            self.rows_all = torch.cat(self.row_idx) if self.uniform else None

    def start(self):
        """Enqueue the gather (RCCL send/recv over xGMI on GPUs, gloo on CPU) behind the work already queued on the current stream
        and return its handle (None for one rank): the caller's stream is free for the next frame while the strips travel."""
        if self.world == 1:
            return None
        return dist.gather(self.local, self.gather_list, dst=0, async_op=True)

    def finish(self):
        """De-interleave on rank 0 (on the current stream; the gather must have been waited for on it)."""
        if self.rank != 0:
            return None
        if self.world == 1:
            self.full[self.row_idx[0]] = self.local[: self.counts[0]].view(-1, self.width)
        elif self.uniform:
            self.full[self.rows_all] = self.gbuf.view(-1, self.width)
        else:
            for k in range(self.world):
                if self.counts[k]:
                    self.full[self.row_idx[k]] = self.gather_list[k][: self.counts[k]].view(-1, self.width)
        return self.full

    def exchange(self):
        """start() + wait + finish() on the current stream."""
        work = self.start()
        if work is not None:
            work.wait()
        return self.finish()
