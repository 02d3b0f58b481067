"""Row-tile partition of a frame over ranks and its reassembly on rank 0.

Pixels are independent in the reference's render loop (src/scene.rs:240-246), so N ranks (one per GPU)
each render the rows of their cyclic tiles -- tile i of ``tile_rows`` consecutive rows belongs to rank
``i % world`` -- and the only exchange is one gather of the float64 rows to rank 0, which puts them back
in row order.  ``part_row_indices`` states the same rule ``rtamd_part_rows`` / ``rtamd_render_rows``
implement on the device.  Nothing here touches the GPU library, so the exchange is testable on CPU
tensors with the gloo backend.
"""
import numpy as np
import torch
import torch.distributed as dist


def part_row_indices(vsize, tile_rows, part, n_parts):
    """Rows (increasing) of the frame that ``part`` of ``n_parts`` renders."""
    if vsize < 0 or tile_rows < 1 or n_parts < 1 or not 0 <= part < n_parts:
        raise ValueError("bad rows spec")
    n_tiles = (vsize + tile_rows - 1) // tile_rows
    rows = [np.arange(t * tile_rows, min(vsize, (t + 1) * tile_rows)) for t in range(part, n_tiles, n_parts)]
    return np.concatenate(rows).astype(np.int64) if rows else np.zeros(0, dtype=np.int64)


class FrameGather:
    """Buffers and the exchange step of one rank.

    ``local``  (max_rows, hsize, 3) float64: this rank's rows of the frame being rendered, compact, in
               increasing row order (ranks with fewer rows leave the tail unused so that every rank sends
               the same size).  There are two such buffers: ``exchange`` starts the gather of the one just
               filled and hands out the other, so the next frame renders while this one travels;
    ``frame``  (vsize, hsize, 3) float64 on rank 0, else None.
    """

    def __init__(self, hsize, vsize, tile_rows, rank, world, device):
        self.rank, self.world = rank, world
        self.rows = [part_row_indices(vsize, tile_rows, p, world) for p in range(world)]
        max_rows = max(len(r) for r in self.rows)
        self.n_rows = len(self.rows[rank])
        self.frame = torch.empty((vsize, hsize, 3), dtype=torch.float64, device=device) if rank == 0 else None
        self._pending = None
        if world == 1:
            self.local = self.frame  # a single rank renders straight into the frame
            return
        self._locals = [torch.empty((max_rows, hsize, 3), dtype=torch.float64, device=device) for _ in range(2)]
        self._cur = 0
        self.local = self._locals[0]
        self._gathered = [torch.empty_like(self.local) for _ in range(world)] if rank == 0 else None
        self._dst = [torch.from_numpy(r).to(device) for r in self.rows] if rank == 0 else None

    def _place(self):
        """Rank 0: the gathered parts go to their rows of the frame."""
        if self.rank == 0:
            for p in range(self.world):
                self.frame.index_copy_(0, self._dst[p], self._gathered[p][: self._dst[p].numel()])

    def gather(self):
        """Collective: every rank calls it after filling ``local``; rank 0 ends with the whole frame."""
        if self.world == 1:
            return self.frame
        self.finish()
        dist.gather(self.local, self._gathered, dst=0)
        self._place()
        return self.frame

    def exchange(self):
        """Collective, pipelined form of ``gather``: completes the frame submitted by the previous call
        (rank 0's ``frame`` then holds that frame), starts the gather of the frame just rendered into
        ``local`` without waiting for it, and makes ``local`` the other buffer.  ``finish`` completes the
        last frame.  With a device backend everything is ordered on streams: the caller may launch the next
        render right away, and it overlaps this frame's transfer."""
        if self.world == 1:
            return
        self.finish()
        self._pending = dist.gather(self.local, self._gathered, dst=0, async_op=True)
        self._cur ^= 1
        self.local = self._locals[self._cur]

    def finish(self):
        """Completes the frame whose gather ``exchange`` started (no-op when there is none)."""
        if self._pending is not None:
            self._pending.wait()
            self._pending = None
            self._place()
        return self.frame
