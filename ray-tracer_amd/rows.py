"""Row-tile partition of a frame over ranks and its reassembly on rank 0.

Pixels are independent in the reference's render loop (src/scene.rs:240-246), so N ranks (one per GPU)
each render the rows of their cyclic tiles -- tile i of ``tile_rows`` consecutive rows belongs to part
``i % n_parts``, and a rank holds one part or a few adjacent ones -- and the only exchange is one gather of the
float64 rows to rank 0, which puts them back in row order.  Rank 0's own rows do not travel, so it may hold a
larger share than the others (``shares``): every other rank reaches rank 0 over one xGMI link, and for short
frames that link, not the rendering, is what a frame waits for.  ``part_row_indices`` states the same rule ``rtamd_part_rows`` / ``rtamd_render_rows``
implement on the device.  Nothing here touches the GPU library, so the exchange is testable on CPU
tensors with the gloo backend.
"""
import numpy as np
import torch
import torch.distributed as dist


def part_row_indices(vsize, tile_rows, part, n_parts, span=1):
    """Rows (increasing) of the frame that parts ``part .. part + span - 1`` of ``n_parts`` render."""
    if vsize < 0 or tile_rows < 1 or n_parts < 1 or span < 1 or part < 0 or part + span > n_parts:
        raise ValueError("bad rows spec")
    n_tiles = (vsize + tile_rows - 1) // tile_rows
    rows = [np.arange(t * tile_rows, min(vsize, (t + 1) * tile_rows))
            for first in range(part, n_tiles, n_parts) for t in range(first, min(first + span, n_tiles))]
    return np.concatenate(rows).astype(np.int64) if rows else np.zeros(0, dtype=np.int64)


def shares_for(world, rank0_parts=1, other_parts=1):
    """(n_parts, [(part, span) per rank]): rank 0 holds ``rank0_parts`` adjacent parts, every other rank
    ``other_parts``.  (1, 1) is the plain cyclic split with n_parts = world."""
    if world < 1 or rank0_parts < 1 or other_parts < 1:
        raise ValueError("bad shares")
    spans = [(0, rank0_parts)] + [(rank0_parts + (r - 1) * other_parts, other_parts) for r in range(1, world)]
    return rank0_parts + (world - 1) * other_parts, spans


def pick_share(trial_ms, margin=0.97):
    """From measured ms per frame of the splits tried ({"k0:k": ms}, the equal split "1:1" among them): the
    fastest one if it beats the equal split by the margin, else the equal split.  Returns (k0, k)."""
    best = min(trial_ms, key=lambda s: (trial_ms[s], s))
    if best != "1:1" and trial_ms[best] < margin * trial_ms["1:1"]:
        k0, k = best.split(":")
        return int(k0), int(k)
    return 1, 1


class FrameGather:
    """Buffers and the exchange step of one rank, for two frames in flight.

    A frame slot holds ``local`` (rows, hsize, 3) float64 -- this rank's rows, compact, in increasing row
    order; what travels is its first ``sent_rows`` rows, the most any rank other than 0 holds (ranks with fewer
    leave the tail unused so that every rank sends the same size; rank 0 sends a stand-in and takes its own rows
    from ``local``) -- and on rank 0 the gathered parts and the assembled ``frame`` (vsize, hsize, 3).
    ``begin`` hands out the next slot's ``local`` after completing the frame that slot carried before, ``submit`` starts that slot's gather without
    waiting for it.  So a frame's rows travel while the next frame renders, and -- when the caller alternates two
    streams -- two renders may overlap as well: nothing a slot touches is shared with the other slot.
    ``gather`` is the plain blocking form.
    """

    def __init__(self, hsize, vsize, tile_rows, rank, world, device, collective=None, shares=None, dtype=torch.float64):
        # dtype: torch.float64 -- Canvas.pixels, 24 bytes per pixel -- or torch.uint8 -- the 8-bit channels an image file
        # holds (src/tuple.rs:226-237), 3 bytes per pixel: an eighth of the bytes over every link
        self.rank, self.world, self.dtype = rank, world, dtype
        self.n_parts, self.spans = shares_for(world) if shares is None else shares
        self.part, self.span = self.spans[rank]
        # a single rank needs no exchange; `collective=True` makes it go through the gather all the same (a rehearsal
        # of the multi-rank code with one process)
        self._collective = world > 1 if collective is None else bool(collective)
        self.rows = [part_row_indices(vsize, tile_rows, p, self.n_parts, s) for p, s in self.spans]
        if sorted(np.concatenate(self.rows).tolist()) != list(range(vsize)):
            raise ValueError("the shares do not cover every row exactly once")
        self.n_rows = len(self.rows[rank])
        self.sent_rows = max([len(r) for r in self.rows[1:]], default=0) or len(self.rows[0])
        max_rows = max(self.n_rows, self.sent_rows)
        self._frames = [torch.empty((vsize, hsize, 3), dtype=dtype, device=device) if rank == 0 else None for _ in range(2)]
        self._pending = [None, None]
        self._cur = 0      # the slot begin() handed out last
        self._newest = 0   # the slot submitted last
        if not self._collective:
            self._locals = self._frames  # a single rank renders straight into the frame
        else:
            self._locals = [torch.empty((max_rows, hsize, 3), dtype=dtype, device=device) for _ in range(2)]
            self._gathered = [[torch.empty((self.sent_rows, hsize, 3), dtype=dtype, device=device) for _ in range(world)]
                              if rank == 0 else None for _ in range(2)]
            self._dst = [torch.from_numpy(r).to(device) for r in self.rows] if rank == 0 else None
        self.local = self._locals[0]
        self.frame = self._frames[0]
        # bytes a rank other than 0 puts on its link per frame (0 for a single rank that does not rehearse the exchange)
        self.sent_bytes = self.sent_rows * hsize * 3 * torch.empty((), dtype=dtype).element_size() if self._collective else 0

    def _complete(self, slot):
        """Waits for the gather of ``slot`` (if one is under way) and, on rank 0, puts the parts into the frame."""
        if self._pending[slot] is None:
            return
        self._pending[slot].wait()
        self._pending[slot] = None
        if self.rank == 0:
            for p in range(self.world):
                src = self._locals[slot] if p == 0 else self._gathered[slot][p]  # rank 0's rows did not travel
                self._frames[slot].index_copy_(0, self._dst[p], src[: self._dst[p].numel()])

    def begin(self):
        """The buffer to render the next frame into (``local``); completes the frame its slot carried before."""
        self._cur ^= 1
        if self._collective:
            self._complete(self._cur)
        self.local = self._locals[self._cur]
        return self.local

    def submit(self):
        """Collective: starts the gather of the frame rendered into ``local``; returns at once.  With a device
        backend the transfer is ordered after the work queued so far on the current stream."""
        self._newest = self._cur
        if self._collective:
            self._pending[self._cur] = dist.gather(self.local[: self.sent_rows], self._gathered[self._cur], dst=0, async_op=True)

    def finish(self):
        """Completes every frame under way, oldest first; rank 0 gets the newest frame (``frame``)."""
        if self._collective:
            self._complete(self._newest ^ 1)
            self._complete(self._newest)
        self.frame = self._frames[self._newest]
        return self.frame

    def gather(self):
        """Collective, blocking: every rank calls it after filling ``local``; rank 0 ends with the whole frame."""
        self.submit()
        return self.finish()
