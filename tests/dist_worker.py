"""One rank of the CPU rehearsal of the multi-GPU frame assembly (started by test_distributed.py).

The rows a rank would render on its GPU are produced here by the CPU oracle -- the exchange, the row
partition and the reassembly are the code bench.py runs (ray-tracer_amd/rows.py), over gloo."""
import importlib
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    w, h, tile_rows, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    rank0_parts, other_parts = (int(sys.argv[5]), int(sys.argv[6])) if len(sys.argv) > 6 else (1, 1)
    u8 = len(sys.argv) > 7 and sys.argv[7] == "u8"  # the 8-bit channels travel instead of the f64 pixels
    blank = 0 if u8 else float("nan")
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rows_mod = importlib.import_module("ray-tracer_amd.rows")
    import oracle_lib
    o = oracle_lib.OracleScene(os.path.join(ROOT, "tests", "golden", "scenes", "cover.sg"))
    shares = rows_mod.shares_for(world, rank0_parts, other_parts)
    fg = rows_mod.FrameGather(w, h, tile_rows, rank, world, torch.device("cpu"), shares=shares, dtype=torch.uint8 if u8 else torch.float64)
    assert fg.sent_bytes == fg.sent_rows * w * 3 * (1 if u8 else 8)
    mine = rows_mod.part_row_indices(h, tile_rows, fg.part, fg.n_parts, fg.span)
    assert fg.n_rows == len(mine)
    if (rank0_parts, other_parts) == (1, 1):
        assert np.array_equal(mine, rows_mod.part_row_indices(h, tile_rows, rank, world))
    fg.local.fill_(blank)
    if len(mine):
        pix = (mine[:, None] * w + np.arange(w)[None, :]).ravel()
        rgb, rgb8, _ = o.render_pixels(pix, w, h)
        fg.local[: len(mine)] = torch.from_numpy((rgb8 if u8 else rgb).reshape(len(mine), w, 3))
    for _ in range(2):  # the exchange is repeatable
        frame = fg.gather()
    # the pipelined form bench.py uses: two frames under way (here: frame + k)
    mine_rows = fg.local[: len(mine)].clone()
    first = frame.clone() if rank == 0 else None
    for k in range(1, 6):
        buf = fg.begin()  # completes frame k - 2, whose slot this is
        if rank == 0 and k > 2:
            assert torch.equal(fg._frames[fg._cur], first + (k - 2)), k
        buf.fill_(blank)
        buf[: len(mine)] = mine_rows + k  # (8-bit channels wrap around, the same on every rank)
        fg.submit()
    frame = fg.finish()
    if rank == 0:
        assert torch.equal(frame, first + 5)
        assert torch.equal(fg._frames[fg._newest ^ 1], first + 4)
        frame = first
    dist.barrier()
    if rank == 0:
        np.save(out, frame.numpy())
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
