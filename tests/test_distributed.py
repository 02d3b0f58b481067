"""The N>1 path on CPU: cyclic row tiles over ranks, one gather to rank 0 (gloo stands in for RCCL)."""
import importlib
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, sg_path


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("vsize,tile_rows,n_parts", [(2160, 8, 8), (27, 4, 2), (5, 8, 4), (16, 1, 3), (0, 8, 2)])
def test_row_partition_is_a_cyclic_tiling(vsize, tile_rows, n_parts):
    rows_mod = importlib.import_module("ray-tracer_amd.rows")
    parts = [rows_mod.part_row_indices(vsize, tile_rows, p, n_parts) for p in range(n_parts)]
    allrows = np.concatenate(parts)
    assert sorted(allrows.tolist()) == list(range(vsize))  # every row exactly once
    for p, r in enumerate(parts):
        assert np.all(np.diff(r) > 0) and np.all((r // tile_rows) % n_parts == p)
    if vsize >= tile_rows * n_parts:
        assert max(map(len, parts)) - min(map(len, parts)) <= tile_rows  # balanced to within one tile


def test_row_partition_rejects_bad_specs():
    rows_mod = importlib.import_module("ray-tracer_amd.rows")
    for bad in [(10, 0, 0, 1), (10, 8, 2, 2), (10, 8, -1, 2), (10, 8, 0, 0)]:
        with pytest.raises(ValueError):
            rows_mod.part_row_indices(*bad)


@pytest.mark.parametrize("world,w,h,tile_rows", [(2, 40, 27, 4), (3, 24, 10, 8)])
def test_frame_gather_over_gloo(oracle, tmp_path, world, w, h, tile_rows):
    """world processes each produce their rows; rank 0 must end with exactly the frame a single process
    renders (the second case leaves rank 2 without any row)."""
    out = str(tmp_path / "frame.npy")
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_worker.py"), str(w), str(h),
                                       str(tile_rows), out], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    logs = []
    for p in procs:
        try:
            logs.append(p.communicate(timeout=240)[0])
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail("distributed rehearsal timed out")
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)[-3000:]
    whole, _, _ = oracle.OracleScene(sg_path("cover")).render(w, h)
    assert np.array_equal(np.load(out), whole)
