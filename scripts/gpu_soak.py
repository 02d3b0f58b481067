#!/usr/bin/env python3
"""Development aid: many frames in a row, alone and two at a time on two streams, every one compared with the first
bit for bit (the work queue hands pixels to waves in a different order every time; the frame must not care)."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rt = importlib.import_module("ray-tracer_amd")
import torch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
for name, W, H in (("cover", 3840, 2160), ("teapot_room", 1920, 1080), ("csg", 1920, 1080), ("groups", 1920, 1080)):
    flat = rt.Scene.from_sg(os.path.join(ROOT, "tests", "golden", "scenes", name + ".sg")).set_resolution(W, H).flatten()
    ctxs = [rt.Context(0), rt.Context(0)]
    for c in ctxs:
        c.upload(flat)
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    ref = torch.empty((H, W, 3), dtype=torch.float64, device="cuda")
    ctxs[0].render_rows(8, 0, 1, rt.MODE_FAST, d_rgb=ref.data_ptr())
    torch.cuda.synchronize()
    bufs = [torch.empty_like(ref), torch.empty_like(ref)]
    bad = 0
    for k in range(n):
        for j in range(2):
            with torch.cuda.stream(streams[j]):
                ctxs[j].render_rows(8, 0, 1, rt.MODE_FAST, d_rgb=bufs[j].data_ptr(), stream=streams[j].cuda_stream)
        torch.cuda.synchronize()
        bad += int(not torch.equal(bufs[0], ref)) + int(not torch.equal(bufs[1], ref))
    print(f"{name} {W}x{H}: {2 * n} frames, two at a time: {bad} differ from the first", flush=True)
    assert bad == 0
