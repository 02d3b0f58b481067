"""rtamd_multi_*: one frame from several contexts below any host language -- each renders its interleaved row tiles on its
own GPU and writes its runs of rows into the frame on the first.  More than one GPU is not available to the tests, so
the device list names GPU 0 two, three and five times: the same contexts, streams, buffers and strided copies, with the
"peer" being the same card.  The frame must equal rtamd_render's to the bit."""
import numpy as np
import pytest

from conftest import sg_path

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,w,h", [("cover", 200, 117), ("teapot_lowres_room", 160, 80), ("csg", 96, 50)])
def test_multi_context_frame_equals_single_context_frame(rt, gpu_ctx, name, w, h):
    flat = rt.Scene.from_sg(sg_path(name)).set_resolution(w, h).flatten()
    gpu_ctx.upload(flat)
    want, want8 = gpu_ctx.render(w, h, rt.MODE_FAST, want_u8=True)
    strict = gpu_ctx.render(w, h, rt.MODE_STRICT)
    for devices, tile_rows, k0, k in (([0], 8, 1, 1), ([0, 0], 8, 1, 1), ([0, 0, 0], 8, 2, 1), ([0, 0], 3, 3, 2), ([0] * 5, 8, 1, 1),
                                      ([0, 0], 100, 1, 1), ([0, 0, 0], 1, 1, 3)):
        m = rt.MultiContext(devices).upload(flat)
        m.reset_stats()
        rgb, u8 = m.render(rt.MODE_FAST, tile_rows, k0, k, want_u8=True)
        st, render_ms, place_ms = m.stats()
        assert np.array_equal(rgb, want) and np.array_equal(u8, want8), (devices, tile_rows, k0, k)
        assert st.primary_rays == w * h and len(render_ms) == len(devices)
        assert np.array_equal(m.render(rt.MODE_STRICT, tile_rows, k0, k), strict)
        m.close()


def test_multi_context_frame_on_the_device_and_errors(rt, gpu_ctx):
    import torch
    w, h = 320, 180
    flat = rt.Scene.from_sg(sg_path("tables")).set_resolution(w, h).flatten()
    gpu_ctx.upload(flat)
    want, want8 = gpu_ctx.render(w, h, rt.MODE_FAST, want_u8=True)
    m = rt.MultiContext([0, 0, 0, 0])
    with pytest.raises(rt.RtamdError, match="-3"):
        m.render_on_device()
    m.upload(flat)
    d64, d8 = m.render_on_device(rt.MODE_FAST, 8, 2, 1, want=rt.WANT_F64 | rt.WANT_U8)
    assert d64 and d8
    class Raw:  # a view of device memory torch can read
        def __init__(self, ptr, typestr):
            self.__cuda_array_interface__ = {"shape": (h, w, 3), "typestr": typestr, "data": (ptr, False), "version": 2}
    out = torch.as_tensor(Raw(d64, "<f8"), device="cuda").cpu().numpy()
    out8 = torch.as_tensor(Raw(d8, "|u1"), device="cuda").cpu().numpy()
    assert np.array_equal(out, want) and np.array_equal(out8, want8)
    d64, d8 = m.render_on_device(rt.MODE_FAST, 8, 1, 1, want=rt.WANT_U8)
    assert d64 == 0 and d8
    with pytest.raises(rt.RtamdError):
        m.render(rt.MODE_FAST, 0)
    with pytest.raises(rt.RtamdError):
        rt.MultiContext([0, 99])
    with pytest.raises(rt.RtamdError):
        rt.MultiContext([])
    m.close()


def _union_of_spheres(tmp_path, n):
    """A CSG union of n unit spheres in a row along z, looked at along z: a ray through them meets 2 n intersections."""
    import sgtext
    s = sgtext.SG()
    m = s.mat(color=(0.8, 0.3, 0.2))
    def chain(k, z):
        if k == 1:
            s.shape("sphere", m, sgtext.translation(0, 0, z))
            return
        s.csg("union", m)
        s.shape("sphere", m, sgtext.translation(0, 0, z))
        chain(k - 1, z + 1.5)
    chain(n, 0.0)
    s.set_camera(64, 32, 0.6, (0, 0, -6), (0, 0, 0))
    s.light((-5, 5, -5))
    return s.write(tmp_path / f"union_{n}.sg")


def test_multi_context_reports_an_overflowing_csg_list_as_rtamd_render_does(rt, gpu_ctx, tmp_path):
    """14 spheres put 28 intersections on the central rays, more than the per-lane list of RTAMD_CSG_LIST_CAP = 24 holds:
    rtamd_render answers RTAMD_E_OVERFLOW (-5), and so must the frame assembled from several contexts, on every call."""
    flat = rt.Scene.from_sg(_union_of_spheres(tmp_path, 14)).flatten()
    gpu_ctx.upload(flat)
    with pytest.raises(rt.RtamdError, match="-5"):
        gpu_ctx.render(64, 32, rt.MODE_FAST)
    m = rt.MultiContext([0, 0]).upload(flat)
    for _ in range(2):  # (the counters are cumulative: the second frame overflows as the first did)
        with pytest.raises(rt.RtamdError, match="-5"):
            m.render(rt.MODE_FAST, 8, 1, 1)
    # a scene that fits renders through the same contexts afterwards
    ok = rt.Scene.from_sg(_union_of_spheres(tmp_path, 6)).flatten()
    gpu_ctx.upload(ok)
    want = gpu_ctx.render(64, 32, rt.MODE_FAST)
    m.upload(ok)
    assert np.array_equal(m.render(rt.MODE_FAST, 8, 1, 1), want)
    m.close()


def test_two_different_gpus_assemble_the_single_gpu_frame(rt, gpu_ctx):
    """The cross-device path proper (peer copies from the second GPU's stream into the frame on the first): runs only
    where the machine has more than one GPU, which no box of the test pool has -- the first multi-GPU box checks it."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU: the cross-device copies cannot run here")
    w, h = 400, 225
    flat = rt.Scene.from_sg(sg_path("cover")).set_resolution(w, h).flatten()
    gpu_ctx.upload(flat)
    want, want8 = gpu_ctx.render(w, h, rt.MODE_FAST, want_u8=True)
    n = min(torch.cuda.device_count(), 8)
    for devices in ([0, 1], list(range(n))):
        m = rt.MultiContext(devices).upload(flat)
        rgb, u8 = m.render(rt.MODE_FAST, 8, 1, 1, want_u8=True)
        assert np.array_equal(rgb, want) and np.array_equal(u8, want8), devices
        m.close()
