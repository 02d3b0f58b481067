"""Random scenes (spheres, cubes, cylinders, cones, planes, triangles; rotations, shears and non-uniform scales; groups,
CSG objects; patterns, glass, mirrors, one to three lights) rendered small on the GPU in both arithmetic modes and by the
CPU oracle, compared with the bars of tests/test_gpu_parity.py: STRICT 1e-10, FAST 1e-8 per channel, 8-bit within 1 LSB.
A ray that grazes an edge or a pattern boundary may land on either side of it in fast arithmetic, so a handful of its pixels
per frame (at most 0.1 %, at least 3) may miss the bars (strict arithmetic: none); every frame must be finite, render each pixel once and keep its
CSG lists within bounds.  As a program it runs a longer campaign: python tests/test_gpu_random_scenes.py [n_scenes] [seed] [big]."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import sgtext as sg

pytestmark = pytest.mark.gpu

def rand_xf(r, big=False):
    m = sg.translation(*(r.uniform(-2.5, 2.5, 3) + (0.137, 0.211, 0.319)))
    for _ in range(r.integers(0, 3)):
        m = m @ [sg.rotation_x, sg.rotation_y, sg.rotation_z][r.integers(0, 3)](r.uniform(0.1, 3.0))
    s = r.uniform(0.3, 1.3, 3) if r.random() < 0.6 else np.full(3, r.uniform(0.3, 1.3))
    if big:
        s = s * 1.7
    m = m @ sg.scaling(*s)
    if r.random() < 0.15:  # a shear
        sh = np.eye(4)
        sh[0, 1] = r.uniform(-0.5, 0.5)
        m = m @ sh
    return m


def rand_mat(r, s):
    kw = dict(color=r.uniform(0.1, 1.0, 3), ambient=r.uniform(0.05, 0.3), diffuse=r.uniform(0.3, 0.9), specular=r.uniform(0.0, 0.9),
              shininess=float(r.choice([10.0, 50.0, 200.0, 300.0, 37.5])))
    u = r.random()
    if u < 0.2:
        kw.update(reflective=r.uniform(0.2, 0.9))
    elif u < 0.4:
        kw.update(transparency=r.uniform(0.4, 0.95), ri=float(r.choice([1.0, 1.33, 1.5, 2.4])), reflective=r.uniform(0.0, 0.9), diffuse=0.2)
    if r.random() < 0.35:
        kind = str(r.choice(["stripe", "gradient", "ring", "checkers"]))
        kw.update(pattern=(kind, r.uniform(0, 1, 3), r.uniform(0, 1, 3), sg.translation(0.13, 0.07, 0.21) @ sg.scaling(*r.uniform(0.2, 0.9, 3)) @ sg.rotation_y(r.uniform(0, 1))))
    return s.mat(**kw)


def rand_leaf(r, s, in_csg=False):
    kind = str(r.choice(["sphere", "cube", "cylinder", "cone"] + ([] if in_csg else ["sphere", "cube"])))
    kw = {}
    if kind in ("cylinder", "cone"):
        lo = r.uniform(-1.5, -0.2)
        kw = dict(mn=lo, mx=lo + r.uniform(0.5, 2.0), closed=bool(r.random() < 0.7))
        if kind == "cone" and r.random() < 0.5:
            kw = dict(mn=-1.0, mx=0.0, closed=True)
    s.shape(kind, rand_mat(r, s), rand_xf(r), shadow=bool(r.random() < 0.9), **kw)


def rand_scene(seed, path):
    r = np.random.default_rng(seed)
    s = sg.SG()
    # a floor, sometimes walls
    s.shape("plane", rand_mat(r, s), sg.translation(0, -2.6, 0))
    if r.random() < 0.5:
        s.shape("plane", rand_mat(r, s), sg.translation(0, 0, 7.3) @ sg.rotation_x(np.pi / 2))
    for _ in range(r.integers(3, 14)):
        u = r.random()
        if u < 0.6:
            rand_leaf(r, s)
        elif u < 0.75:
            n = int(r.integers(2, 5))
            s.group(n, rand_mat(r, s), rand_xf(r))
            for _ in range(n):
                if r.random() < 0.2:
                    s.group(2, rand_mat(r, s), rand_xf(r))
                    rand_leaf(r, s), rand_leaf(r, s)
                else:
                    rand_leaf(r, s)
        elif u < 0.9:
            s.csg(str(r.choice(["union", "intersection", "difference"])), rand_mat(r, s), 0)
            if r.random() < 0.3:
                s.csg(str(r.choice(["union", "intersection", "difference"])), rand_mat(r, s), 0)
                rand_leaf(r, s, True), rand_leaf(r, s, True)
            else:
                rand_leaf(r, s, True)
            rand_leaf(r, s, True)
        else:  # a few triangles: a tetrahedron somewhere
            v, f = sg.tetrahedron()
            m = rand_xf(r)
            mt = rand_mat(r, s)
            s.group(len(f), mt, m)
            for a, b, c in f:
                s.tri(mt, v[a], v[b], v[c])
    for _ in range(r.integers(1, 4)):
        s.light(r.uniform(-8, 8, 3) * (1, 0.5, 1) + (0.3, 7.0, -6.0), r.uniform(0.3, 0.8, 3))
    w, h = (64, 48) if seed % 3 else (128, 72)
    s.set_camera(w, h, r.uniform(0.7, 1.3), (r.uniform(-3, 3) + 0.17, r.uniform(0.5, 4.0), -9.1), (0.1, -0.3, 0.2))
    s.write(path)
    return w, h




def check_scene(rt, oracle, ctx, seed, path, big=False):
    """-> (ok, report line).  big: the frame at 2048x1152 (the kernels, bounds records and runs of triangles that large
    frames get), compared on a lattice of its pixels."""
    w, h = rand_scene(seed, path)
    if big:
        w, h = 2048, 1152
        pix = np.arange(seed % 97, w * h, 787)
        ref, ref8, _ = oracle.OracleScene(path).render_pixels(pix, w, h, threads=8)
    else:
        pix = np.arange(w * h)
        ref, ref8, _ = oracle.OracleScene(path).render(w, h, threads=8)
        ref, ref8 = ref.reshape(-1, 3), ref8.reshape(-1, 3)
    ctx.upload(rt.Scene.from_sg(path).set_resolution(w, h).flatten())
    line = [f"scene {seed} {w}x{h}:"]
    ok = True
    for name, mode, tol in (("strict", rt.MODE_STRICT, 1e-10), ("fast", rt.MODE_FAST, 1e-8)):
        ctx.reset_stats()
        rgb, u8 = ctx.render(w, h, mode, want_u8=True)
        st = ctx.stats()
        rgb, u8 = rgb.reshape(-1, 3)[pix], u8.reshape(-1, 3)[pix]
        diff = np.abs(rgb - ref).max(axis=1)
        off = int((diff > tol).sum())
        lsb = np.abs(u8.astype(int) - ref8.astype(int)).max(axis=1)
        far = int((lsb > 1).sum())
        # A leaf the bounds records skip by mistake, or a slip of the slot-based CSG filter, shows as a handful of pixels
        # with another object's colour: differences of 1e-2 and more.  STRICT arithmetic performs the reference's operations
        # (rays level by level through groups and CSG objects included): not a single pixel beyond its tolerance.  The
        # allowance below is for FAST pixels on a knife edge (a tie in t against a real miss).
        gross = int((diff > 1e-6).sum())
        where = [f"({int(p % w)},{int(p // w)}: {diff[i]:.2e})" for i, p in enumerate(pix) if diff[i] > tol or lsb[i] > 1][:8]
        line.append(f"{name} beyond tolerance {off}, beyond 1e-6 {gross}, 8-bit beyond 1 LSB {far}, overflows {st.csg_overflows}{' at ' + ' '.join(where) if where else ''};")
        few = max(3, len(pix) // 1000)
        ok = ok and bool(np.isfinite(rgb).all()) and st.primary_rays == w * h and st.csg_overflows == 0 and off <= few and far <= few
        if mode == rt.MODE_STRICT:
            ok = ok and off == 0 and far == 0
    return ok, " ".join(line)


@pytest.mark.parametrize("seed", [100000 + k for k in range(40)])
def test_random_scene_matches_the_oracle(rt, oracle, gpu_ctx, tmp_path, seed):
    ok, line = check_scene(rt, oracle, gpu_ctx, seed, str(tmp_path / "scene.sg"))
    print(line)
    assert ok, line


@pytest.mark.parametrize("seed", [300000 + k for k in range(6)])
def test_random_scene_at_a_large_frame_size(rt, oracle, gpu_ctx, tmp_path, seed):
    ok, line = check_scene(rt, oracle, gpu_ctx, seed, str(tmp_path / "scene.sg"), big=True)
    print(line)
    assert ok, line


if __name__ == "__main__":
    import importlib
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    rt_ = importlib.import_module("ray-tracer_amd")
    import oracle_lib
    n_scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    big_ = len(sys.argv) > 3 and sys.argv[3] == "big"
    ctx_ = rt_.Context(0)
    tmp = tempfile.mkdtemp(prefix="rtamd_random_")
    bad = 0
    for k in range(n_scenes):
        seed = seed0 * 100000 + k
        path = os.path.join(tmp, f"s{seed}.sg")
        try:
            ok, line = check_scene(rt_, oracle_lib, ctx_, seed, path, big_)
        except Exception as e:
            ok, line = False, f"scene {seed}: {type(e).__name__}: {e}"
        print(line + ("" if ok else "   <<<<<< LOOK"), flush=True)
        if not ok:
            bad += 1
            out = os.path.join(root, "gpurun_out")
            os.makedirs(out, exist_ok=True)
            if os.path.exists(path):
                import shutil
                shutil.copy(path, os.path.join(out, os.path.basename(path)))
    print(f"{n_scenes} scenes, {bad} to look at")
