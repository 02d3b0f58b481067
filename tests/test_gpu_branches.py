"""GPU parity for device branches that no committed scene reaches: gradient / ring / test patterns
(src/pattern.rs:28-52), CSG objects with a transform of their own and under transformed groups (their children's
normals and patterns ignore every transform above the child, src/shapes/csg.rs has no refresh_parents; src/shape.rs:49-70),
triangles below a CSG node, lights with fractional shininess next to whole ones (src/material.rs:121-126), rays with
direction components that are exactly zero against bounds records, and runs of triangles the upload leaves empty.
Same bars as tests/test_gpu_parity.py: STRICT 1e-10, FAST 1e-8 per channel against the oracle, 8-bit within 1 LSB."""
import numpy as np
import pytest

import sgtext as sg
from conftest import sg_path

pytestmark = pytest.mark.gpu

STRICT_TOL = 1e-10
FAST_TOL = 1e-8


def _check_frame(rt, oracle, ctx, path, w, h, max_remaining=4, may_differ=0.0):
    """may_differ: share of the pixels that may miss the f64 tolerance (they still count for the 8-bit bar)."""
    ref, ref8, _ = oracle.OracleScene(path).render(w, h, threads=8)
    ctx.upload(rt.Scene.from_sg(path).set_resolution(w, h).flatten(max_remaining))
    out = {}
    for name, m, tol in (("strict", rt.MODE_STRICT, STRICT_TOL), ("fast", rt.MODE_FAST, FAST_TOL)):
        ctx.reset_stats()
        rgb, u8 = ctx.render(w, h, m, want_u8=True)
        st = ctx.stats()
        err = float(np.abs(rgb - ref).max())
        lsb = np.abs(u8.astype(int) - ref8.astype(int)).max(axis=2)
        print(f"{name}: max|diff| {err:.3e}, bit-exact {100 * (rgb == ref).all(axis=2).mean():.2f}%, 8-bit equal {100 * (lsb == 0).mean():.3f}%")
        assert np.isfinite(rgb).all() and st.csg_overflows == 0 and st.primary_rays == w * h
        off = np.argwhere(np.abs(rgb - ref).max(axis=2) > tol)
        if len(off):
            print(f"  {len(off)} pixels beyond {tol}: (y, x) {off[:20].tolist()}")
        assert len(off) <= may_differ * w * h, (name, err)
        assert (lsb <= 1).mean() >= 0.999
        out[name] = rgb
    return ref, out


def _check_rays(rt, oracle, ctx, path, rays, remainings=(0, 1, 4), max_remaining=5):
    o = oracle.OracleScene(path)
    ctx.upload(rt.Scene.from_sg(path).flatten(max_remaining))
    rays = np.asarray(rays, dtype=float).reshape(-1, 6)
    for remaining in remainings:
        ref = o.color_at(rays, remaining)
        for m, tol in ((rt.MODE_STRICT, STRICT_TOL), (rt.MODE_FAST, FAST_TOL)):
            got = ctx.color_at(rays, remaining, m)
            worst = float((np.abs(got - ref) / np.maximum(1.0, np.abs(ref))).max())
            assert worst <= tol, (remaining, m, worst)
    return ref


def _room(s, floor_mat):
    s.shape("plane", floor_mat, sg.translation(0, -1, 0))
    wall = s.mat(color=(0.7, 0.75, 0.8), specular=0.1, shininess=10)
    s.shape("plane", wall, sg.translation(0, 0, 6) @ sg.rotation_x(np.pi / 2))


# ------------------------------------------------------------------ patterns
def _pattern_scene(tmp_path, w=200, h=100):
    s = sg.SG()
    s.set_camera(w, h, 1.0, (0.5, 2.2, -7.5), (0, 0.3, 0))
    s.light((-6, 8, -9), (1, 0.95, 0.9))
    ring = s.mat(specular=0.2, reflective=0.15, pattern=("ring", (0.9, 0.3, 0.2), (0.1, 0.2, 0.7),
                                                        sg.translation(0.3, 0, -0.4) @ sg.rotation_y(0.5) @ sg.scaling(0.37, 1, 0.61)))
    _room(s, ring)
    grad = s.mat(diffuse=0.8, pattern=("gradient", (1, 0, 0.1), (0, 0.4, 1), sg.rotation_z(0.7) @ sg.scaling(0.5, 0.5, 0.5) @ sg.translation(0.13, 0, 0)))
    s.shape("sphere", grad, sg.translation(-2.2, 0.1, 0.4) @ sg.rotation_y(0.3) @ sg.scaling(1.1, 0.9, 1.0))
    tst = s.mat(ambient=0.3, diffuse=0.6, specular=0.3, shininess=40, pattern=("test", (0, 0, 0), (0, 0, 0),
                                                                                sg.scaling(2.0, 1.5, 2.5) @ sg.translation(-0.5, -0.5, -0.5)))
    s.shape("cube", tst, sg.translation(0.2, -0.07, 0.8) @ sg.rotation_y(0.6) @ sg.scaling(0.8, 0.8, 0.8))
    # the same patterns two groups deep: pattern_at_object goes through every parent (src/pattern.rs:57-61, src/shape.rs:49-55)
    plain = s.mat()
    s.group(2, plain, sg.translation(2.4, 0.2, 0.3) @ sg.rotation_z(0.25))
    s.group(2, plain, sg.scaling(0.9, 1.2, 0.9) @ sg.rotation_y(-0.4))
    ring2 = s.mat(reflective=0.1, pattern=("ring", (0.2, 0.8, 0.3), (0.9, 0.9, 0.1), sg.scaling(0.11, 0.11, 0.11) @ sg.rotation_x(0.9)))
    s.shape("sphere", ring2, sg.translation(0, 0.3, 0))
    s.shape("cylinder", grad, sg.translation(0.9, 0, 0.5) @ sg.scaling(0.3, 0.5, 0.3), mn=-1, mx=1.5, closed=True)
    grad2 = s.mat(transparency=0.4, ri=1.2, reflective=0.2,
                  pattern=("gradient", (0.1, 0.9, 0.9), (0.9, 0.1, 0.5), sg.rotation_y(1.1) @ sg.scaling(0.7, 1, 1)))
    s.shape("cube", grad2, sg.translation(-1.1, -0.4, -0.6) @ sg.scaling(0.35, 0.6, 0.35))
    return s.write(tmp_path / "patterns.sg"), w, h


def test_gradient_ring_and_test_patterns(rt, oracle, gpu_ctx, tmp_path):
    path, w, h = _pattern_scene(tmp_path)
    ref, _ = _check_frame(rt, oracle, gpu_ctx, path, w, h)
    assert len(np.unique(ref.reshape(-1, 3).round(6), axis=0)) > 2000  # gradients and test patterns vary from pixel to pixel
    rng = np.random.default_rng(5)
    rays = np.concatenate([rng.uniform(-3, 3, (384, 3)) + [0, 1.5, -2], rng.normal(size=(384, 3))], axis=1)
    _check_rays(rt, oracle, gpu_ctx, path, rays)


# ------------------------------------------------------------------ CSG below transforms
def _csg_transform_scene(tmp_path, w=200, h=100):
    s = sg.SG()
    s.set_camera(w, h, 0.9, (0.8, 2.6, -8.0), (0, 0.4, 0))
    s.light((-5, 7, -8))
    s.light((6, 5, -4), (0.4, 0.4, 0.5))
    floor = s.mat(specular=0, reflective=0.2, pattern=("checkers", (0.3, 0.3, 0.3), (0.7, 0.7, 0.7), sg.scaling(0.8, 0.8, 0.8)))
    _room(s, floor)
    plain = s.mat()
    stripe = s.mat(color=(1, 0.2, 0.1), pattern=("stripe", (1, 0.3, 0.1), (0.2, 0.2, 0.9), sg.rotation_z(0.6) @ sg.scaling(0.21, 0.21, 0.21)))
    grad = s.mat(color=(0.1, 0.9, 0.3), specular=0.4, shininess=30,
                 pattern=("gradient", (0.1, 0.9, 0.3), (0.9, 0.9, 0.1), sg.scaling(0.6, 1, 1) @ sg.rotation_y(0.3)))
    glass = s.mat(color=(0.05, 0.05, 0.15), diffuse=0.3, reflective=0.6, transparency=0.7, ri=1.4)
    # a CSG with a transform of its own: rays go through it (src/shape.rs:39-42), its children's normals and patterns do not
    s.csg("difference", plain, sg.translation(-2.1, 0.3, 0.2) @ sg.rotation_y(0.7) @ sg.scaling(1.0, 1.3, 0.8))
    s.shape("cube", stripe, sg.rotation_z(0.2))
    s.shape("sphere", grad, sg.translation(0.5, 0.6, -0.5) @ sg.scaling(0.9, 0.9, 0.9))
    # a CSG under two transformed groups, next to an ordinary leaf that does get the chain
    s.group(2, plain, sg.translation(2.0, 0.0, 0.6) @ sg.rotation_z(-0.3))
    s.group(2, plain, sg.rotation_y(0.5) @ sg.scaling(0.8, 1.1, 0.8))
    s.csg("intersection", plain, sg.translation(0, 0.4, 0) @ sg.rotation_x(0.4))
    s.shape("sphere", stripe, sg.scaling(1.25, 1.25, 1.25))
    s.csg("union", plain, sg.rotation_y(0.3))  # nested, with its own transform as well
    s.shape("cube", grad, sg.scaling(0.9, 0.9, 0.9))
    s.shape("cylinder", glass, sg.rotation_x(np.pi / 2) @ sg.scaling(0.4, 1, 0.4), mn=-2, mx=2, closed=True)
    s.shape("cone", stripe, sg.translation(1.4, -1.0, -0.3) @ sg.scaling(0.5, 1.2, 0.5), mn=0, mx=1, closed=True)
    s.shape("sphere", glass, sg.translation(0.1, -0.3, -1.6) @ sg.scaling(0.7, 0.7, 0.7))
    return s.write(tmp_path / "csg_xf.sg"), w, h


def test_csg_with_transforms_and_under_groups(rt, oracle, gpu_ctx, tmp_path):
    """A child of a transformed CSG gets its normal from a point in the wrong space (the quirk under test): for a cube
    that is the normal of the wrong face, over_point then stays ON the surface, and whether its shadow ray meets that
    surface at t = +-1e-16 is decided by the last bit -- which the two arithmetic modes round differently.  Such pixels
    have positive measure here (a few in 100 000 were seen), so a few may miss the f64 tolerance."""
    path, w, h = _csg_transform_scene(tmp_path)
    _check_frame(rt, oracle, gpu_ctx, path, w, h, may_differ=3e-4)
    rng = np.random.default_rng(11)
    rays = np.concatenate([rng.uniform(-3.5, 3.5, (384, 3)) + [0, 1.0, -1], rng.normal(size=(384, 3))], axis=1)
    _check_rays(rt, oracle, gpu_ctx, path, rays)


def test_csg_quirk_is_visible(rt, oracle, gpu_ctx, tmp_path):
    """The scene above with the CSG transforms moved into the children would look the same if normals and patterns
    followed the CSG's transform; they do not (the reference's Csg never hands parent_transforms down), so the frames
    differ -- the test above is not passing on a case where the quirk cannot show."""
    s = sg.SG()
    s.set_camera(96, 48, 0.9, (0, 1.5, -6), (0, 0, 0))
    s.light((-5, 7, -8))
    plain = s.mat()
    stripe = s.mat(pattern=("stripe", (1, 0.3, 0.1), (0.2, 0.2, 0.9), sg.scaling(0.21, 0.21, 0.21)))
    t = sg.rotation_y(0.7) @ sg.scaling(1.0, 1.5, 0.8)
    s.csg("union", plain, t)
    s.shape("sphere", stripe)
    s.shape("cube", stripe, sg.translation(1.2, 0, 0) @ sg.scaling(0.5, 0.5, 0.5))
    a = s.write(tmp_path / "quirk_a.sg")
    s2 = sg.SG()
    s2.set_camera(96, 48, 0.9, (0, 1.5, -6), (0, 0, 0))
    s2.light((-5, 7, -8))
    plain = s2.mat()
    stripe = s2.mat(pattern=("stripe", (1, 0.3, 0.1), (0.2, 0.2, 0.9), sg.scaling(0.21, 0.21, 0.21)))
    s2.group(2, plain, t)
    s2.shape("sphere", stripe)
    s2.shape("cube", stripe, sg.translation(1.2, 0, 0) @ sg.scaling(0.5, 0.5, 0.5))
    b = s2.write(tmp_path / "quirk_b.sg")
    ra, _ = _check_frame(rt, oracle, gpu_ctx, a, 96, 48)
    rb, _ = _check_frame(rt, oracle, gpu_ctx, b, 96, 48)
    assert np.abs(ra - rb).max() > 0.05


# ------------------------------------------------------------------ triangles below CSG nodes
def _mesh_nodes(s, mat, verts, faces, smooth, xf=0):
    s.group(len(faces), mat, xf)
    for f in faces:
        p = [verts[i] for i in f]
        s.tri(mat, *p, normals=[q / np.linalg.norm(q) for q in p] if smooth else None)


@pytest.mark.parametrize("shape,smooth", [("tetrahedron", False), ("icosahedron", True), ("icosahedron", False)])
def test_triangles_below_a_csg_node(rt, oracle, gpu_ctx, tmp_path, shape, smooth):
    """Difference(cube, mesh) and Intersection(mesh, sphere): the tetrahedron keeps every object at or below 12 possible
    intersections per ray (the short per-lane lists), the icosahedron does not (the long ones); both kernels are the
    variants with the code for triangles."""
    verts, faces = getattr(sg, shape)()
    w, h = 192, 96
    s = sg.SG()
    s.set_camera(w, h, 0.9, (0.6, 2.4, -7.0), (0, 0.3, 0))
    s.light((-5, 7, -8))
    floor = s.mat(specular=0, reflective=0.3, pattern=("checkers", (0.3, 0.3, 0.3), (0.7, 0.7, 0.7), sg.scaling(0.8, 0.8, 0.8)))
    _room(s, floor)
    plain = s.mat()
    red = s.mat(color=(1, 0.2, 0.2), shininess=50)
    green = s.mat(color=(0.2, 1, 0.3), reflective=0.2)
    glass = s.mat(color=(0.05, 0.05, 0.15), diffuse=0.3, reflective=0.5, transparency=0.8, ri=1.5)
    # (the CSG nodes keep the identity and the groups below them only translate: a transform on a CSG node puts the
    # normals of its children in the wrong space -- that quirk has a test of its own above)
    s.csg("difference", plain)
    s.shape("cube", red, sg.translation(-1.7, 0.2, 0) @ sg.rotation_y(0.5))
    _mesh_nodes(s, green, verts * 1.35, faces, smooth, sg.translation(-1.7, 0.2, 0))
    s.csg("intersection", plain)
    _mesh_nodes(s, glass, verts * 1.3, faces, smooth, sg.translation(1.7, 0.3, 0.2))
    s.shape("sphere", red, sg.translation(1.9, 0.5, 0.0) @ sg.scaling(1.1, 1.1, 1.1))
    # the same mesh outside any CSG, so that the frame also holds a run the ordinary traversal tests
    s.group(1, plain, sg.translation(0, -0.4, -1.8) @ sg.scaling(0.5, 0.5, 0.5))
    _mesh_nodes(s, green, verts, faces, smooth)
    path = s.write(tmp_path / "csg_mesh.sg")
    _check_frame(rt, oracle, gpu_ctx, path, w, h)
    rng = np.random.default_rng(3)
    rays = np.concatenate([rng.uniform(-3, 3, (256, 3)) + [0, 1.0, -1], rng.normal(size=(256, 3))], axis=1)
    _check_rays(rt, oracle, gpu_ctx, path, rays, remainings=(0, 2, 5))


# ------------------------------------------------------------------ shininess
def test_fractional_and_whole_shininess_in_one_frame(rt, oracle, gpu_ctx, tmp_path):
    """FAST arithmetic raises the specular base to a whole-number shininess by repeated multiplication and calls pow for
    any other; lanes of one wave meet both here (src/material.rs:121-126: powf for every shininess)."""
    w, h = 192, 96
    s = sg.SG()
    s.set_camera(w, h, 0.8, (0, 1.2, -7), (0, 0.2, 0))
    s.light((-3, 6, -8))
    s.light((4, 3, -6), (0.5, 0.5, 0.5))
    floor = s.mat(color=(0.6, 0.6, 0.65), specular=0.8, shininess=12.75)
    _room(s, floor)
    for i, shin in enumerate((10.5, 200.0, 50.25, 7.0, 0.5, 1.0, 333.3, 64.0, 0.0, 2.5)):
        x, y = (i % 5) * 1.1 - 2.2, (i // 5) * 1.1 - 0.45
        m = s.mat(color=(0.2 + 0.08 * i, 0.5, 0.9 - 0.07 * i), specular=0.9, shininess=shin, reflective=0.1 if i % 3 == 0 else 0.0)
        s.shape("sphere", m, sg.translation(x, y, 0.1 * i) @ sg.scaling(0.5, 0.5, 0.5))
    path = s.write(tmp_path / "shininess.sg")
    _check_frame(rt, oracle, gpu_ctx, path, w, h)


# ------------------------------------------------------------------ rays parallel to an axis
def _axis_rays():
    """Rays parallel to an axis (two zero components) from a lattice of origins on all six sides of the scene, and
    rays with one zero component.  The lattice avoids round coordinates: the scenes are built from round numbers, and
    a ray that lies exactly in the plane of a face, or touches a sphere, is decided by the last bit of the ray's
    object-space origin, which the two arithmetic modes round differently."""
    rays = []
    for axis in range(3):
        for sign in (1.0, -1.0):
            d = [0.0, 0.0, 0.0]
            d[axis] = sign
            for a in np.linspace(-2.6, 2.6, 14) + 0.0137:
                for b in np.linspace(-0.9, 2.4, 12) + 0.0071:
                    o = [0.0, 0.0, 0.0]
                    o[axis] = -4.5 * sign
                    if axis == 0:
                        o[1], o[2] = b, a
                    elif axis == 1:
                        o[0], o[2] = a, b - 0.75
                    else:
                        o[0], o[1] = a, b
                    rays.append(o + d)
    for a in np.linspace(-2.0, 2.0, 9) + 0.0137:
        for b in np.linspace(0.05, 1.9, 8) + 0.0071:
            rays.append([a, b, -4.5, 0.0, 0.3, 1.0])
            rays.append([-4.5, b, a, 1.0, 0.0, 0.25])
            rays.append([a, 4.0, b - 1.0, 0.2, -1.0, 0.0])
    return np.array(rays)


@pytest.mark.parametrize("name", ["teapot_lowres_room", "groups", "cover", "csg"])
def test_rays_with_zero_direction_components(rt, oracle, gpu_ctx, name):
    """A direction component that is exactly zero makes the reciprocal infinite; the slab tests of bounds records,
    runs of triangles and CSG boxes must still see a box the ray passes through (origin between the two planes of
    that axis) and skip one it does not.  Axis-parallel rays from a lattice of origins, against scenes whose
    traversal lists hold bounds records, runs of triangles, axis-aligned cubes and CSG objects."""
    rays = _axis_rays()
    ref = _check_rays(rt, oracle, gpu_ctx, sg_path(name), rays, remainings=(0, 2, 4))
    assert (ref.sum(axis=1) > 0).mean() > 0.3


def test_axis_aligned_camera_with_odd_size_on_a_mesh(rt, oracle, gpu_ctx, tmp_path):
    """An axis-aligned camera with odd width and height: the centre column and row have direction components that are
    exactly zero (the case an image shows as a missing line through a mesh that rests on the floor)."""
    verts, faces = sg.icosahedron()
    w, h = 65, 41
    s = sg.SG()
    s.set_camera(w, h, 0.7, (0, 0, -6), (0, 0, 0))
    s.light((-5, 7, -8))
    # (checkers moved off the round numbers: the centre column would otherwise run along a boundary of the pattern)
    floor = s.mat(specular=0, pattern=("checkers", (0.3, 0.3, 0.3), (0.7, 0.7, 0.7), sg.translation(0.37, 0.11, 0.21)))
    s.shape("plane", floor, sg.translation(0, -1.5, 0))
    green = s.mat(color=(0.2, 1, 0.3), reflective=0.2)
    s.group(3, green)
    for dx in (-1.3, 0.0, 1.3):  # three objects of 20 triangles: runs under bounds records
        # (turned: an icosahedron has edges in its planes of symmetry, and a ray along an edge is decided by the last bit)
        s.group(len(faces), green, sg.translation(dx, 0.013, 0) @ sg.rotation_z(0.37) @ sg.rotation_x(0.21))
        for f in faces:
            s.tri(green, *[verts[i] for i in f])
    path = s.write(tmp_path / "axis_cam.sg")
    ref, out = _check_frame(rt, oracle, gpu_ctx, path, w, h)
    assert ref[h // 2, w // 2].sum() > 0.2 and out["fast"][h // 2, w // 2].sum() > 0.2  # the centre pixel looks at the mesh


# ------------------------------------------------------------------ runs of triangles the upload leaves empty
def test_pools_of_triangles_that_leave_empty_runs(rt, oracle, gpu_ctx, tmp_path):
    """Two OBJ groups of 70 triangles under one transform: the host cuts each into runs of 64 + 6, the upload pools the
    140 and cuts them again into three runs by place -- the fourth record keeps no triangle.  It must be passed over
    without a load (its first triangle would lie past the end of the table)."""
    rng = np.random.default_rng(21)
    lines = []
    faces = {"a": [], "b": []}
    nv = 0
    for g in ("a", "b"):
        for i in range(70):
            c = rng.uniform([-30, -30, 2], [30, 30, 30])
            for p_ in c + rng.uniform(-7, 7, (3, 3)):
                lines.append("v %.5f %.5f %.5f" % tuple(p_))
            faces[g].append("f %d %d %d" % (nv + 1, nv + 2, nv + 3))
            nv += 3
    obj = tmp_path / "two_groups.obj"
    obj.write_text("\n".join(lines + ["g a"] + faces["a"] + ["g b"] + faces["b"]) + "\n")
    sc = rt.Scene.builtin("teapot_room", str(obj)).set_resolution(120, 60)
    p = str(tmp_path / "two_groups.sg")
    sc.save_sg(p)
    _check_frame(rt, oracle, gpu_ctx, p, 120, 60)
    rng = np.random.default_rng(2)
    rays = np.concatenate([rng.uniform(-3, 3, (256, 3)) + [0, 1.5, 0], rng.normal(size=(256, 3))], axis=1)
    _check_rays(rt, oracle, gpu_ctx, p, rays, remainings=(1, 4))


def test_rays_in_the_plane_of_a_cube_face(rt, oracle, gpu_ctx, tmp_path):
    """src/shapes/cube.rs:31-49 with a direction component of zero and the origin exactly on that axis's face:
    (+-1 - origin) * INFINITY is 0 * inf = NaN on one side, and f64::max / f64::min over the axes ignore it, so the
    ray still hits.  Cubes with exact transforms (identity, powers of two), so no rounding blurs the case; one is
    axis-aligned, one rotated by a quarter turn about y written exactly."""
    s = sg.SG()
    s.set_camera(16, 8, 0.9, (0, 1.5, -9), (0, 0, 0))
    s.light((-5, 7, -8))
    red = s.mat(color=(1, 0.2, 0.2))
    blue = s.mat(color=(0.2, 0.2, 1), reflective=0.25)
    s.shape("cube", red)
    s.shape("cube", blue, sg.translation(4, 0, 0) @ sg.scaling(0.5, 2, 0.5))
    quarter = np.array([[0, 0, 1, 0], [0, 1, 0, 0], [-1, 0, 0, 0], [0, 0, 0, 1]], dtype=float)
    s.shape("cube", red, sg.translation(-4, 0, 0) @ quarter)
    path = s.write(tmp_path / "faces.sg")
    rays = []
    for cx, hx, hy, hz in ((0, 1, 1, 1), (4, 0.5, 2, 0.5), (-4, 1, 1, 1)):
        for sy in (-1, 1):
            for z in (-0.25, 0.0, 0.375):
                rays.append([cx - 8, sy * hy, z, 1, 0, 0])       # in the plane of the top / bottom face
                rays.append([cx + 0.125, sy * hy, -8, 0, 0, 1])
        for sz in (-1, 1):
            rays.append([cx - 8, 0.25, sz * hz, 1, 0, 0])         # in the plane of the front / back face
            rays.append([cx + 0.25, 8, sz * hz, 0, -1, 0])
        for sx in (-1, 1):
            rays.append([cx + sx * hx, 0.5, -8, 0, 0, 1])         # in the plane of a side face
            rays.append([cx + sx * hx, 8, 0.25, 0, -1, 0])
    ref = _check_rays(rt, oracle, gpu_ctx, path, rays, remainings=(0, 2))
    assert (ref.sum(axis=1) > 0).mean() > 0.9  # the reference does hit


# ------------------------------------------------------------------ executed-work counters
@pytest.mark.parametrize("name", ["cover", "cylinders", "groups", "teapot_lowres_room", "csg"])
def test_counting_kernels_render_the_same_frame_and_count(rt, gpu_ctx, manifest, name):
    """rtamd_count_work switches to the kernels that count the tests their traversals execute.  The frame does not
    change by a bit; the counters are zero without the switch; with it, every kind of leaf the scene has is counted,
    no kind it lacks is, and no traversal tested more than the reference would (every leaf for every ray)."""
    w, h = manifest[name]["small"]
    flat = rt.Scene.from_sg(sg_path(name)).set_resolution(w, h).flatten()
    gpu_ctx.upload(flat)
    kinds = [flat.scene.leaves[i].kind for i in range(flat.scene.n_leaves)]
    fields = ["tests_sphere", "tests_plane", "tests_cube", "tests_slab2", "tests_slab3", "tests_cylinder", "tests_cone", "tests_caps",
              "tests_tri_first", "tests_tri_second", "tests_box", "tests_run_transform"]
    for mode in (rt.MODE_FAST, rt.MODE_STRICT):
        gpu_ctx.reset_stats()
        plain = gpu_ctx.render(w, h, mode)
        st = gpu_ctx.stats()
        assert all(getattr(st, f) == 0 for f in fields)
        try:
            gpu_ctx.count_work(True)
            gpu_ctx.reset_stats()
            counted = gpu_ctx.render(w, h, mode)
            st = gpu_ctx.stats()
        finally:
            gpu_ctx.count_work(False)
        assert np.array_equal(plain, counted)
        c = {f: getattr(st, f) for f in fields}
        print(name, mode, c)
        traversals = st.rays  # container passes included
        per_kind = {0: c["tests_sphere"], 1: c["tests_plane"], 2: c["tests_cube"] + c["tests_slab2"], 3: c["tests_cylinder"],
                    4: c["tests_cone"], 5: c["tests_run_transform"]}
        for kind, n in per_kind.items():
            have = kinds.count(kind)
            assert (n > 0) == (have > 0), (kind, n, have)
            # at most one test per leaf of that kind per traversal (a cube near a face plane is tested both ways)
            assert n <= have * traversals * (2 if kind == 2 else 1)
        assert c["tests_slab3"] <= c["tests_slab2"] and c["tests_tri_second"] <= c["tests_tri_first"]
        if mode == rt.MODE_STRICT:
            assert c["tests_slab2"] == 0
        n_tris = flat.scene.n_tris
        assert c["tests_tri_first"] <= n_tris * traversals and (c["tests_tri_first"] > 0) == (n_tris > 0)
