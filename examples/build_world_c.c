/* A host in plain C99 that holds an object tree of its own and hands it to the GPU through the scene builder of
 * include/rtamd_host.h -- the calls a Rust `World` walker makes through `extern "C"`: one rth_add_* per leaf,
 * rth_csg_begin / rth_csg_end around the two children of a Csg (src/shapes/csg.rs:5-15), rth_add_light, rth_set_camera,
 * then rth_flatten -> rtamd_scene_upload -> rtamd_render.
 *
 * The scene is the CSG scene of the benchmark configurations: src/bin/csg.rs:5-91 (a red cube minus a green capped
 * cylinder, a green "holder" cylinder, same camera and light) inside the room of src/bin/reflect_refract.rs:20-69, plus a
 * glass lens Intersection(Sphere, Cube).  The library builds the same scene itself (rth_builtin_scene("csg")); this
 * program checks that what it built call by call flattens to the same bytes, renders it, and writes the frame.
 *
 *   build_world_c WIDTH HEIGHT OUT.f64 [DEVICE]     (OUT.f64: WIDTH * HEIGHT * 3 doubles, row-major; "-" renders nothing)
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rtamd_host.h"

typedef struct { double m[16]; } M4; /* row-major, as the reference's Matrix4x4([[f64; 4]; 4]) */

static M4 identity(void) {
    M4 r;
    memset(&r, 0, sizeof r);
    r.m[0] = r.m[5] = r.m[10] = r.m[15] = 1.0;
    return r;
}
static M4 mul(M4 a, M4 b) { /* src/matrix.rs:296-311 */
    M4 r;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++)
            r.m[4 * i + j] = a.m[4 * i] * b.m[j] + a.m[4 * i + 1] * b.m[4 + j] + a.m[4 * i + 2] * b.m[8 + j] + a.m[4 * i + 3] * b.m[12 + j];
    return r;
}
/* the fluent builders pre-multiply (src/matrix.rs:153-171): id().scale(..).rotate_x(..) = rotation_x * scaling */
static M4 translate(M4 t, double x, double y, double z) {
    M4 r = identity();
    r.m[3] = x, r.m[7] = y, r.m[11] = z;
    return mul(r, t);
}
static M4 scale(M4 t, double x, double y, double z) {
    M4 r = identity();
    r.m[0] = x, r.m[5] = y, r.m[10] = z;
    return mul(r, t);
}
static M4 rotate_x(M4 t, double a) {
    M4 r = identity();
    r.m[5] = cos(a), r.m[6] = -sin(a), r.m[9] = sin(a), r.m[10] = cos(a);
    return mul(r, t);
}
static M4 rotate_y(M4 t, double a) {
    M4 r = identity();
    r.m[0] = cos(a), r.m[2] = sin(a), r.m[8] = -sin(a), r.m[10] = cos(a);
    return mul(r, t);
}
static M4 rotate_z(M4 t, double a) {
    M4 r = identity();
    r.m[0] = cos(a), r.m[1] = -sin(a), r.m[4] = sin(a), r.m[5] = cos(a);
    return mul(r, t);
}
/* src/scene.rs:157-169 */
static M4 view_transform(const double from[3], const double to[3], const double up[3]) {
    double f[3], u[3], l[3], t[3], n;
    for (int k = 0; k < 3; k++) f[k] = to[k] - from[k];
    n = sqrt(f[0] * f[0] + f[1] * f[1] + f[2] * f[2] + 0.0 * 0.0);
    for (int k = 0; k < 3; k++) f[k] /= n;
    n = sqrt(up[0] * up[0] + up[1] * up[1] + up[2] * up[2] + 0.0 * 0.0);
    for (int k = 0; k < 3; k++) u[k] = up[k] / n;
    l[0] = f[1] * u[2] - f[2] * u[1], l[1] = f[2] * u[0] - f[0] * u[2], l[2] = f[0] * u[1] - f[1] * u[0];
    t[0] = l[1] * f[2] - l[2] * f[1], t[1] = l[2] * f[0] - l[0] * f[2], t[2] = l[0] * f[1] - l[1] * f[0];
    M4 o = identity(), tr = identity();
    o.m[0] = l[0], o.m[1] = l[1], o.m[2] = l[2];
    o.m[4] = t[0], o.m[5] = t[1], o.m[6] = t[2];
    o.m[8] = -f[0], o.m[9] = -f[1], o.m[10] = -f[2];
    tr.m[3] = -from[0], tr.m[7] = -from[1], tr.m[11] = -from[2];
    return mul(o, tr);
}

static int failures = 0;
#define CHECK(call)                                                        \
    do {                                                                   \
        if ((call) != 0) {                                                 \
            fprintf(stderr, "%s failed: %s\n", #call, rth_last_error());   \
            failures++;                                                    \
        }                                                                  \
    } while (0)

static rth_material coloured(double r, double g, double b) {
    rth_material m;
    rth_material_default(&m);
    m.color[0] = r, m.color[1] = g, m.color[2] = b;
    return m;
}

static void add_room(rth_world* w) { /* src/bin/reflect_refract.rs:10-69 */
    rth_material floor_m, ceil_m, wall_m;
    M4 t;
    rth_material_default(&floor_m);
    floor_m.pattern = RTH_PAT_CHECKERS;
    floor_m.pat_a[0] = floor_m.pat_a[1] = floor_m.pat_a[2] = 0.35;
    floor_m.pat_b[0] = floor_m.pat_b[1] = floor_m.pat_b[2] = 0.65;
    floor_m.specular = 0.0, floor_m.reflective = 0.4;
    t = rotate_y(identity(), 0.31415);
    CHECK(rth_add_plane(w, t.m, &floor_m, 1));
    ceil_m = coloured(0.8, 0.8, 0.8);
    ceil_m.ambient = 0.3, ceil_m.specular = 0.0;
    t = translate(identity(), 0, 5, 0);
    CHECK(rth_add_plane(w, t.m, &ceil_m, 1));
    rth_material_default(&wall_m);
    wall_m.pattern = RTH_PAT_STRIPE;
    wall_m.pat_a[0] = wall_m.pat_a[1] = wall_m.pat_a[2] = 0.45;
    wall_m.pat_b[0] = wall_m.pat_b[1] = wall_m.pat_b[2] = 0.55;
    {
        M4 s = identity(), r = identity();
        s.m[0] = s.m[5] = s.m[10] = 0.25;
        r.m[0] = cos(1.5708), r.m[2] = sin(1.5708), r.m[8] = -sin(1.5708), r.m[10] = cos(1.5708);
        t = mul(s, r); /* scaling * rotation_y, src/bin/reflect_refract.rs:13 */
        memcpy(wall_m.pat_transform, t.m, sizeof t.m);
    }
    wall_m.ambient = 0.0, wall_m.diffuse = 0.4, wall_m.specular = 0.0, wall_m.reflective = 0.3;
    t = translate(rotate_z(rotate_y(identity(), 1.5708), 1.5708), -5, 0, 0);
    CHECK(rth_add_plane(w, t.m, &wall_m, 1));
    t = translate(rotate_z(rotate_y(identity(), 1.5708), 1.5708), 5, 0, 0);
    CHECK(rth_add_plane(w, t.m, &wall_m, 1));
    t = translate(rotate_x(identity(), 1.5708), 0, 0, 5);
    CHECK(rth_add_plane(w, t.m, &wall_m, 1));
    t = translate(rotate_x(identity(), 1.5708), 0, 0, -5);
    CHECK(rth_add_plane(w, t.m, &wall_m, 1));
}

static int same_bytes(const char* what, const void* a, const void* b, size_t n) {
    if (n == 0 || memcmp(a, b, n) == 0) return 1;
    printf("flat scene: %s differs\n", what);
    return 0;
}

int main(int argc, char** argv) {
    if (argc < 4) {
        fprintf(stderr, "usage: %s WIDTH HEIGHT OUT.f64|- [DEVICE]\n", argv[0]);
        return 2;
    }
    const int width = atoi(argv[1]), height = atoi(argv[2]);
    const char* out_path = argv[3];
    const int device = argc > 4 ? atoi(argv[4]) : 0;
    const double PI = 3.14159265358979323846;

    rth_world* w = rth_world_new();
    if (!w) return 1;
    {
        const double from[3] = {-2.6, 1.5, -3.9}, to[3] = {-0.6, 1.0, -0.8}, up[3] = {0, 1, 0};
        const double at[3] = {-4.9, 4.9, -1.0}, white[3] = {1, 1, 1};
        M4 view = view_transform(from, to, up);
        CHECK(rth_set_camera(w, width, height, 1.152, view.m));
        CHECK(rth_add_light(w, at, white));
    }
    add_room(w);
    {
        rth_material red = coloured(1, 0, 0), green = coloured(0, 1, 0), glass = coloured(0.05, 0.05, 0.2);
        M4 t;
        /* Csg::difference(cube, cylinder), src/bin/csg.rs */
        CHECK(rth_csg_begin(w, RTAMD_CSG_DIFFERENCE, NULL, NULL, 1));
        CHECK(rth_add_cube(w, NULL, &red, 1));
        CHECK(rth_add_cylinder(w, NULL, &green, 1, -2.0, 2.0, 1));
        CHECK(rth_csg_end(w));
        t = rotate_x(scale(identity(), 0.5, 0.5, 0.5), PI / 2.0);
        CHECK(rth_add_cylinder(w, t.m, &green, 1, -4.0, 4.0, 1));
        /* the glass lens: Csg::intersection(sphere, cube) */
        glass.ambient = 0.0, glass.diffuse = 0.4, glass.specular = 0.9, glass.shininess = 300.0;
        glass.reflective = 0.9, glass.transparency = 0.9, glass.refractive_index = 1.5;
        CHECK(rth_csg_begin(w, RTAMD_CSG_INTERSECTION, NULL, NULL, 1));
        t = translate(identity(), -2.2, 1.0, -1.0);
        CHECK(rth_add_sphere(w, t.m, &glass, 1));
        t = translate(rotate_y(scale(identity(), 0.7, 0.7, 0.7), 0.5), -2.2, 1.0, -1.0);
        CHECK(rth_add_cube(w, t.m, &glass, 1));
        CHECK(rth_csg_end(w));
    }
    if (failures) return 1;

    void* scene = rth_world_scene(w);
    void* flat = scene ? rth_flatten(scene, 4) : NULL;
    if (!flat) {
        fprintf(stderr, "flatten: %s\n", rth_last_error());
        return 1;
    }
    const rtamd_scene* mine = rth_flat_view(flat);

    /* the library's own construction of the same scene, at the same size */
    void* ref_scene = rth_builtin_scene("csg", NULL);
    if (!ref_scene || rth_set_resolution(ref_scene, width, height) != 0) {
        fprintf(stderr, "builtin scene: %s\n", rth_last_error());
        return 1;
    }
    void* ref_flat = rth_flatten(ref_scene, 4);
    if (!ref_flat) return 1;
    const rtamd_scene* ref = rth_flat_view(ref_flat);
    int same = mine->n_leaves == ref->n_leaves && mine->n_chain == ref->n_chain && mine->n_materials == ref->n_materials &&
               mine->n_csg_nodes == ref->n_csg_nodes && mine->n_csg_roots == ref->n_csg_roots && mine->n_visits == ref->n_visits &&
               mine->n_classes == ref->n_classes && mine->n_lights == ref->n_lights && mine->n_tris == ref->n_tris;
    if (same) {
        same &= same_bytes("camera", &mine->camera, &ref->camera, sizeof mine->camera);
        same &= same_bytes("lights", mine->lights, ref->lights, (size_t)mine->n_lights * sizeof(rtamd_light));
        same &= same_bytes("leaves", mine->leaves, ref->leaves, (size_t)mine->n_leaves * sizeof(rtamd_leaf));
        same &= same_bytes("chain", mine->chain, ref->chain, (size_t)mine->n_chain * 12 * sizeof(double));
        same &= same_bytes("materials", mine->materials, ref->materials, (size_t)mine->n_materials * sizeof(rtamd_material));
        same &= same_bytes("csg nodes", mine->csg_nodes, ref->csg_nodes, (size_t)mine->n_csg_nodes * sizeof(rtamd_csg_node));
        same &= same_bytes("csg roots", mine->csg_roots, ref->csg_roots, (size_t)mine->n_csg_roots * sizeof(rtamd_csg_root));
        same &= same_bytes("visits", mine->visits, ref->visits, (size_t)mine->n_visits * sizeof(rtamd_visit));
    }
    printf("built call by call: %d leaves, %d materials, %d csg nodes in %d objects, %d visits, %d classes\n", mine->n_leaves, mine->n_materials,
           mine->n_csg_nodes, mine->n_csg_roots, mine->n_visits, mine->n_classes);
    printf("flat scene equals the library's own csg scene: %s\n", same ? "yes" : "NO");
    int rc = same ? 0 : 1;

    if (strcmp(out_path, "-") != 0) {
        rtamd_ctx* ctx = NULL;
        const size_t n = (size_t)width * (size_t)height * 3;
        double* rgb = (double*)malloc(n * sizeof(double));
        if (!rgb) return 1;
        if (rtamd_ctx_create(device, &ctx) != RTAMD_OK) {
            fprintf(stderr, "rtamd_ctx_create: %s\n", rtamd_last_error(NULL));
            return 1;
        }
        if (rtamd_scene_upload(ctx, mine) != RTAMD_OK || rtamd_render(ctx, RTAMD_MODE_STRICT, rgb, NULL) != RTAMD_OK) {
            fprintf(stderr, "render: %s\n", rtamd_last_error(ctx));
            return 1;
        }
        FILE* f = fopen(out_path, "wb");
        if (!f || fwrite(rgb, sizeof(double), n, f) != n) {
            fprintf(stderr, "cannot write %s\n", out_path);
            return 1;
        }
        fclose(f);
        rtamd_stats st;
        if (rtamd_get_stats(ctx, &st) == RTAMD_OK)
            printf("rendered %dx%d in strict arithmetic: %llu rays, kernel %.3f ms\n", width, height, (unsigned long long)st.rays, st.kernel_ms);
        rtamd_ctx_destroy(ctx);
        free(rgb);
    }
    rth_free_flat(ref_flat);
    rth_free_scene(ref_scene);
    rth_free_flat(flat);
    rth_world_free(w);
    return rc;
}
