/* A host in plain C: one frame of a scene described in the flat form of include/rtamd.h, rendered on one GPU through
 * rtamd_render and on a list of GPUs through rtamd_multi_render, the two frames compared.  What a binding in any language
 * does, without a binding (the reference's Camera::render, src/scene.rs:236-248, is the call being replaced).
 *
 *   gcc -std=c99 -Iinclude examples/render_c.c -Lray-tracer_amd/lib -lrtamd -Wl,-rpath,$PWD/ray-tracer_amd/lib -lm -o render_c
 *   ./render_c [width height [device ...]]
 *
 * The scene: a floor, a matte ball and a glass ball under one light; built here so that the example needs no loader. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rtamd.h"

static void identity34(double* m) {
    memset(m, 0, 12 * sizeof(double));
    m[0] = m[5] = m[10] = 1.0;
}

/* world -> object of scaling(s) then translation(t): the inverse, rows 0..2 */
static void inv_scale_translate(double* m, double s, double tx, double ty, double tz) {
    identity34(m);
    m[0] = m[5] = m[10] = 1.0 / s;
    m[3] = -tx / s, m[7] = -ty / s, m[11] = -tz / s;
}

static int check(int rc, const char* what, rtamd_ctx* ctx) {
    if (rc != RTAMD_OK) fprintf(stderr, "%s: error %d: %s\n", what, rc, rtamd_last_error(ctx));
    return rc;
}

int main(int argc, char** argv) {
    const int w = argc > 2 ? atoi(argv[1]) : 320, h = argc > 2 ? atoi(argv[2]) : 180;
    int devices[16], n_devices = 0, i;
    for (i = 3; i < argc && n_devices < 16; i++) devices[n_devices++] = atoi(argv[i]);
    if (n_devices == 0) devices[n_devices++] = 0, devices[n_devices++] = 0; /* the same GPU twice: two ranks on one card */

    /* ---- the scene, flat (include/rtamd.h says what each table holds) */
    rtamd_material mats[3];
    rtamd_leaf leaves[3];
    rtamd_visit visits[3];
    rtamd_light light;
    double chain[3 * 12];
    rtamd_scene s;
    memset(mats, 0, sizeof mats), memset(leaves, 0, sizeof leaves), memset(visits, 0, sizeof visits), memset(&s, 0, sizeof s);
    for (i = 0; i < 3; i++) {
        mats[i].color[0] = mats[i].color[1] = mats[i].color[2] = 1.0;
        mats[i].ambient = 0.1, mats[i].diffuse = 0.9, mats[i].specular = 0.9, mats[i].shininess = 200.0, mats[i].refractive_index = 1.0;
        mats[i].pattern_kind = RTAMD_PAT_NONE;
    }
    mats[0].specular = 0.0, mats[0].reflective = 0.2;                                               /* floor */
    mats[1].color[0] = 0.8, mats[1].color[1] = 0.3, mats[1].color[2] = 0.2;                          /* matte ball */
    mats[2].diffuse = 0.1, mats[2].transparency = 0.9, mats[2].reflective = 0.9, mats[2].refractive_index = 1.5; /* glass ball */
    mats[2].color[0] = mats[2].color[1] = mats[2].color[2] = 0.05;

    leaves[0].kind = RTAMD_PLANE, identity34(leaves[0].w2o);
    leaves[1].kind = RTAMD_SPHERE, inv_scale_translate(leaves[1].w2o, 0.7, -1.1, 0.7, 0.4);
    leaves[2].kind = RTAMD_SPHERE, inv_scale_translate(leaves[2].w2o, 1.0, 0.8, 1.0, -0.3);
    for (i = 0; i < 3; i++) {
        leaves[i].flags = RTAMD_LEAF_SHADOW, leaves[i].material = i, leaves[i].eq_class = i;
        leaves[i].chain_start = i, leaves[i].chain_len = 1; /* one level: the leaf's own inverse */
        leaves[i].ray_start = i, leaves[i].ray_len = 1;     /* ... which is all a ray passes, too (w2o is that matrix) */
        memcpy(chain + 12 * i, leaves[i].w2o, 12 * sizeof(double));
        leaves[i].p0 = -INFINITY, leaves[i].p1 = INFINITY;
        visits[i].unit = i, visits[i].flags = RTAMD_VISIT_FIRST | RTAMD_VISIT_CLASS_END, visits[i].eq_class = i, visits[i].class_material = i;
    }
    light.position[0] = -6.0, light.position[1] = 8.0, light.position[2] = -8.0;
    light.intensity[0] = light.intensity[1] = light.intensity[2] = 1.0;

    /* Camera::new + view_transform(from (0, 1.5, -6) looking down +z), src/scene.rs:172-213: its inverse is a translation */
    {
        const double fov = 1.0, half_view = tan(fov / 2.0), aspect = (double)w / (double)h;
        s.camera.hsize = w, s.camera.vsize = h;
        s.camera.half_width = aspect >= 1.0 ? half_view : half_view * aspect;
        s.camera.half_height = aspect >= 1.0 ? half_view / aspect : half_view;
        s.camera.pixel_size = s.camera.half_width * 2.0 / (double)w;
        identity34(s.camera.inv);
        /* orientation of view_transform for forward = +z, up = +y: x and z mirrored (left = forward x up) */
        s.camera.inv[0] = -1.0, s.camera.inv[10] = -1.0;
        s.camera.inv[3] = 0.0, s.camera.inv[7] = 1.5, s.camera.inv[11] = -6.0;
    }
    s.max_remaining = 4;
    s.n_lights = 1, s.lights = &light;
    s.n_leaves = 3, s.leaves = leaves;
    s.n_chain = 3, s.chain = chain;
    s.n_materials = 3, s.materials = mats;
    s.n_visits = 3, s.visits = visits;
    s.n_classes = 3;

    /* ---- one GPU */
    {
        const size_t bytes = (size_t)w * h * 3 * sizeof(double);
        double* one = (double*)rtamd_host_alloc(bytes);
        double* many = (double*)malloc(bytes);
        unsigned char* many8 = (unsigned char*)malloc((size_t)w * h * 3);
        rtamd_ctx* ctx = NULL;
        rtamd_multi* multi = NULL;
        rtamd_stats st;
        double worst = 0.0;
        size_t k;
        if (!one || !many || !many8) {
            fprintf(stderr, "no memory for the frame (page-locked memory needs a GPU)\n");
            return 2;
        }
        if (check(rtamd_ctx_create(devices[0], &ctx), "rtamd_ctx_create", NULL)) return 1;
        if (check(rtamd_scene_upload(ctx, &s), "rtamd_scene_upload", ctx)) return 1;
        if (check(rtamd_render(ctx, RTAMD_MODE_FAST, one, NULL), "rtamd_render", ctx)) return 1;
        if (check(rtamd_get_stats(ctx, &st), "rtamd_get_stats", ctx)) return 1;
        printf("one GPU: %dx%d, %llu rays, kernel %.3f ms, centre pixel %.4f %.4f %.4f\n", w, h, (unsigned long long)st.rays, st.kernel_ms,
               one[((size_t)(h / 2) * w + w / 2) * 3], one[((size_t)(h / 2) * w + w / 2) * 3 + 1], one[((size_t)(h / 2) * w + w / 2) * 3 + 2]);

        /* ---- the same frame from a list of GPUs, assembled on the first */
        if (rtamd_multi_create(devices, n_devices, &multi) != RTAMD_OK) {
            fprintf(stderr, "rtamd_multi_create: %s\n", rtamd_multi_last_error(multi));
            return 1;
        }
        if (rtamd_multi_upload(multi, &s) != RTAMD_OK || rtamd_multi_render(multi, RTAMD_MODE_FAST, 8, 1, 1, many, many8, RTAMD_WANT_F64 | RTAMD_WANT_U8) != RTAMD_OK) {
            fprintf(stderr, "rtamd_multi: %s\n", rtamd_multi_last_error(multi));
            return 1;
        }
        for (k = 0; k < (size_t)w * h * 3; k++) {
            const double d = fabs(one[k] - many[k]);
            if (!(d <= worst)) worst = d;
        }
        printf("%d ranks: largest difference from the one-GPU frame %.3g (must be 0)\n", n_devices, worst);
        rtamd_multi_destroy(multi);
        rtamd_ctx_destroy(ctx);
        rtamd_host_free(one), free(many), free(many8);
        return worst == 0.0 ? 0 : 3;
    }
}
