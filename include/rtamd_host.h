/*
 * rtamd_host.h -- C ABI of the host library (librtamd_host.so): the scene side of Camera::render.
 *
 * include/rtamd.h takes a scene that is already flat (`rtamd_scene`).  This header is the step before it, for a host
 * that holds the reference's own object tree -- World {objects: Vec<Box<dyn Shape>>, lights} (src/scene.rs:6-9), every
 * shape with Props {transform, material, parent_transforms, shadow} (src/shape.rs:5-12), Group {children}
 * (src/shapes/group.rs:8-11), Csg {op, left, right} (src/shapes/csg.rs:5-15) -- and wants the flattening done for it:
 * walk the tree once, call one rth_add_* / rth_*_begin / rth_*_end per node, then rth_flatten.  What flattening
 * means (mesh runs, composed inverses by the reference's cofactor routine, value-equality classes of
 * src/shape.rs:23-27,119-123, `left.includes` masks of src/shapes/csg.rs:64, the class-major visit list) is done
 * once, here, for every host language.
 *
 * Conventions: plain pointers and sizes; a 4x4 matrix is 16 doubles row-major, as the reference's
 * `Matrix4x4([[f64; 4]; 4])` lies in memory (src/matrix.rs:68); a Tuple is 4 doubles x, y, z, w (src/tuple.rs:7-13);
 * `material` may be NULL for `Material::new()` (src/material.rs:20-33), `transform` may be NULL for the identity;
 * functions that return int give 0 or -1 with the reason in rth_last_error() (thread-local); nothing unwinds or
 * aborts across the boundary.  A singular transform is reported by rth_flatten (the reference panics in
 * `inverse().unwrap()`, src/shape.rs:40).
 */
#ifndef RTAMD_HOST_H
#define RTAMD_HOST_H

#include "rtamd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Reason of the last failure on the calling thread. */
const char* rth_last_error(void);

/* ---- Material (src/material.rs:8-18) and its optional Pattern (src/pattern.rs:4-16). */
enum { RTH_PAT_NONE = 0, RTH_PAT_STRIPE = 1, RTH_PAT_GRADIENT = 2, RTH_PAT_RING = 3, RTH_PAT_CHECKERS = 4, RTH_PAT_TEST = 5 };
typedef struct rth_material {
    double color[3];
    double ambient, diffuse, specular, shininess, reflective, transparency, refractive_index;
    int32_t pattern; /* RTH_PAT_*; the three fields below are read only when it is not RTH_PAT_NONE */
    int32_t pad_;
    double pat_a[3], pat_b[3];
    double pat_transform[16]; /* Pattern::transform, not inverted */
} rth_material;
/* Material::new(): white, ambient 0.1, diffuse 0.9, specular 0.9, shininess 200, the rest 0, refractive index 1, no
 * pattern (identity pattern transform). */
void rth_material_default(rth_material* out);

/* ---- World under construction (src/scene.rs:6-9 plus the Camera that renders it, :171-179). */
typedef struct rth_world rth_world;
rth_world* rth_world_new(void); /* NULL: out of memory */
void rth_world_free(rth_world* w);

/* One call per leaf shape, in the order of `objects` / `children` (the order decides ties in t, src/scene.rs:19-26).
 * A shape lands in the group or CSG node that is open, or in World::objects when none is.
 * shadow: Props::shadow (src/shape.rs:9). */
int rth_add_sphere(rth_world* w, const double* transform, const rth_material* material, int shadow);   /* src/shapes/sphere.rs */
int rth_add_plane(rth_world* w, const double* transform, const rth_material* material, int shadow);    /* src/shapes/plane.rs */
int rth_add_cube(rth_world* w, const double* transform, const rth_material* material, int shadow);     /* src/shapes/cube.rs */
/* min, max, closed: src/shapes/cylinder.rs:6-11 / src/shapes/cone.rs:6-11 (infinities for an unbounded one) */
int rth_add_cylinder(rth_world* w, const double* transform, const rth_material* material, int shadow, double min, double max, int closed);
int rth_add_cone(rth_world* w, const double* transform, const rth_material* material, int shadow, double min, double max, int closed);
/* points: p1, p2, p3 as three Tuples (12 doubles); Triangle::new, src/shapes/triangle.rs:22-35 */
int rth_add_triangle(rth_world* w, const double* transform, const rth_material* material, int shadow, const double* points);
/* normals: n1, n2, n3 as three Tuples (12 doubles); Triangle::smooth, src/shapes/triangle.rs:37-49 */
int rth_add_smooth_triangle(rth_world* w, const double* transform, const rth_material* material, int shadow, const double* points,
                            const double* normals);

/* Group (src/shapes/group.rs): the shapes added between begin and end are its children.  Its material is what
 * refresh_parents hands to children that still have Material::new() (:76-89). */
int rth_group_begin(rth_world* w, const double* transform, const rth_material* material, int shadow);
int rth_group_end(rth_world* w);
/* Csg (src/shapes/csg.rs:5-15): exactly two shapes (leaves, groups or CSG nodes) between begin and end, left then right.
 * op: RTAMD_CSG_UNION / _INTERSECTION / _DIFFERENCE. */
int rth_csg_begin(rth_world* w, int op, const double* transform, const rth_material* material, int shadow);
int rth_csg_end(rth_world* w);

/* PointLight::new(position, intensity), src/ray.rs:27-37; lights keep the order they are added in (src/scene.rs:28-51). */
int rth_add_light(rth_world* w, const double position[3], const double intensity[3]);
/* Camera::new(hsize, vsize, field_of_view) with `transform` (src/scene.rs:171-209); may be called again to replace it. */
int rth_set_camera(rth_world* w, int hsize, int vsize, double field_of_view, const double* transform);

/* The scene the calls above built (World + Camera), as the handle the functions below take.  Borrowed: it lives as
 * long as `w` and must not be passed to rth_free_scene.  NULL (and an error) while a group or CSG node is still open. */
void* rth_world_scene(rth_world* w);

/* ---- Scenes from files: what src/bin/yaml_renderer.rs:366-408 and src/obj.rs:16-135 read, and this library's own
 * loss-free text dump (".sg").  The handle is owned by the caller: rth_free_scene. */
void* rth_load_yaml(const char* path);
void* rth_load_yaml_str(const char* text);
void* rth_load_sg(const char* path);
/* "reflect_refract" | "teapot_room" | "teapot" | "csg" | "default_world": the scenes src/bin/{reflect_refract,teapot,csg}.rs
 * and `impl Default for World` (src/scene.rs:139-155) build in code; arg: path of the OBJ file where one is needed. */
void* rth_builtin_scene(const char* name, const char* arg);
int rth_save_sg(void* scene, const char* path);
void rth_free_scene(void* scene);
int rth_scene_info(void* scene, int* hsize, int* vsize, int* n_objects, int* n_lights);
/* Camera::new(hsize, vsize, same field of view) with the same transform. */
int rth_set_resolution(void* scene, int hsize, int vsize);

/* ---- World + Camera -> rtamd_scene.  What Camera::render does before its pixel loop (world.refresh_parents(),
 * src/scene.rs:237) and what a device needs in its place: see the top of this file.  max_remaining: 4 in the
 * reference (src/scene.rs:243).  NULL on failure.  The view stays valid until rth_free_flat. */
void* rth_flatten(void* scene, int max_remaining);
const rtamd_scene* rth_flat_view(void* flat);
void rth_free_flat(void* flat);

/* ---- Camera::render (src/scene.rs:236-248) through the GPU library, on the device RTAMD_DEVICE names (default 0).
 * rgb: hsize * vsize * 3 doubles (Canvas.pixels), rgb8: the 8-bit channels of src/tuple.rs:226-237; either may be NULL
 * (not both).  With rgb NULL only 3 bytes per pixel cross the PCIe link.  stats may be NULL. */
int rth_render(void* scene, int mode, double* rgb, uint8_t* rgb8, rtamd_stats* stats);

/* ---- Canvas output (src/tuple.rs:343-390, src/bin/yaml_renderer.rs:419-429) on caller pixels (w * h * 3 doubles). */
int rth_canvas_save(const double* rgb, int w, int h, const char* path, int as_ppm);
int rth_canvas_quantize(const double* rgb, int64_t n_pixels, uint8_t* out);
/* PNG from 8-bit channels (w * h * 3 bytes) as they come from the device. */
int rth_save_png8(const uint8_t* rgb8, int w, int h, const char* path);
char* rth_canvas_ppm(const double* rgb, int w, int h); /* malloc'ed P3 text; rth_free_str */
void rth_free_str(char* s);

/* ---- ObjParser facts (src/obj.rs), for tests of the loader. */
int rth_obj_info(const char* text, int* n_vertices, int* n_normals, int* n_default_tris, int* n_groups, int* n_group_tris,
                 int* lines_ignored, int* n_smooth);
int rth_obj_triangle(const char* text, int k, double* out21);

#ifdef __cplusplus
}
#endif
#endif /* RTAMD_HOST_H */
