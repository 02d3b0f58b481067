/*
 * rtamd.h -- C ABI of the MI355X (gfx950) render path.
 *
 * The reference (joshuabenuck/ray-tracer, Rust) has no FFI or plugin interface; its de-facto
 * boundary for the per-pixel render loop is one method,
 *     pub fn render(&self, world: &mut World) -> Canvas        (src/scene.rs:236-248)
 * with inputs Camera {hsize, vsize, field_of_view, transform, pixel_size, half_width,
 * half_height} (src/scene.rs:171-179), World {objects, lights} (src/scene.rs:6-9) and output
 * Canvas {width, height, pixels: Vec<Color>} (src/tuple.rs:309-313).  This header is what a
 * Rust `extern "C"` block (or any other host) binds instead: the host flattens World + Camera
 * once into the plain-old-data `rtamd_scene` below, uploads it, and asks for pixels.
 *
 * Conventions: every pointer in `rtamd_scene` is host memory owned by the caller and is copied
 * by rtamd_scene_upload; functions return 0 or a negative RTAMD_E_* code and never unwind or
 * abort across the boundary (the reference panics on a singular transform, src/shape.rs:40 --
 * here the flattening host reports it before upload); calls are blocking unless stated; one
 * context drives one GPU, distinct contexts are independent.
 */
#ifndef RTAMD_H
#define RTAMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RTAMD_OK 0
#define RTAMD_E_INVALID -1   /* bad argument or malformed scene (message in rtamd_last_error) */
#define RTAMD_E_HIP -2       /* a HIP runtime call failed */
#define RTAMD_E_NO_SCENE -3  /* render requested before a scene was uploaded */
#define RTAMD_E_LIMIT -4     /* scene exceeds a documented limit of the device path */
#define RTAMD_E_OVERFLOW -5  /* a per-lane CSG intersection list overflowed during the last render */

/* Leaf kinds (src/shapes/{sphere,plane,cube,cylinder,cone,triangle}.rs). A MESH leaf is a run of
 * consecutive (DFS order) triangles that share transform chain, material, shadow flag and
 * flat/smooth kind; every triangle of it keeps its own place in the DFS order. */
enum { RTAMD_SPHERE = 0, RTAMD_PLANE = 1, RTAMD_CUBE = 2, RTAMD_CYLINDER = 3, RTAMD_CONE = 4, RTAMD_MESH = 5 };
/* Pattern kinds (src/pattern.rs:4-10). */
enum { RTAMD_PAT_NONE = 0, RTAMD_PAT_STRIPE, RTAMD_PAT_GRADIENT, RTAMD_PAT_RING, RTAMD_PAT_CHECKERS, RTAMD_PAT_TEST };
/* CSG operations (src/shapes/csg.rs:5-9). */
enum { RTAMD_CSG_UNION = 0, RTAMD_CSG_INTERSECTION = 1, RTAMD_CSG_DIFFERENCE = 2 };

#define RTAMD_LEAF_CLOSED 1u  /* cylinder/cone `closed` */
#define RTAMD_LEAF_SHADOW 2u  /* Props::shadow (src/shape.rs:9): leaf takes part in is_shadowed */
#define RTAMD_LEAF_SMOOTH 4u  /* MESH of Normal::Smooth triangles (src/shapes/triangle.rs:5-8) */

#define RTAMD_MAX_LIGHTS 16
#define RTAMD_MAX_CHAIN 8           /* transform levels (ancestors + own) of one leaf */
#define RTAMD_MAX_RAY_LEVELS 80     /* ray_len of one leaf: groups, the CSG objects of one nest (at most 63), its own */
#define RTAMD_MAX_CSG_LEAVES 64     /* leaves below one top-most CSG node */
#define RTAMD_CSG_LIST_CAP 24       /* per-lane intersection list of one top-most CSG node */
#define RTAMD_MAX_MESH_TRIS 4096    /* triangles in one MESH leaf; the flattening host splits longer runs */
#define RTAMD_MAX_REMAINING 8       /* recursion budget supported (the reference renders with 4) */

/* One primitive leaf, in depth-first order of the object tree (= the order in which the
 * reference's stable sorts keep intersections of equal t, src/scene.rs:19-26,
 * src/shapes/group.rs:37-45). 152 bytes. */
typedef struct rtamd_leaf {
    double w2o[12];      /* rows 0..2 of inverse(own) * inverse(parent_n) * ... * inverse(parent_1):
                            world ray -> object ray (src/shape.rs:39-42 applied through every group
                            and CSG level; the direction is not re-normalised, so t is world-scale) */
    double p0, p1;       /* cylinder/cone min, max (src/shapes/cylinder.rs:6-11) */
    int32_t kind;        /* RTAMD_SPHERE ... RTAMD_MESH */
    uint32_t flags;      /* RTAMD_LEAF_* */
    int32_t material;    /* index into materials (after Group::refresh_parents inheritance) */
    int32_t eq_class;    /* 0 .. n_classes-1: which set of leaves that are value-equal under `PartialEq for
                            dyn Shape` (src/shape.rs:23-27,119-123) this one belongs to; classes are numbered
                            by first appearance in DFS order */
    int32_t chain_start; /* first of chain_len 3x4 matrices in `chain`: inverse of each entry of the
                            leaf's parent_transforms (root first), then inverse of its own transform
                            (src/shape.rs:49-70). Leaves below a CSG node have only their own. */
    int32_t chain_len;
    int32_t tri_start;   /* MESH: first triangle in tri_geom / tri_normals */
    int32_t tri_count;   /* MESH: number of triangles; 0 otherwise */
    int32_t ray_start;   /* first of ray_len 3x4 matrices in `chain`: what the reference multiplies a ray by on its way
                            to this leaf, one level after the other -- the inverse of every group and CSG object above
                            it, root first, then its own (src/shape.rs:39-42 called from src/shapes/group.rs:37-45 and
                            src/shapes/csg.rs:90-97).  For a leaf with groups alone above it that is its chain_start /
                            chain_len.  RTAMD_MODE_STRICT transforms rays level by level like the reference (w2o, the
                            product of the levels, differs from that in the last bits); RTAMD_MODE_FAST uses w2o. */
    int32_t ray_len;     /* at most RTAMD_MAX_RAY_LEVELS; 0 or 1: rays are transformed by w2o in both modes */
} rtamd_leaf;

/* 232 bytes. */
typedef struct rtamd_material {
    double color[3];
    double ambient, diffuse, specular, shininess, reflective, transparency, refractive_index;
    double pat_a[3], pat_b[3];
    double pat_inv[12]; /* rows 0..2 of inverse(pattern.transform) (src/pattern.rs:57-61) */
    int32_t pattern_kind;
    int32_t pad_;
} rtamd_material;

/* One CSG node of a top-most CSG object, nodes in post-order. Leaves of the left child are
 * [leaf_lo, leaf_mid), of the right child [leaf_mid, leaf_hi) (global leaf indices).
 * Bit (leaf - leaf_lo) of left_mask says `left.includes(leaf)` (src/shapes/csg.rs:64), which by
 * value equality can also be true for a leaf of the right child. */
typedef struct rtamd_csg_node {
    uint64_t left_mask;
    int32_t op;
    int32_t leaf_lo, leaf_mid, leaf_hi;
    int32_t pad_;
} rtamd_csg_node;

/* A top-most CSG object: csg_nodes[node_start .. node_start+node_count) and leaves [leaf_lo, leaf_hi). */
typedef struct rtamd_csg_root {
    int32_t node_start, node_count;
    int32_t leaf_lo, leaf_hi;
} rtamd_csg_root;

#define RTAMD_VISIT_CSG 1u        /* unit indexes csg_roots, else leaves */
#define RTAMD_VISIT_FIRST 2u      /* first visit of this unit: the only one closest-hit and shadow rays use */
#define RTAMD_VISIT_CLASS_END 4u  /* last visit of its equality class */

/* Traversal order of the device: units grouped by equality class (the n1/n2 container walk of
 * src/intersection.rs:85-112 is evaluated per class). A plain leaf appears once; a CSG root
 * appears once per class present among its leaves. */
typedef struct rtamd_visit {
    int32_t unit;
    int32_t eq_class;
    uint32_t flags;
    int32_t class_material; /* material of the class (refractive index source) */
} rtamd_visit;

typedef struct rtamd_camera {
    int32_t hsize, vsize;
    double pixel_size, half_width, half_height; /* Camera::new, src/scene.rs:182-209 */
    double inv[12];                             /* rows 0..2 of inverse(camera.transform) */
} rtamd_camera;

typedef struct rtamd_light {
    double position[3];
    double intensity[3];
} rtamd_light;

typedef struct rtamd_scene {
    rtamd_camera camera;
    int32_t max_remaining; /* 4 in the reference (src/scene.rs:243) */
    int32_t n_lights;
    const rtamd_light* lights;
    int32_t n_leaves;
    const rtamd_leaf* leaves;
    int32_t n_chain;
    const double* chain; /* n_chain x 12 */
    int32_t n_tris;
    const double* tri_geom;    /* n_tris x 9: p1, e1 = p2-p1, e2 = p3-p1 (src/shapes/triangle.rs:22-49) */
    const double* tri_normals; /* n_tris x 9: flat: normal,0,0 ; smooth: n1, n2, n3 */
    int32_t n_materials;
    const rtamd_material* materials;
    int32_t n_csg_nodes;
    const rtamd_csg_node* csg_nodes;
    int32_t n_csg_roots;
    const rtamd_csg_root* csg_roots;
    int32_t n_visits;
    const rtamd_visit* visits;
    int32_t n_classes;
} rtamd_scene;

/* Which rows of the frame a render call produces: rows are cut into tiles of `tile_rows`
 * consecutive rows, tile i belongs to part (i % n_parts); this call renders the `span` parts
 * part .. part + span - 1 (span 0 means 1), so that callers can hold unequal shares of the frame and
 * still keep the shares interleaved (the cost of a row varies a lot over a frame).
 * Output is compact: the selected rows in increasing order, hsize pixels each.
 * {tile_rows, 0, 1, 0} = the whole frame. */
typedef struct rtamd_rows {
    int32_t tile_rows, part, n_parts, span;
} rtamd_rows;

#define RTAMD_MODE_STRICT 0 /* no fused multiply-add, IEEE divide, rays through groups and CSG objects one level at a time: the
                               reference's f64 operations in its order (measured against a CPU restatement of it: equal to the
                               last bit of pow, at most 5e-16 per channel) */
#define RTAMD_MODE_FAST 1   /* contraction to FMA, reciprocal-multiply and one matrix per leaf allowed: results differ in the last
                               bits (measured: below 1e-10 per channel, except where a ray grazes an edge: there up to 1e-8,
                               and on either side of the edge in a few pixels per million); the 8-bit channels are the same */

typedef struct rtamd_stats {
    uint64_t rays;          /* World::intersect-equivalent traversals executed (primary, reflection,
                               refraction, shadow, and the n1/n2 container pass) */
    uint64_t primary_rays, shadow_rays, reflect_rays, refract_rays, container_passes;
    uint64_t shade_points;  /* shade_hit-equivalents */
    uint64_t light_samples; /* lighting() evaluations */
    uint64_t csg_overflows; /* lanes whose CSG list overflowed (results then invalid) */
    double kernel_ms;       /* duration of the last render kernel, from HIP events on its stream */
    /* Work the traversals executed, in lane-tests: one ray against one leaf, box or triangle.  The reference tests
     * every leaf for every ray (src/scene.rs:19-26); the device passes over leaves no ray of a wave can meet, so these
     * are what an achieved-throughput figure has to be built from. */
    uint64_t tests_sphere, tests_plane;
    uint64_t tests_cube;       /* the reference's slab formula, src/shapes/cube.rs:31-66 */
    uint64_t tests_slab2;      /* axis-aligned cube as a world-space box (fast arithmetic): first two slabs */
    uint64_t tests_slab3;      /* ... third slab + hit bookkeeping, only for rays the first two did not reject */
    uint64_t tests_cylinder, tests_cone;
    uint64_t tests_caps;       /* end caps of closed cylinders / cones */
    uint64_t tests_tri_first;  /* Moeller-Trumbore up to u (src/shapes/triangle.rs:57-68) */
    uint64_t tests_tri_second; /* ... v and t (:69-80), only for rays the first half did not reject */
    uint64_t tests_box;        /* bounds of leaf groups, triangle runs, CSG objects (no counterpart in the reference) */
    uint64_t tests_run_transform; /* Ray::transform into the space of a run of triangles (src/ray.rs:18-23) */
    uint64_t team_shade_points; /* of shade_points: those whose shadow rays went to the lanes beside the pixel's own, all
                                   lights at once (small frames, fast arithmetic; same pixels) -- always counted */
} rtamd_stats;

typedef struct rtamd_ctx rtamd_ctx;

/* Library / device. Replaces nothing in the reference (it has no device). */
int rtamd_ctx_create(int device, rtamd_ctx** out);
void rtamd_ctx_destroy(rtamd_ctx* ctx);
const char* rtamd_last_error(const rtamd_ctx* ctx); /* ctx may be NULL: creation errors */

/* Replaces the scene state Camera::render reads (World after refresh_parents, Camera):
 * src/scene.rs:236-238. Validates indices and limits, then copies to the device. A scene that fails
 * validation, or whose copy to the device fails, leaves the previously uploaded one (if any) in place.
 * Clears the counters of rtamd_get_stats. */
int rtamd_scene_upload(rtamd_ctx* ctx, const rtamd_scene* scene);

/* Number of rows / pixels a rows-spec selects for the uploaded scene. */
int64_t rtamd_part_rows(const rtamd_ctx* ctx, const rtamd_rows* rows);

/* Replaces the pixel loop of Camera::render (src/scene.rs:240-246) for the selected rows:
 * ray_for_pixel + color_at(ray, max_remaining) + Canvas::write_pixel.
 * d_rgb: device pointer, 3 doubles per pixel (Canvas.pixels, src/tuple.rs:309-313), may be NULL;
 * d_rgb8: device pointer, 3 bytes per pixel quantised as `From<Color> for (usize,usize,usize)`
 * (src/tuple.rs:226-237), may be NULL. Asynchronous on `stream` (a hipStream_t, NULL = default).
 * A context owns the scratch of ONE render under way (work queue, frame stacks, counters, timing events): start
 * the next render, rtamd_color_at or upload of the same context on the same stream, or after this one has finished;
 * to overlap frames on one GPU use one context per frame in flight.  The call itself may allocate and wait for the
 * device the first time a partition of the frame is used (it builds the work-queue order for it). */
int rtamd_render_rows(rtamd_ctx* ctx, const rtamd_rows* rows, int mode, double* d_rgb, uint8_t* d_rgb8, void* stream);

/* No counterpart in the reference.  Does ahead of time what the first rtamd_render_rows of a partition of the frame does
 * on the host and with blocking copies (it builds the order of the work queue for it): nothing is rendered.  A host that
 * drives several GPUs calls it for every GPU's rows before it queues the first render, so that the renders start together
 * (rtamd_multi_render does).  Optional: rtamd_render_rows prepares what it finds unprepared. */
int rtamd_prepare_rows(rtamd_ctx* ctx, const rtamd_rows* rows, int mode);

/* No counterpart in the reference.  rtamd_render_rows for callers that cannot allocate device memory or wait for the
 * device: code under HIP graph capture, serving loops, harnesses that forbid hipMalloc.  Same arguments, same pixels
 * to the bit.  The work-queue counters and the per-lane stacks of shade-point frames live in a file-scope device
 * array of RTAMD_STATIC_SCRATCH_BYTES that every GPU the library runs on holds until the process ends; the call
 * makes no hipMalloc / hipFree and no blocking copy (it also leaves out the table that puts costly pixels first in
 * the work queue: the order of work differs, the results do not).  Limit: the frame stacks of at least one block per
 * CU (or of the whole grid, if smaller) must fit -- recursion budget x 8 doubles (fast) or 22 doubles (strict) x 256
 * lanes per block in 64 MiB, i.e. any budget in fast arithmetic, up to 5 in strict; beyond that RTAMD_E_LIMIT with
 * the figures in rtamd_last_error.  With fewer blocks than rtamd_render_rows would launch (strict, budget 2..5) it is
 * slower.  All contexts of one GPU share the array: calls that overlap on one GPU race (and so do a call of this and a
 * call of rtamd_render_rows on the same context, which share its counters).  The call records no timing events (an
 * event recorded while a stream is being captured can be neither waited for nor read): rtamd_stats.kernel_ms is 0
 * after it. */
#define RTAMD_STATIC_SCRATCH_BYTES (((size_t)64 << 20) + 8192)
int rtamd_render_rows_static(rtamd_ctx* ctx, const rtamd_rows* rows, int mode, double* d_rgb, uint8_t* d_rgb8, void* stream);

/* Camera::render for the whole frame into host memory (either pointer may be NULL). Blocking.
 * The device-side frame is kept by the context between calls; the copy back runs at the rate of the
 * PCIe link when the destination came from rtamd_host_alloc, and through the driver's staging otherwise.
 * A frame of 1.5 M pixels or more is rendered as four chunks of rows on a stream of the context's own, each copied
 * to the host while the chunks after it render (same pixels; rtamd_stats.kernel_ms then spans all the chunks). */
int rtamd_render(rtamd_ctx* ctx, int mode, double* h_rgb, uint8_t* h_rgb8);

/* Page-locked host memory for the frame: what Canvas::new's Vec<Color> (src/tuple.rs:315-321) becomes when
 * the canvas is to receive a GPU render. NULL when the allocation fails. */
void* rtamd_host_alloc(size_t bytes);
void rtamd_host_free(void* p);

/* World::color_at(ray, remaining) (src/scene.rs:53-63) for n caller-supplied rays:
 * h_rays = n x 6 doubles (origin xyz, direction xyz), h_rgb = n x 3 doubles. Blocking. */
int rtamd_color_at(rtamd_ctx* ctx, int mode, const double* h_rays, int64_t n, int remaining, double* h_rgb);

/* No counterpart in the reference.  on != 0: the renders of this context that follow use kernels that also count the
 * tests their traversals execute (the tests_* fields of rtamd_stats; they stay zero otherwise).  Same pixels; the
 * counting costs 5-15 % of a frame, which is why it is a choice. */
int rtamd_count_work(rtamd_ctx* ctx, int on);

/* Counters accumulated since the last reset (waits for the stream of the last render). */
int rtamd_get_stats(rtamd_ctx* ctx, rtamd_stats* out);
int rtamd_reset_stats(rtamd_ctx* ctx);

/* ---- Several GPUs, one frame.  No counterpart in the reference: its pixel loop (src/scene.rs:240-246) is one thread;
 * what makes the split possible is that its pixels are independent.  A host in any language gets an assembled frame
 * from N GPUs through these calls; no collective library, no Python. */

/* Where the rows of a rows-spec lie in the frame: `n_chunks` runs of `chunk_rows` rows that start at rows
 * first_row, first_row + stride_rows, ..., and after them a last run of `tail_rows` rows (0: none) at
 * first_row + n_chunks * stride_rows.  The compact output of rtamd_render_rows holds them in that order.  Pure
 * arithmetic (no context, no GPU): a host that moves the rows itself -- ncclSend / ncclRecv, MPI, its own peer
 * copies -- copies runs instead of rows. */
typedef struct rtamd_layout {
    int64_t first_row, chunk_rows, stride_rows, n_chunks, tail_rows;
} rtamd_layout;
int rtamd_rows_layout(int vsize, const rtamd_rows* rows, rtamd_layout* out);

typedef struct rtamd_multi rtamd_multi;

/* One context per entry of `devices` (1..64 entries; an entry may repeat).  Rank 0 -- devices[0] -- holds the
 * assembled frame; every other GPU must be able to write to it (peer access, xGMI on an MI355X node). */
int rtamd_multi_create(const int* devices, int n_devices, rtamd_multi** out);
void rtamd_multi_destroy(rtamd_multi* m);
const char* rtamd_multi_last_error(const rtamd_multi* m); /* m may be NULL: creation errors */

/* rtamd_scene_upload to every rank (the scene is small and read-only: replicated, not partitioned). */
int rtamd_multi_upload(rtamd_multi* m, const rtamd_scene* scene);

#define RTAMD_WANT_F64 1
#define RTAMD_WANT_U8 2
/* Camera::render over all ranks: rows in tiles of `tile_rows`, tile i in part i % n_parts, rank 0 renders the first
 * `first_parts` parts and every other rank `other_parts` (n_parts = first_parts + (n - 1) * other_parts; 1, 1 is the
 * plain cyclic split -- rank 0's rows do not travel, so a larger share for it can pay).  Every rank renders into a
 * buffer on its own GPU and writes its runs of rows into the frame on rank 0's GPU (one strided copy per rank, each
 * over its own link).  The frame is copied to h_rgb / h_rgb8 where those are not NULL; `want` (RTAMD_WANT_*) asks for
 * a frame that stays on the device (rtamd_multi_frame); with h_rgb NULL and only RTAMD_WANT_U8 (or h_rgb8) asked for,
 * 3 bytes per pixel instead of 24 cross the links.  Blocking.  Same pixels as rtamd_render, to the bit, and the same
 * refusal: RTAMD_E_OVERFLOW when a per-lane CSG intersection list overflowed on any rank. */
int rtamd_multi_render(rtamd_multi* m, int mode, int tile_rows, int first_parts, int other_parts, double* h_rgb, uint8_t* h_rgb8, int want);

/* Device pointers (on devices[0]) of the frame the last rtamd_multi_render assembled; NULL for a kind it was not asked for. */
int rtamd_multi_frame(rtamd_multi* m, double** d_rgb, uint8_t** d_rgb8);

/* Counters summed over the ranks (kernel_ms: the slowest rank's kernel), and per rank (arrays of n_devices, may be
 * NULL) the time of its render kernel and of putting its rows in place in the last rtamd_multi_render. */
int rtamd_multi_get_stats(rtamd_multi* m, rtamd_stats* total, double* render_ms, double* place_ms);
int rtamd_multi_reset_stats(rtamd_multi* m);

#ifdef __cplusplus
}
#endif
#endif /* RTAMD_H */
