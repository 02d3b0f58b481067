// Structures shared by the host side of the GPU library (rtamd_api.hip) and its kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rtamd.h"

// Device copy of an rtamd_leaf plus what the upload derives from the scene. 184 bytes.
struct DLeaf {
    double w2o[12];
    double p0, p1;
    int32_t kind;
    uint32_t flags;
    int32_t material;
    int32_t eq_class;
    int32_t chain_start;
    int32_t chain_len;
    int32_t tri_start;
    int32_t tri_count;
    int32_t order_base;  // rank of this leaf's first possible intersection in depth-first order
    int32_t ray_start;   // DLEAF_LEVELS: first of ray_len matrices of DScene::ray_levels a ray passes on its way to the leaf
    double inv_diag[3];  // DLEAF_DIAG: 1 / w2o[0], 1 / w2o[5], 1 / w2o[10]
    int32_t ray_len;
    int32_t pad_;
};
#define DLEAF_DIAG 0x100u    // device-side flag: the 3x3 part of w2o is diagonal
#define DLEAF_LEVELS 0x200u  // device-side flag: w2o is the product of two or more levels (groups, CSG objects, the leaf's own);
                             // STRICT arithmetic transforms rays by the levels one after the other (src/shape.rs:39-42)

// Everything shading a hit on one leaf needs, in one record, so that a lane finds it with one level of
// loads: the leaf's parameters, the first two levels of its transform chain (deeper chains continue in
// DScene::chain) and a copy of its material. 480 bytes.
struct alignas(16) DShade {
    int32_t kind;
    uint32_t flags;
    int32_t eq_class;
    int32_t chain_len;
    int32_t chain_start;
    int32_t tri_start;
    int32_t pad_[2];
    double p0, p1;
    double chain0[12], chain1[12];  // chain[chain_start], chain[chain_start + 1] (the second only if chain_len > 1)
    rtamd_material mat;
};

// One step of the traversal order, self-contained for a plain leaf so that one batch of scalar
// loads brings everything its intersection needs. 128 bytes, 16-byte aligned.
struct alignas(16) DVisit {
    double w2o[12];         //   0  DLEAF_DIAG leaves: slots 1, 2, 4 (zeros of the matrix) hold 1/w2o[0], 1/w2o[5], 1/w2o[10],
                            //      slots 6, 8, 9 the world-space centre -w2o[3]/w2o[0], -w2o[7]/w2o[5], -w2o[11]/w2o[10]
    double p0, p1;          //  96  cylinder/cone min, max; MESH: the bytes of p0 hold int32 tri_start, tri_count, and
                            //      w2o[0..5] hold the run's world-space bounds (lo xyz, hi xyz) instead of the matrix
    uint32_t kind_flags;    // 112  kind | leaf flags << 8
    int32_t run;            // 116  depth-first list: records from this one on that are axis-aligned cubes with one shadow
                            //      flag (0: this record is not such a cube)
    int32_t unit;           // 120  leaf index, or csg root index when the visit has RTAMD_VISIT_CSG
    uint32_t vflags_class;  // 124  RTAMD_VISIT_* | eq_class << 8
};

#define DBOUNDS_HAS_CONE 1u  // leaf flag of a bounds record: a cone is among the records it covers
#define DVISIT_BOUNDS 6u  // kind of a bounds record: w2o[0..5] = lo xyz, hi xyz (world space); unit = records to skip;
                          // with DBOUNDS_HAS_CONE the bytes of p0 hold the 64-bit set of the cone orientations below it
                          // (entry k of DScene::cone_quads -> bit k % 64)

// Small CSG objects (at most RTAMD_CSG_LIST_SMALL possible intersections on a ray) are evaluated without a sorted list:
// every possible intersection has a fixed slot -- the leaves of the object in depth-first order, each with its 2
// (sphere, cube), 1 (plane, triangle) or 4 (cylinder, cone) slots -- and the tables below say, per slot and per node,
// what the filter of src/shapes/csg.rs:54-82 needs.
struct DCsgSlot {
    int32_t key;       // leaf << RTAMD_KEY_SHIFT | slot inside the leaf
    int32_t eq_class;  // of the leaf
    uint32_t shadow;   // the leaf takes part in is_shadowed
    uint32_t pad_;
};
struct DCsgSlotNode {  // one per CSG node, in the order of rtamd_csg_node
    uint32_t range;  // slots of the leaves below this node
    uint32_t left;   // ... of those that `left.includes` (src/shapes/csg.rs:64)
    int32_t op;
    uint32_t pad_;
};

struct DScene {
    rtamd_camera cam;
    int32_t max_remaining;
    int32_t n_lights;
    int32_t n_visits;
    int32_t n_leaves;
    int32_t ri_safe;  // every refractive index is finite and > 0
    int32_t n_visits_dfs;
    int32_t n_cone_quads;  // entries of cone_quads
    int32_t pad_;
    double diag_thr;  // FAST: smallest |direction component| for which every axis-aligned cube sees |scaled component| >= EPSILON
    const rtamd_light* lights;
    const DLeaf* leaves;
    const DShade* shade;  // one per leaf
    const double* chain;
    const double* ray_levels;  // leaves with DLEAF_LEVELS: their levels, outermost first (12 doubles each)
    const double* tri_geom;
    const double* tri_sorted;  // the same triangles in the order of the runs the traversal tests (9 doubles each)
    const int32_t* tri_key;    // per triangle of tri_sorted: leaf << RTAMD_KEY_SHIFT | slot of its place in tri_geom
    const double* cone_quads;  // 8 doubles per distinct cone orientation of the scene: d -> a = Qxx dx^2 + Qyy dy^2 + Qzz dz^2 + 2Qxy dx dy + 2Qxz dx dz + 2Qyz dy dz, then |Q| (Frobenius) and a spare
    const double* mesh_box;  // per leaf 6 doubles: enlarged world-space bounds (lo xyz, hi xyz) of a MESH leaf's triangles
    const double* tri_normals;
    const rtamd_material* materials;
    const rtamd_csg_node* csg_nodes;
    const rtamd_csg_root* csg_roots;
    const DCsgSlot* csg_slots;         // small CSG objects only (kernel variant 3): slots of root r start at csg_slot0[r]
    const DCsgSlotNode* csg_slot_nodes;
    const int32_t* csg_slot0;
    const DVisit* visits;      // grouped by equality class: the order container passes need
    const DVisit* visits_dfs;  // one record per plain leaf / top-most CSG object in depth-first order
    const int32_t* class_material;  // material of each equality class
};

// Per-launch counters, one 64-bit word each (device memory, accumulated with atomics).
enum {
    ST_PRIMARY = 0,
    ST_SHADOW_RAYS,
    ST_REFLECT,
    ST_REFRACT,
    ST_CONTAINER,
    ST_SHADE_POINTS,
    ST_LIGHT_SAMPLES,
    ST_CSG_OVERFLOW,
    // work the traversal executed, in lane-tests: one ray against one leaf, box or triangle, counted for the lanes the
    // test was meant for (a wave executes it for all 64)
    ST_T_SPHERE,
    ST_T_PLANE,
    ST_T_CUBE,    // the reference's formula (cubes that are not axis-aligned, rays nearly parallel to a face, STRICT)
    ST_T_SLAB2,   // axis-aligned cube, first two slabs (FAST)
    ST_T_SLAB3,   // ... third slab and bookkeeping, for the rays the first two did not reject
    ST_T_CYLINDER,
    ST_T_CONE,
    ST_T_CAPS,    // end caps of a closed cylinder or cone
    ST_T_TRI1,    // triangle: determinant and u
    ST_T_TRI2,    // triangle: v and t, for the rays the first half did not reject
    ST_T_BOX,     // bounds records, boxes of runs of triangles and of CSG objects
    ST_T_RUNX,    // ray transformed into the space of a run of triangles
    ST_TEAM_POINTS,  // shade points whose shadow rays a team of lanes traced side by side
    ST_COUNT
};

#define RTAMD_CSG_LIST_SMALL 8  // most intersections one CSG object can put on a ray for the list-free kernel variant
#define RTAMD_BLOCK_W 8          // a work-queue block: 64 pixels, this many wide (16 x 4 is no faster and writes as much)
#define RTAMD_BLOCK_H (64 / RTAMD_BLOCK_W)
#define RTAMD_KEY_SHIFT 12       // intersection key = leaf index << 12 | slot inside the leaf (RTAMD_MAX_MESH_TRIS = 4096)
#define RTAMD_STAT_STRIDE 32     // 64-bit words per block in the counter array (ST_COUNT counters, the rest for diagnostic builds)
#define RTAMD_FRAME_F64_STRICT 22  // doubles per saved shade-point frame: STRICT keeps the whole shade point,
#define RTAMD_FRAME_F64_FAST 8     // FAST the refraction ray it still owes (origin, direction, weight, budget)
#define RTAMD_FRAME_F64 RTAMD_FRAME_F64_STRICT  // what the context's own buffer is sized for
#define RTAMD_STATIC_FRAME_BYTES ((size_t)64 << 20)  // frame stacks inside the static scratch of rtamd_render_rows_static
#define RTAMD_QUEUES 64        // counters the work queue is split into
#define RTAMD_QUEUE_STRIDE 16  // 64-bit words between two counters (128 bytes)

struct DWork {
    // which pixels
    int32_t hsize, vsize;
    int32_t tile_rows, part, n_parts, span;  // tiles i with i % n_parts in [part, part + span)
    int32_t remaining;          // recursion budget of the primary rays
    int64_t n_pixels;           // pixels of this part (or number of custom rays)
    int64_t n_groups;           // groups of 64 pixels
    int64_t n_queue;            // entries of the work queue: one per group, `team` per group that `order` marks
    int32_t gcols, n_rows;      // gcols != 0: a group is a block of RTAMD_BLOCK_W x RTAMD_BLOCK_H pixels, gcols blocks side by side, over the part's
                                // n_rows rows; gcols == 0: a group is 64 consecutive pixels (ceil(n_pixels / 64) groups)
    int64_t group_stride;       // coprime to n_groups: queue position g stands for pixel group (g * stride) % n_groups
    const double* custom_rays;  // NULL: camera rays; else n_pixels x 6
    const int32_t* order;       // NULL: groups in strided order; else queue position -> group of 64 pixels
    int32_t refill_min;         // a wave asks for its next group of pixels when at least this many of its lanes are idle
    int32_t mesh_few;           // a run of triangles that at most this many rays of a wave come near is tested one triangle per lane
    int32_t team;               // lanes per pixel (2 or 4) in the groups `order` marks with its sign bit; 1: no such groups
    double* out_rgb;            // may be NULL
    uint8_t* out_rgb8;          // may be NULL
    // scratch the library owns
    unsigned long long* queue_head;  // [RTAMD_QUEUES] counters, RTAMD_QUEUE_STRIDE words apart
    unsigned long long* stats;       // [blocks][RTAMD_STAT_STRIDE]: every block adds into its own slot
    double* frames;                  // [max_remaining][RTAMD_FRAME_F64][total_lanes]
    int64_t total_lanes;
};

// Implemented once per arithmetic mode (rtamd_kernels_strict.hip / rtamd_kernels_fast.hip).
// variant: base (0 scene without triangles and CSG, 1 with triangles, 2 with CSG, 3 with CSG objects of at most
// RTAMD_CSG_LIST_SMALL possible intersections per ray) + 4 if there are axis-aligned cubes outside CSG objects
// + 8 if there are cylinders or cones + 16 if the visit lists hold bounds records (bases 2 and 3 ignore all three)
// + 32 CSG scene without triangles + 64 the twin that counts the tests it executes
hipError_t rtamd_launch_strict(const DScene& s, const DWork& w, int variant, int blocks, hipStream_t stream);
hipError_t rtamd_launch_fast(const DScene& s, const DWork& w, int variant, int blocks, hipStream_t stream);
hipError_t rtamd_occupancy_strict(int variant, int* blocks_per_cu);
hipError_t rtamd_occupancy_fast(int variant, int* blocks_per_cu);
// the strict kernels for scenes with leaves that have DLEAF_LEVELS
hipError_t rtamd_launch_levels(const DScene& s, const DWork& w, int variant, int blocks, hipStream_t stream);
hipError_t rtamd_occupancy_levels(int variant, int* blocks_per_cu);
// the fast kernels with teams of lanes per pixel (variant: without the counting bit)
hipError_t rtamd_launch_teams(const DScene& s, const DWork& w, int variant, int blocks, hipStream_t stream);
hipError_t rtamd_occupancy_teams(int variant, int* blocks_per_cu);
// ... built for three waves per SIMD (no register spilled to memory), for frames whose every piece of work gets a wave even so
hipError_t rtamd_launch_teams3(const DScene& s, const DWork& w, int variant, int blocks, hipStream_t stream);
hipError_t rtamd_occupancy_teams3(int variant, int* blocks_per_cu);
