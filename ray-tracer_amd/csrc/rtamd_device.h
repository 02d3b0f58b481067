// Structures shared by the host side of the GPU library (rtamd_api.hip) and its kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rtamd.h"

// Device copy of an rtamd_leaf plus what the upload derives from the scene. 160 bytes.
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
    int32_t pad_[3];
};

struct DScene {
    rtamd_camera cam;
    int32_t max_remaining;
    int32_t n_lights;
    int32_t n_visits;
    int32_t n_leaves;
    int32_t ri_safe;  // every refractive index is finite and > 0
    int32_t pad_;
    const rtamd_light* lights;
    const DLeaf* leaves;
    const double* chain;
    const double* tri_geom;
    const double* tri_normals;
    const rtamd_material* materials;
    const rtamd_csg_node* csg_nodes;
    const rtamd_csg_root* csg_roots;
    const rtamd_visit* visits;
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
    ST_COUNT
};

#define RTAMD_FRAME_F64 20   // doubles per saved shade-point frame
#define RTAMD_CHUNK 256      // pixels a wave takes from the queue at a time

struct DWork {
    // which pixels
    int32_t hsize, vsize;
    int32_t tile_rows, part, n_parts;
    int32_t remaining;          // recursion budget of the primary rays
    int64_t n_pixels;           // pixels of this part (or number of custom rays)
    const double* custom_rays;  // NULL: camera rays; else n_pixels x 6
    double* out_rgb;            // may be NULL
    uint8_t* out_rgb8;          // may be NULL
    // scratch the library owns
    unsigned long long* queue_head;  // next unassigned pixel
    unsigned long long* stats;       // ST_COUNT words
    double* frames;                  // [max_remaining][RTAMD_FRAME_F64][total_lanes]
    int64_t total_lanes;
};

// Implemented once per arithmetic mode (rtamd_kernels_strict.hip / rtamd_kernels_fast.hip).
hipError_t rtamd_launch_strict(const DScene& s, const DWork& w, bool has_csg, int blocks, hipStream_t stream);
hipError_t rtamd_launch_fast(const DScene& s, const DWork& w, bool has_csg, int blocks, hipStream_t stream);
hipError_t rtamd_occupancy_strict(bool has_csg, int* blocks_per_cu);
hipError_t rtamd_occupancy_fast(bool has_csg, int* blocks_per_cu);
