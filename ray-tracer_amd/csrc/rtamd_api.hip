// Host side of the GPU library: the C ABI of include/rtamd.h.
// Owns the device copies of a flattened scene and the scratch the render kernel needs
// (pixel queue head, counters, per-lane shade-point frames); launches the kernel of the
// requested arithmetic mode on the caller's stream.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <array>
#include <functional>
#include <mutex>
#include <new>
#include <utility>
#include <string>
#include <vector>

#include "rtamd_device.h"

static_assert(sizeof(rtamd_leaf) == 152, "rtamd_leaf layout");
static_assert(sizeof(DLeaf) == 184, "DLeaf layout");
static_assert(sizeof(rtamd_material) == 232, "rtamd_material layout");
static_assert(sizeof(rtamd_visit) == 16, "rtamd_visit layout");
static_assert(sizeof(DVisit) == 128, "DVisit layout");

struct rtamd_ctx {
    int device = 0;
    int num_cus = 0;
    std::string err;
    bool have_scene = false, has_csg = false;
    int variant = 0;  // kernel variant of the uploaded scene: base (0 plain, 1 triangles, 2 CSG, 3 small CSG) + 4 with axis-aligned cubes + 8 with cylinders or cones + 16 with bounds records + 32 CSG without triangles
    DScene ds{};
    std::vector<void*> scene_bufs;
    unsigned long long* d_queue = nullptr;
    unsigned long long* d_stats = nullptr;
    double* d_frames = nullptr;
    size_t frames_levels = 0;
    int64_t total_lanes = 0;
    int max_blocks[4][128] = {};  // [mode][variant]; [2]: the fast kernels with teams of lanes per pixel, [3]: those at three waves per SIMD
    bool has_levels = false;      // the scene has leaves with DLEAF_LEVELS: strict arithmetic runs the kernels that know them
    bool count_work = false;       // launches use the kernels that count the tests they execute (rtamd_count_work)
    int stat_blocks = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timed = false;
    // Pixels that may be costly: the screen rectangles (x0, x1, y0, y1, inclusive) around every leaf that is
    // both reflective and transparent -- each hit on one owes two children, so its pixels run the longest
    // chains of dependent rays.  Their groups go to the front of the work queue.
    std::vector<std::array<int, 4>> costly;
    // queue position -> group of 64 pixels, one table per partition of the frame that has been rendered (a few are kept:
    // the parts a multi-GPU rank holds, the chunks rtamd_render cuts a large frame into)
    struct OrderTable {
        int32_t* d = nullptr;
        size_t cap = 0;
        std::array<int64_t, 8> key{{-1, -1, -1, -1, -1, -1, -1, -1}};
        size_t hot = 0, len = 0;  // groups the table puts first; its entries
        int teams = 1;            // lanes per pixel in the groups it marks
        bool used = false;
    };
    static constexpr int kOrderTables = 8;
    OrderTable orders[kOrderTables];
    int order_next = 0;  // the slot the next new partition takes
    // rtamd_render with a large frame: rendered in chunks of rows whose copies to the host overlap the chunks after them
    hipStream_t s_render = nullptr, s_copy = nullptr;
    hipEvent_t ev_chunk[4] = {nullptr, nullptr, nullptr, nullptr};
    bool skip_ev0 = false, skip_ev1 = false;  // (launches between the first and the last chunk leave the timing events alone)
    // device-side frame of rtamd_render (host destination), kept between calls
    double* d_out = nullptr;
    uint8_t* d_out8 = nullptr;
    size_t out_cap = 0, out8_cap = 0;
    unsigned char* static_scratch = nullptr;  // this device's copy of the file-scope `scratch` array
    unsigned long long ovf_seen = 0;          // CSG list overflows counted up to the last check by rtamd_render
};

static std::string g_create_err;

// Scratch of rtamd_render_rows_static: the work-queue counters and the per-lane stacks of shade-point frames that
// rtamd_render_rows keeps in memory the context allocated.  One copy per GPU, held from the moment the library is
// loaded on it until the process ends.  In host code the name below is only a handle: the address the kernels get
// is asked for per device (static_scratch_of), with that device current.
#define RTAMD_STATIC_QUEUE_BYTES ((size_t)RTAMD_QUEUES * RTAMD_QUEUE_STRIDE * sizeof(unsigned long long))
static_assert(RTAMD_STATIC_QUEUE_BYTES + RTAMD_STATIC_FRAME_BYTES == RTAMD_STATIC_SCRATCH_BYTES, "layout of the static scratch");
static __device__ __align__(256) unsigned char scratch[RTAMD_STATIC_SCRATCH_BYTES];

#define RTAMD_MAX_DEVICES 64
static std::mutex g_scratch_lock;
static unsigned char* g_scratch_ptr[RTAMD_MAX_DEVICES];  // per device: where its copy of `scratch` lives

// Address of the current device's copy of `scratch` (device = its index).  NULL and *err set when the runtime refuses.
static unsigned char* static_scratch_of(int device, hipError_t* err) {
    *err = hipSuccess;
    if (device < 0 || device >= RTAMD_MAX_DEVICES) {
        *err = hipErrorInvalidDevice;
        return nullptr;
    }
    std::lock_guard<std::mutex> hold(g_scratch_lock);
    if (!g_scratch_ptr[device]) {
        void* p = nullptr;
        *err = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (*err != hipSuccess) return nullptr;
        g_scratch_ptr[device] = static_cast<unsigned char*>(p);
    }
    return g_scratch_ptr[device];
}

#define HIPCHK(ctx, call)                                                                          \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                        \
            return RTAMD_E_HIP;                                                                    \
        }                                                                                          \
    } while (0)

static int fail(rtamd_ctx* ctx, int code, const std::string& msg) {
    ctx->err = msg;
    return code;
}

// The functions below are declared extern "C" in include/rtamd.h.

const char* rtamd_last_error(const rtamd_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }

int rtamd_ctx_create(int device, rtamd_ctx** out) {
    if (!out) {
        g_create_err = "rtamd_ctx_create: out is NULL";
        return RTAMD_E_INVALID;
    }
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0) {
        g_create_err = std::string("no HIP device available: ") + (e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
        return RTAMD_E_HIP;
    }
    if (device < 0 || device >= n) {
        g_create_err = "device index out of range";
        return RTAMD_E_INVALID;
    }
    auto* ctx = new (std::nothrow) rtamd_ctx();  // (nothing may unwind across the C boundary)
    if (!ctx) {
        g_create_err = "rtamd_ctx_create: out of host memory";
        return RTAMD_E_LIMIT;
    }
    ctx->device = device;
    auto bail = [&](const char* what, hipError_t err) {
        g_create_err = std::string(what) + ": " + hipGetErrorString(err);
        rtamd_ctx_destroy(ctx);  // releases whatever exists so far
        return RTAMD_E_HIP;
    };
    if ((e = hipSetDevice(device)) != hipSuccess) return bail("hipSetDevice", e);
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, device)) != hipSuccess) return bail("hipGetDeviceProperties", e);
    ctx->num_cus = prop.multiProcessorCount;
    for (int v = 0; v < 128; v++) {
        int b = 0;
        if ((e = rtamd_occupancy_strict(v, &b)) != hipSuccess) return bail("occupancy(strict)", e);
        int bl = 0;  // (the strict kernels with and without the code for levels run under one entry: the smaller grid fits both)
        if ((e = rtamd_occupancy_levels(v, &bl)) != hipSuccess) return bail("occupancy(levels)", e);
        ctx->max_blocks[RTAMD_MODE_STRICT][v] = std::max(1, std::min(b, bl)) * ctx->num_cus;
        if ((e = rtamd_occupancy_fast(v, &b)) != hipSuccess) return bail("occupancy(fast)", e);
        ctx->max_blocks[RTAMD_MODE_FAST][v] = std::max(1, b) * ctx->num_cus;
        if ((e = rtamd_occupancy_teams(v, &b)) != hipSuccess) return bail("occupancy(teams)", e);
        ctx->max_blocks[2][v] = std::max(1, b) * ctx->num_cus;
        if ((e = rtamd_occupancy_teams3(v, &b)) != hipSuccess) return bail("occupancy(teams3)", e);
        ctx->max_blocks[3][v] = std::max(1, b) * ctx->num_cus;
    }
    int mb = 0;
    for (int m = 0; m < 4; m++)
        for (int c = 0; c < 128; c++) mb = std::max(mb, ctx->max_blocks[m][c]);
    ctx->total_lanes = (int64_t)mb * 256;
    ctx->stat_blocks = mb;
    if ((e = hipMalloc(&ctx->d_queue, RTAMD_QUEUES * RTAMD_QUEUE_STRIDE * sizeof(unsigned long long))) != hipSuccess) return bail("hipMalloc(queue)", e);
    if ((e = hipMalloc(&ctx->d_stats, (size_t)mb * RTAMD_STAT_STRIDE * sizeof(unsigned long long))) != hipSuccess) return bail("hipMalloc(stats)", e);
    if ((e = hipMemset(ctx->d_stats, 0, (size_t)mb * RTAMD_STAT_STRIDE * sizeof(unsigned long long))) != hipSuccess) return bail("hipMemset", e);
    if ((e = hipEventCreate(&ctx->ev0)) != hipSuccess) return bail("hipEventCreate", e);
    if ((e = hipEventCreate(&ctx->ev1)) != hipSuccess) return bail("hipEventCreate", e);
    ctx->static_scratch = static_scratch_of(device, &e);  // (the device is current)
    if (!ctx->static_scratch) return bail("hipGetSymbolAddress(scratch)", e);
    *out = ctx;
    return RTAMD_OK;
}

static void free_scene(rtamd_ctx* ctx) {
    for (void* p : ctx->scene_bufs) (void)hipFree(p);
    ctx->scene_bufs.clear();
    ctx->have_scene = false;
}

void rtamd_ctx_destroy(rtamd_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    free_scene(ctx);
    if (ctx->d_queue) (void)hipFree(ctx->d_queue);
    if (ctx->d_stats) (void)hipFree(ctx->d_stats);
    if (ctx->d_frames) (void)hipFree(ctx->d_frames);
    for (auto& t : ctx->orders)
        if (t.d) (void)hipFree(t.d);
    if (ctx->s_render) (void)hipStreamDestroy(ctx->s_render);
    if (ctx->s_copy) (void)hipStreamDestroy(ctx->s_copy);
    for (auto& e : ctx->ev_chunk)
        if (e) (void)hipEventDestroy(e);
    if (ctx->d_out) (void)hipFree(ctx->d_out);
    if (ctx->d_out8) (void)hipFree(ctx->d_out8);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    delete ctx;
}

template <class T>
static int upload(rtamd_ctx* ctx, const T* src, size_t n, const T** dst) {
    void* p = nullptr;
    size_t bytes = std::max<size_t>(n, 1) * sizeof(T);
    HIPCHK(ctx, hipMalloc(&p, bytes));
    ctx->scene_bufs.push_back(p);
    if (n) HIPCHK(ctx, hipMemcpy(p, src, n * sizeof(T), hipMemcpyHostToDevice));
    *dst = (const T*)p;
    return RTAMD_OK;
}

static int ensure_frames(rtamd_ctx* ctx, int levels) {
    if ((size_t)levels <= ctx->frames_levels) return RTAMD_OK;
    if (ctx->d_frames) {
        HIPCHK(ctx, hipDeviceSynchronize());
        HIPCHK(ctx, hipFree(ctx->d_frames));
        ctx->d_frames = nullptr;
        ctx->frames_levels = 0;
    }
    size_t bytes = (size_t)levels * RTAMD_FRAME_F64 * (size_t)ctx->total_lanes * sizeof(double);
    HIPCHK(ctx, hipMalloc(&ctx->d_frames, bytes));
    ctx->frames_levels = (size_t)levels;
    return RTAMD_OK;
}

static int scene_upload(rtamd_ctx* ctx, const rtamd_scene* s);

int rtamd_scene_upload(rtamd_ctx* ctx, const rtamd_scene* s) {
    if (!ctx || !s) return RTAMD_E_INVALID;
    try {  // the host-side tables below are std::vectors: nothing may unwind across the C boundary
        return scene_upload(ctx, s);
    } catch (const std::bad_alloc&) {
        return fail(ctx, RTAMD_E_LIMIT, "scene upload: out of host memory");
    } catch (...) {
        return fail(ctx, RTAMD_E_INVALID, "scene upload: unexpected failure while building the device tables");
    }
}

// Keeps the previously uploaded scene alive until the new one is complete on the device: a failure on the way
// frees what the new one had got so far and puts the old buffers back.
namespace {
struct UploadGuard {
    rtamd_ctx* ctx;
    std::vector<void*> old;
    bool done = false;
    explicit UploadGuard(rtamd_ctx* c) : ctx(c) { old.swap(c->scene_bufs); }
    ~UploadGuard() {
        if (done) {
            for (void* p : old) (void)hipFree(p);
        } else {
            for (void* p : ctx->scene_bufs) (void)hipFree(p);
            ctx->scene_bufs.swap(old);
        }
    }
};
}  // namespace

static int scene_upload(rtamd_ctx* ctx, const rtamd_scene* s) {
    HIPCHK(ctx, hipSetDevice(ctx->device));
    // ---- validation: every index the kernel follows is checked here
    if (s->camera.hsize <= 0 || s->camera.vsize <= 0) return fail(ctx, RTAMD_E_INVALID, "camera size must be positive");
    if (s->max_remaining < 0 || s->max_remaining > RTAMD_MAX_REMAINING) return fail(ctx, RTAMD_E_LIMIT, "max_remaining out of range");
    if (s->n_lights < 0 || s->n_lights > RTAMD_MAX_LIGHTS) return fail(ctx, RTAMD_E_LIMIT, "too many lights");
    if (s->n_leaves >= (1 << (31 - RTAMD_KEY_SHIFT))) return fail(ctx, RTAMD_E_LIMIT, "too many leaves");
    if (s->n_leaves < 0 || s->n_chain < 0 || s->n_tris < 0 || s->n_materials < 0 || s->n_csg_nodes < 0 || s->n_csg_roots < 0 ||
        s->n_visits < 0 || s->n_classes < 0)
        return fail(ctx, RTAMD_E_INVALID, "negative count");
    if (s->n_classes > std::max(s->n_leaves, 1)) return fail(ctx, RTAMD_E_INVALID, "more equality classes than leaves");
    if ((s->n_lights && !s->lights) || (s->n_leaves && !s->leaves) || (s->n_chain && !s->chain) ||
        (s->n_tris && (!s->tri_geom || !s->tri_normals)) || (s->n_materials && !s->materials) ||
        (s->n_csg_nodes && !s->csg_nodes) || (s->n_csg_roots && !s->csg_roots) || (s->n_visits && !s->visits))
        return fail(ctx, RTAMD_E_INVALID, "NULL array with a non-zero count");
    std::vector<DLeaf> dl((size_t)s->n_leaves);
    std::vector<double> ray_levels;  // the levels of the leaves that get DLEAF_LEVELS
    long long order = 0;
    for (int i = 0; i < s->n_leaves; i++) {
        const rtamd_leaf& L = s->leaves[i];
        if (L.kind < RTAMD_SPHERE || L.kind > RTAMD_MESH) return fail(ctx, RTAMD_E_INVALID, "leaf kind out of range");
        if (L.material < 0 || L.material >= s->n_materials) return fail(ctx, RTAMD_E_INVALID, "leaf material index out of range");
        if (L.chain_len < 1 || L.chain_len > RTAMD_MAX_CHAIN || L.chain_start < 0 || L.chain_start + L.chain_len > s->n_chain)
            return fail(ctx, RTAMD_E_INVALID, "leaf transform chain out of range");
        if (L.ray_len < 0 || L.ray_len > RTAMD_MAX_RAY_LEVELS || (L.ray_len > 0 && (L.ray_start < 0 || L.ray_start + L.ray_len > s->n_chain)))
            return fail(ctx, RTAMD_E_INVALID, "leaf ray levels out of range");
        if (L.eq_class < 0 || L.eq_class >= std::max(s->n_classes, 1)) return fail(ctx, RTAMD_E_INVALID, "leaf class out of range");
        int slots = 0;
        if (L.kind == RTAMD_MESH) {
            if (L.tri_count > RTAMD_MAX_MESH_TRIS) return fail(ctx, RTAMD_E_LIMIT, "mesh leaf with more than RTAMD_MAX_MESH_TRIS triangles");
            if (L.tri_count < 1 || L.tri_start < 0 || (long long)L.tri_start + L.tri_count > s->n_tris)
                return fail(ctx, RTAMD_E_INVALID, "mesh triangle range out of range");
            slots = L.tri_count;
        } else {
            slots = 4;
        }
        DLeaf& D = dl[i];
        std::memset(&D, 0, sizeof D);
        std::memcpy(D.w2o, L.w2o, sizeof L.w2o);
        D.p0 = L.p0, D.p1 = L.p1, D.kind = L.kind, D.flags = L.flags, D.material = L.material, D.eq_class = L.eq_class;
        D.chain_start = L.chain_start, D.chain_len = L.chain_len, D.tri_start = L.tri_start, D.tri_count = L.tri_count;
        D.order_base = (int)order;
        if (L.ray_len > 1) {
            // A level whose matrix is the identity (a CSG object or a group nobody moved) hands the ray on as it came:
            // 1 * x + 0 * y + 0 * z (+ 0) is x, at most with another sign on a zero, and the intersection tests never read
            // the sign of a zero (every division by a component of the ray is behind |component| >= EPSILON).  Such levels
            // are left out; a leaf with one level left is transformed by w2o, which then is that level (same argument).
            const size_t first = ray_levels.size();
            for (int k = 0; k < L.ray_len; k++) {
                const double* lv = s->chain + ((size_t)L.ray_start + (size_t)k) * 12;
                bool identity = true;
                for (int q = 0; q < 12; q++) identity = identity && lv[q] == (q % 5 == 0 ? 1.0 : 0.0);
                if (!identity) ray_levels.insert(ray_levels.end(), lv, lv + 12);
            }
            const size_t n = (ray_levels.size() - first) / 12;
            if (n > 1)
                D.flags |= DLEAF_LEVELS, D.ray_start = (int32_t)(first / 12), D.ray_len = (int32_t)n;
            else
                ray_levels.resize(first);
        }
        const double* m = L.w2o;
        if (m[1] == 0.0 && m[2] == 0.0 && m[4] == 0.0 && m[6] == 0.0 && m[8] == 0.0 && m[9] == 0.0 && m[0] != 0.0 && m[5] != 0.0 &&
            m[10] != 0.0) {
            D.flags |= DLEAF_DIAG;
            D.inv_diag[0] = 1.0 / m[0], D.inv_diag[1] = 1.0 / m[5], D.inv_diag[2] = 1.0 / m[10];
        }
        order += slots;
        if (order >= 0x7fffff00LL) return fail(ctx, RTAMD_E_LIMIT, "too many intersections slots for 32-bit order keys");
    }
    bool ri_safe = true;
    for (int i = 0; i < s->n_materials; i++) {
        const rtamd_material& M = s->materials[i];
        if (M.pattern_kind < RTAMD_PAT_NONE || M.pattern_kind > RTAMD_PAT_TEST) return fail(ctx, RTAMD_E_INVALID, "pattern kind out of range");
        if (!(std::isfinite(M.refractive_index) && M.refractive_index > 0.0)) ri_safe = false;
    }
    bool csg_small = true;  // no CSG object can put more than RTAMD_CSG_LIST_SMALL intersections on a ray
    for (int i = 0; i < s->n_csg_roots; i++) {
        const rtamd_csg_root& R = s->csg_roots[i];
        if (R.node_start < 0 || R.node_count < 1 || R.node_start + R.node_count > s->n_csg_nodes || R.leaf_lo < 0 ||
            R.leaf_hi > s->n_leaves || R.leaf_lo > R.leaf_hi)
            return fail(ctx, RTAMD_E_INVALID, "csg root out of range");
        if (R.leaf_hi - R.leaf_lo > RTAMD_MAX_CSG_LEAVES) return fail(ctx, RTAMD_E_LIMIT, "csg object with too many leaves");
        long long most = 0;  // sphere, cube: 2; plane: 1; cylinder, cone: walls 2 + caps 2; a triangle: 1
        for (int l = R.leaf_lo; l < R.leaf_hi; l++) {
            const rtamd_leaf& L = s->leaves[l];
            most += L.kind == RTAMD_PLANE ? 1 : L.kind == RTAMD_MESH ? L.tri_count : (L.kind == RTAMD_CYLINDER || L.kind == RTAMD_CONE) ? 4 : 2;
        }
        if (most > RTAMD_CSG_LIST_SMALL) csg_small = false;
        for (int q = R.node_start; q < R.node_start + R.node_count; q++) {
            const rtamd_csg_node& N = s->csg_nodes[q];
            if (N.op < RTAMD_CSG_UNION || N.op > RTAMD_CSG_DIFFERENCE || N.leaf_lo < R.leaf_lo || N.leaf_hi > R.leaf_hi ||
                N.leaf_lo > N.leaf_mid || N.leaf_mid > N.leaf_hi)
                return fail(ctx, RTAMD_E_INVALID, "csg node out of range");
        }
    }
    for (int i = 0; i < s->n_visits; i++) {
        const rtamd_visit& V = s->visits[i];
        int lim = (V.flags & RTAMD_VISIT_CSG) ? s->n_csg_roots : s->n_leaves;
        if (V.unit < 0 || V.unit >= lim) return fail(ctx, RTAMD_E_INVALID, "visit unit out of range");
        if (V.class_material < 0 || V.class_material >= s->n_materials) return fail(ctx, RTAMD_E_INVALID, "visit material out of range");
        if (V.eq_class < 0 || V.eq_class >= std::max(s->n_classes, 1) || V.eq_class >= (1 << 23))
            return fail(ctx, RTAMD_E_INVALID, "visit class out of range");
    }
    // ---- device copies (the scene uploaded before stays in place until all of them exist)
    HIPCHK(ctx, hipDeviceSynchronize());
    UploadGuard guard(ctx);
    DScene d{};
    d.cam = s->camera;
    d.max_remaining = s->max_remaining, d.n_lights = s->n_lights, d.n_visits = s->n_visits, d.n_leaves = s->n_leaves;
    d.ri_safe = ri_safe ? 1 : 0;
    int rc;
    if ((rc = upload(ctx, s->lights, (size_t)s->n_lights, &d.lights))) return rc;
    if ((rc = upload(ctx, dl.data(), dl.size(), &d.leaves))) return rc;
    std::vector<DShade> sh((size_t)s->n_leaves);
    for (int i = 0; i < s->n_leaves; i++) {
        const rtamd_leaf& L = s->leaves[i];
        DShade& H = sh[i];
        std::memset(&H, 0, sizeof H);
        H.kind = L.kind, H.flags = L.flags, H.eq_class = L.eq_class, H.chain_len = L.chain_len, H.chain_start = L.chain_start;
        H.tri_start = L.tri_start, H.p0 = L.p0, H.p1 = L.p1;
        std::memcpy(H.chain0, s->chain + (size_t)L.chain_start * 12, sizeof H.chain0);
        if (L.chain_len > 1) std::memcpy(H.chain1, s->chain + ((size_t)L.chain_start + 1) * 12, sizeof H.chain1);
        H.mat = s->materials[L.material];
    }
    if ((rc = upload(ctx, sh.data(), sh.size(), &d.shade))) return rc;
    // World-space bounds of every leaf that has any (not planes, not cylinders or cones without ends), a little
    // enlarged.  They are only ever used to pass over leaves no ray of a wave can meet.
    std::vector<double> boxes((size_t)s->n_leaves * 6, 0.0);
    std::vector<char> bounded((size_t)s->n_leaves, 0), has_box((size_t)s->n_leaves, 0);
    for (int i = 0; i < s->n_leaves; i++) {
        const rtamd_leaf& L = s->leaves[i];
        double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        // object -> world: inverse of the affine world -> object matrix (bounds only: rounding is covered by the padding)
        const double* m = L.w2o;
        const double det = m[0] * (m[5] * m[10] - m[6] * m[9]) - m[1] * (m[4] * m[10] - m[6] * m[8]) + m[2] * (m[4] * m[9] - m[5] * m[8]);
        bool finite = std::isfinite(det) && det != 0.0;
        double inv[9] = {0};
        if (finite) {
            inv[0] = (m[5] * m[10] - m[6] * m[9]) / det, inv[1] = (m[2] * m[9] - m[1] * m[10]) / det, inv[2] = (m[1] * m[6] - m[2] * m[5]) / det;
            inv[3] = (m[6] * m[8] - m[4] * m[10]) / det, inv[4] = (m[0] * m[10] - m[2] * m[8]) / det, inv[5] = (m[2] * m[4] - m[0] * m[6]) / det;
            inv[6] = (m[4] * m[9] - m[5] * m[8]) / det, inv[7] = (m[1] * m[8] - m[0] * m[9]) / det, inv[8] = (m[0] * m[5] - m[1] * m[4]) / det;
        }
        auto add_point = [&](const double* q) {  // an object-space point
            double p[3], w[3];
            for (int k = 0; k < 3; k++) p[k] = q[k] - m[4 * k + 3];  // minus the translation
            for (int k = 0; k < 3; k++) {
                w[k] = inv[3 * k] * p[0] + inv[3 * k + 1] * p[1] + inv[3 * k + 2] * p[2];
                if (!std::isfinite(w[k])) finite = false;
                lo[k] = std::min(lo[k], w[k]), hi[k] = std::max(hi[k], w[k]);
            }
        };
        if (L.kind == RTAMD_MESH) {
            for (int t = L.tri_start; finite && t < L.tri_start + L.tri_count; t++) {
                const double* g = s->tri_geom + (size_t)t * 9;
                for (int v = 0; v < 3; v++) {
                    double q[3];
                    for (int k = 0; k < 3; k++) q[k] = v == 0 ? g[k] : g[k] + g[3 * v + k];  // p1, p1 + e1, p1 + e2
                    add_point(q);
                }
            }
        } else {
            double olo[3] = {-1, -1, -1}, ohi[3] = {1, 1, 1};  // sphere, cube
            if (L.kind == RTAMD_PLANE) finite = false;
            if (L.kind == RTAMD_CYLINDER || L.kind == RTAMD_CONE) {
                if (!std::isfinite(L.p0) || !std::isfinite(L.p1)) finite = false;
                olo[1] = L.p0, ohi[1] = L.p1;
                if (L.kind == RTAMD_CONE) ohi[0] = ohi[2] = std::max(std::fabs(L.p0), std::fabs(L.p1)), olo[0] = olo[2] = -ohi[0];
            }
            for (int corner = 0; finite && corner < 8; corner++) {
                double q[3];
                for (int k = 0; k < 3; k++) q[k] = (corner >> k) & 1 ? ohi[k] : olo[k];
                add_point(q);
            }
        }
        // Runs of triangles carry their bounds themselves.
        bounded[(size_t)i] = finite && L.kind != RTAMD_MESH;
        has_box[(size_t)i] = finite;
        for (int k = 0; k < 3; k++) {
            // generous against rounding in this inverse and in the triangle test; no bound at all for odd data
            const double pad = finite ? 1e-6 * std::max({hi[k] - lo[k], std::fabs(lo[k]), std::fabs(hi[k]), 1e-30}) : INFINITY;
            boxes[(size_t)i * 6 + k] = finite ? lo[k] - pad : -INFINITY;
            boxes[(size_t)i * 6 + 3 + k] = finite ? hi[k] + pad : INFINITY;
        }
    }
    if ((rc = upload(ctx, boxes.data(), boxes.size(), &d.mesh_box))) return rc;
    // Runs of triangles for the traversal.  Consecutive MESH leaves that differ in nothing but their triangles (one
    // object cut into runs by the host) are one pool of triangles, and nothing says which triangle has to be tested
    // in which run: the pool is cut again by median splits along the longest axis of its centroids' bounds, so that
    // a run's box is compact whatever order the file had.  The traversal reads this copy of the triangles; every
    // triangle keeps the intersection key of its place in the original order (leaf << RTAMD_KEY_SHIFT | slot), which
    // is what decides between equal t (src/scene.rs:19-26: stable sorts) and what shading looks the triangle up by.
    // CSG children are tested from the leaf table and keep the original runs.
    // Diagnostic (measurements only): the traversal as the reference's World::intersect does it, every leaf and every
    // triangle for every ray -- no pools cut again, no bounds on runs of triangles, no bounds records.
    const bool brute = getenv("RTAMD_BRUTE_FORCE") != nullptr;
    std::vector<double> tri_sorted(s->tri_geom, s->tri_geom + (size_t)s->n_tris * 9);
    std::vector<int32_t> tri_key((size_t)s->n_tris, 0);
    // A MESH leaf is tested as one run of triangles -- unless it is part of a pool that is cut again below: then the pool's
    // first leaf stands for all of its runs (as many records as the cut gives, however many leaves the host made), and
    // the other leaves of the pool get no record of their own.
    struct Run {
        int32_t start, count;
        double box[6];
    };
    std::vector<std::vector<Run>> runs_of((size_t)s->n_leaves);  // per leaf: its runs (empty: a pooled leaf that is not the first)
    // How many triangles the device tests as one run: 64 fill a wave's lanes (one triangle each when few rays come
    // near), but 48 have tighter boxes, which pays when the frame is large enough to be bound by throughput (teapot room:
    // 3.87 against 4.03 ms at 3840x2160, 1.41 against 1.44 at 1920x1080); a small frame waits for its longest chain of
    // rays, and every record more is a step more in it (0.85 against 0.89 ms at 400x200).  Same pixels either way.
    int run_cap = (long long)s->camera.hsize * (long long)s->camera.vsize >= 2000000 ? 48 : 64;
    if (const char* e = getenv("RTAMD_MESH_RUN")) run_cap = std::max(1, std::min(64, atoi(e)));  // (diagnostic)
    // where every plain leaf stands in the host's visit list (a pool's leaves must stand there side by side, in order)
    std::vector<int32_t> visit_of_leaf((size_t)s->n_leaves, -1);
    for (int i = 0; i < s->n_visits; i++)
        if (!(s->visits[i].flags & RTAMD_VISIT_CSG)) visit_of_leaf[(size_t)s->visits[i].unit] = i;
    {
        std::vector<char> in_csg((size_t)s->n_leaves, 0);
        for (int r = 0; r < s->n_csg_roots; r++)
            for (int l = s->csg_roots[r].leaf_lo; l < s->csg_roots[r].leaf_hi; l++) in_csg[(size_t)l] = 1;
        for (int i = 0; i < s->n_leaves; i++) {
            const rtamd_leaf& L = s->leaves[i];
            if (L.kind != RTAMD_MESH) continue;
            {
                Run r{L.tri_start, L.tri_count, {0, 0, 0, 0, 0, 0}};
                std::memcpy(r.box, &boxes[(size_t)i * 6], sizeof r.box);
                runs_of[(size_t)i].push_back(r);
            }
            for (int k = 0; k < L.tri_count; k++) tri_key[(size_t)L.tri_start + k] = (int32_t)(((uint32_t)i << RTAMD_KEY_SHIFT) | (uint32_t)k);
        }
        const std::vector<int32_t> orig_key(tri_key);
        auto same_object = [&](int a, int b) {
            const rtamd_leaf &A = s->leaves[a], &B = s->leaves[b];
            return B.kind == RTAMD_MESH && !in_csg[(size_t)b] && has_box[(size_t)b] && std::memcmp(A.w2o, B.w2o, sizeof A.w2o) == 0 &&
                   A.flags == B.flags && A.material == B.material && A.eq_class == B.eq_class && A.chain_len == B.chain_len &&
                   std::memcmp(s->chain + (size_t)A.chain_start * 12, s->chain + (size_t)B.chain_start * 12, (size_t)A.chain_len * 12 * sizeof(double)) == 0 &&
                   std::max(A.ray_len, 1) == std::max(B.ray_len, 1) &&
                   (A.ray_len <= 1 || std::memcmp(s->chain + (size_t)A.ray_start * 12, s->chain + (size_t)B.ray_start * 12, (size_t)A.ray_len * 12 * sizeof(double)) == 0);
        };
        for (int i = 0; i < s->n_leaves;) {
            const rtamd_leaf& L = s->leaves[i];
            int j = i + 1;
            if (L.kind == RTAMD_MESH && !in_csg[(size_t)i] && has_box[(size_t)i])
                while (j < s->n_leaves && same_object(i, j) && s->leaves[j].tri_start == s->leaves[j - 1].tri_start + s->leaves[j - 1].tri_count &&
                       visit_of_leaf[(size_t)i] >= 0 && visit_of_leaf[(size_t)j] == visit_of_leaf[(size_t)i] + (j - i))
                    j++;
            const int first = L.tri_start, total = s->leaves[j - 1].tri_start + s->leaves[j - 1].tri_count - first;
            const int cap = run_cap;
            if (brute || L.kind != RTAMD_MESH || in_csg[(size_t)i] || !has_box[(size_t)i] || (j - i < 2 && total <= cap)) {
                i = j;
                continue;
            }
            // object -> world for the corners (as for the boxes above)
            const double* m = L.w2o;
            const double det = m[0] * (m[5] * m[10] - m[6] * m[9]) - m[1] * (m[4] * m[10] - m[6] * m[8]) + m[2] * (m[4] * m[9] - m[5] * m[8]);
            const double inv[9] = {(m[5] * m[10] - m[6] * m[9]) / det, (m[2] * m[9] - m[1] * m[10]) / det, (m[1] * m[6] - m[2] * m[5]) / det,
                                   (m[6] * m[8] - m[4] * m[10]) / det, (m[0] * m[10] - m[2] * m[8]) / det, (m[2] * m[4] - m[0] * m[6]) / det,
                                   (m[4] * m[9] - m[5] * m[8]) / det, (m[1] * m[8] - m[0] * m[9]) / det, (m[0] * m[5] - m[1] * m[4]) / det};
            std::vector<double> tlo((size_t)total * 3), thi((size_t)total * 3), cen((size_t)total * 3);
            for (int t = 0; t < total; t++) {
                const double* g = s->tri_geom + ((size_t)first + t) * 9;
                for (int k = 0; k < 3; k++) tlo[(size_t)t * 3 + k] = INFINITY, thi[(size_t)t * 3 + k] = -INFINITY;
                for (int v = 0; v < 3; v++) {
                    double p[3];
                    for (int k = 0; k < 3; k++) p[k] = (v == 0 ? g[k] : g[k] + g[3 * v + k]) - m[4 * k + 3];
                    for (int k = 0; k < 3; k++) {
                        const double w = inv[3 * k] * p[0] + inv[3 * k + 1] * p[1] + inv[3 * k + 2] * p[2];
                        tlo[(size_t)t * 3 + k] = std::min(tlo[(size_t)t * 3 + k], w), thi[(size_t)t * 3 + k] = std::max(thi[(size_t)t * 3 + k], w);
                    }
                }
                for (int k = 0; k < 3; k++) cen[(size_t)t * 3 + k] = 0.5 * (tlo[(size_t)t * 3 + k] + thi[(size_t)t * 3 + k]);
            }
            std::vector<int> order((size_t)total);
            for (int t = 0; t < total; t++) order[(size_t)t] = t;
            std::vector<std::pair<int, int>> runs;  // [begin, end) in `order`
            std::function<void(int, int)> split = [&](int b, int e) {
                if (e - b <= cap) {
                    runs.push_back({b, e});
                    return;
                }
                // Of the three axes and of every cut that leaves whole runs on the left (no more runs than before), the
                // one with the least sum over both sides of box surface x triangles: what a ray that enters a box pays.
                auto surface = [](const double* lo, const double* hi) {
                    const double dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
                    return 2.0 * (dx * dy + dy * dz + dx * dz);
                };
                const int n = e - b;
                double best_cost = INFINITY;
                int best_axis = 0, half = cap;
                std::vector<int> sorted_by[3];
                std::vector<double> right((size_t)n + 1);
                for (int ax = 0; ax < 3; ax++) {
                    std::vector<int>& o = sorted_by[ax];
                    o.assign(order.begin() + b, order.begin() + e);
                    std::stable_sort(o.begin(), o.end(), [&](int x, int y) { return cen[(size_t)x * 3 + ax] < cen[(size_t)y * 3 + ax]; });
                    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
                    for (int q = n - 1; q >= 0; q--) {  // right[q]: surface of the box of o[q..n)
                        for (int k = 0; k < 3; k++) lo[k] = std::min(lo[k], tlo[(size_t)o[(size_t)q] * 3 + k]), hi[k] = std::max(hi[k], thi[(size_t)o[(size_t)q] * 3 + k]);
                        right[(size_t)q] = surface(lo, hi);
                    }
                    for (int k = 0; k < 3; k++) lo[k] = INFINITY, hi[k] = -INFINITY;
                    for (int q = 0; q + 1 < n; q++) {
                        for (int k = 0; k < 3; k++) lo[k] = std::min(lo[k], tlo[(size_t)o[(size_t)q] * 3 + k]), hi[k] = std::max(hi[k], thi[(size_t)o[(size_t)q] * 3 + k]);
                        if ((q + 1) % cap != 0) continue;
                        const double cost = surface(lo, hi) * (q + 1) + right[(size_t)q + 1] * (n - q - 1);
                        if (cost < best_cost) best_cost = cost, best_axis = ax, half = q + 1;
                    }
                }
                std::copy(sorted_by[best_axis].begin(), sorted_by[best_axis].end(), order.begin() + b);
                split(b, b + half);
                split(b + half, e);
            };
            split(0, total);
            if (getenv("RTAMD_UPLOAD_DUMP")) {  // diagnostic
                auto area = [&](int l, const std::vector<double>& bx) {
                    const double* q = &bx[(size_t)l * 6];
                    const double dx = q[3] - q[0], dy = q[4] - q[1], dz = q[5] - q[2];
                    return dx > 0 ? 2 * (dx * dy + dy * dz + dx * dz) : 0.0;
                };
                double before = 0;
                for (int l = i; l < j; l++) before += area(l, boxes);
                fprintf(stderr, "[rtamd upload] pool of %d triangles in leaves %d..%d: %zu runs of <= %d, box area before %.4f\n", total, i, j - 1,
                        runs.size(), cap, before);
            }
            int pos = first;
            std::vector<Run> cut;
            for (const auto& rr : runs) {
                Run r{pos, rr.second - rr.first, {0, 0, 0, 0, 0, 0}};
                double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
                for (int q = rr.first; q < rr.second; q++, pos++) {
                    const int t = order[(size_t)q];
                    std::memcpy(&tri_sorted[(size_t)pos * 9], s->tri_geom + ((size_t)first + t) * 9, 9 * sizeof(double));
                    tri_key[(size_t)pos] = orig_key[(size_t)first + t];
                    for (int k = 0; k < 3; k++) lo[k] = std::min(lo[k], tlo[(size_t)t * 3 + k]), hi[k] = std::max(hi[k], thi[(size_t)t * 3 + k]);
                }
                for (int k = 0; k < 3; k++) {
                    const double pad = 1e-6 * std::max({hi[k] - lo[k], std::fabs(lo[k]), std::fabs(hi[k]), 1e-30});
                    r.box[k] = lo[k] - pad, r.box[3 + k] = hi[k] + pad;
                }
                cut.push_back(r);
            }
            for (int l = i; l < j; l++) runs_of[(size_t)l].clear();
            runs_of[(size_t)i] = cut;
            if (getenv("RTAMD_UPLOAD_DUMP")) {
                double after = 0;
                for (const Run& r : runs_of[(size_t)i]) {
                    const double dx = r.box[3] - r.box[0], dy = r.box[4] - r.box[1], dz = r.box[5] - r.box[2];
                    after += dx > 0 ? 2 * (dx * dy + dy * dz + dx * dz) : 0.0;
                }
                fprintf(stderr, "[rtamd upload]   box area after %.4f\n", after);
            }
            i = j;
        }
    }
    if ((rc = upload(ctx, tri_sorted.data(), tri_sorted.size(), &d.tri_sorted))) return rc;
    if ((rc = upload(ctx, tri_key.data(), tri_key.size(), &d.tri_key))) return rc;
    // The pixel rectangle (x0, x1, y0, y1) that bounds what the camera sees of a world-space box; false when there is
    // none (part of the box at or behind the camera, degenerate camera).
    const rtamd_camera& C = s->camera;
    const double* ci = C.inv;  // camera -> world; its inverse takes a world point to the camera's frame
    const double cdet = ci[0] * (ci[5] * ci[10] - ci[6] * ci[9]) - ci[1] * (ci[4] * ci[10] - ci[6] * ci[8]) + ci[2] * (ci[4] * ci[9] - ci[5] * ci[8]);
    double w2c[9] = {0};
    const bool cam_ok = std::isfinite(cdet) && cdet != 0.0;
    if (cam_ok) {
        w2c[0] = (ci[5] * ci[10] - ci[6] * ci[9]) / cdet, w2c[1] = (ci[2] * ci[9] - ci[1] * ci[10]) / cdet, w2c[2] = (ci[1] * ci[6] - ci[2] * ci[5]) / cdet;
        w2c[3] = (ci[6] * ci[8] - ci[4] * ci[10]) / cdet, w2c[4] = (ci[0] * ci[10] - ci[2] * ci[8]) / cdet, w2c[5] = (ci[2] * ci[4] - ci[0] * ci[6]) / cdet;
        w2c[6] = (ci[4] * ci[9] - ci[5] * ci[8]) / cdet, w2c[7] = (ci[1] * ci[8] - ci[0] * ci[9]) / cdet, w2c[8] = (ci[0] * ci[5] - ci[1] * ci[4]) / cdet;
    }
    auto screen_rect = [&](const double* box, double r[4]) {
        if (!cam_ok) return false;
        r[0] = r[2] = INFINITY, r[1] = r[3] = -INFINITY;
        for (int corner = 0; corner < 8; corner++) {
            double pw[3], pc[3];
            for (int k = 0; k < 3; k++) pw[k] = box[((corner >> k) & 1 ? 3 : 0) + k] - ci[4 * k + 3];
            for (int k = 0; k < 3; k++) pc[k] = w2c[3 * k] * pw[0] + w2c[3 * k + 1] * pw[1] + w2c[3 * k + 2] * pw[2];
            if (!(pc[2] < -1e-9)) return false;
            const double px = (C.half_width - pc[0] / -pc[2]) / C.pixel_size, py = (C.half_height - pc[1] / -pc[2]) / C.pixel_size;
            if (!std::isfinite(px) || !std::isfinite(py)) return false;
            r[0] = std::min(r[0], px), r[1] = std::max(r[1], px), r[2] = std::min(r[2], py), r[3] = std::max(r[3], py);
        }
        return true;
    };
    if ((rc = upload(ctx, s->chain, (size_t)s->n_chain * 12, &d.chain))) return rc;
    if ((rc = upload(ctx, ray_levels.data(), ray_levels.size(), &d.ray_levels))) return rc;
    if ((rc = upload(ctx, s->tri_geom, (size_t)s->n_tris * 9, &d.tri_geom))) return rc;
    if ((rc = upload(ctx, s->tri_normals, (size_t)s->n_tris * 9, &d.tri_normals))) return rc;
    if ((rc = upload(ctx, s->materials, (size_t)s->n_materials, &d.materials))) return rc;
    if ((rc = upload(ctx, s->csg_nodes, (size_t)s->n_csg_nodes, &d.csg_nodes))) return rc;
    if ((rc = upload(ctx, s->csg_roots, (size_t)s->n_csg_roots, &d.csg_roots))) return rc;
    {
        // slot tables of small CSG objects (see DCsgSlot); larger objects get none and use the sorted lists
        std::vector<DCsgSlot> slots;
        std::vector<DCsgSlotNode> snodes((size_t)s->n_csg_nodes, DCsgSlotNode{0, 0, 0, 0});
        std::vector<int32_t> slot0((size_t)s->n_csg_roots, 0);
        for (int r = 0; csg_small && r < s->n_csg_roots; r++) {
            const rtamd_csg_root& R = s->csg_roots[r];
            slot0[(size_t)r] = (int32_t)slots.size();
            std::vector<int> first((size_t)(R.leaf_hi - R.leaf_lo) + 1, 0);  // first slot of every leaf, relative to the root
            for (int l = R.leaf_lo; l < R.leaf_hi; l++) {
                const rtamd_leaf& L = s->leaves[l];
                const int n = L.kind == RTAMD_PLANE ? 1 : L.kind == RTAMD_MESH ? L.tri_count : (L.kind == RTAMD_CYLINDER || L.kind == RTAMD_CONE) ? 4 : 2;
                first[(size_t)(l - R.leaf_lo) + 1] = first[(size_t)(l - R.leaf_lo)] + n;
                for (int k = 0; k < n; k++)
                    slots.push_back(DCsgSlot{(int32_t)(((uint32_t)l << RTAMD_KEY_SHIFT) | (uint32_t)k), L.eq_class, (L.flags & RTAMD_LEAF_SHADOW) ? 1u : 0u, 0u});
            }
            auto bits = [&](int la, int lb) {  // slots of leaves [la, lb)
                const int a = first[(size_t)(la - R.leaf_lo)], b = first[(size_t)(lb - R.leaf_lo)];
                return (b >= 32 ? 0xffffffffu : ((1u << b) - 1u)) & ~((1u << a) - 1u);
            };
            for (int q = R.node_start; q < R.node_start + R.node_count; q++) {
                const rtamd_csg_node& N = s->csg_nodes[q];
                DCsgSlotNode& D = snodes[(size_t)q];
                D.range = bits(N.leaf_lo, N.leaf_hi), D.op = N.op, D.left = 0;
                for (int l = N.leaf_lo; l < N.leaf_hi; l++)
                    if ((N.left_mask >> (l - N.leaf_lo)) & 1ull) D.left |= bits(l, l + 1);
            }
        }
        if ((rc = upload(ctx, slots.data(), slots.size(), &d.csg_slots))) return rc;
        if ((rc = upload(ctx, snodes.data(), snodes.size(), &d.csg_slot_nodes))) return rc;
        if ((rc = upload(ctx, slot0.data(), slot0.size(), &d.csg_slot0))) return rc;
    }
    std::vector<DVisit> dv;
    std::vector<int32_t> class_mat((size_t)std::max(s->n_classes, 1), 0);
    double diag_thr = 0.0;
    // The record of a CSG object: bounds of everything its operations can keep (no bounds at all when that takes in a leaf
    // that has none, or a cone: see the bounds records below).  What the filter of src/shapes/csg.rs:54-82 keeps of a
    // difference are hits on the left child and hits of the right child between an odd and an even hit of the left one; of
    // an intersection, hits of either child between such hits of the other.  A sphere or a cube gives a ray none or two
    // hits (src/shapes/sphere.rs:40-55, src/shapes/cube.rs:46-66), with nothing but its inside between them: where the
    // left child of a difference, or either child of an intersection, is such a leaf, its box holds all that is kept --
    // whatever the other child is, a plane included.  Everything else (unions, groups, children that are cylinders or
    // objects themselves) gets the bounds of its leaves together.
    struct CsgBox {
        double lo[3], hi[3];
        bool finite, pair;
    };
    std::function<CsgBox(int, int, int, int)> csg_range_box = [&](int root, int a, int b, int below) {
        const rtamd_csg_root& R = s->csg_roots[root];
        int node = -1;  // the CSG node that is this range of leaves (children come before their parents: post-order)
        for (int q = R.node_start; q < below; q++)
            if (s->csg_nodes[q].leaf_lo == a && s->csg_nodes[q].leaf_hi == b && b > a) node = q;
        CsgBox B{{INFINITY, INFINITY, INFINITY}, {-INFINITY, -INFINITY, -INFINITY}, true, false};
        auto take_in = [&](const CsgBox& C) {
            B.finite = B.finite && C.finite;
            for (int k = 0; k < 3; k++) B.lo[k] = std::min(B.lo[k], C.lo[k]), B.hi[k] = std::max(B.hi[k], C.hi[k]);
        };
        if (node < 0) {  // a leaf, or the leaves of a group
            for (int l = a; l < b; l++) {
                CsgBox C{};
                for (int k = 0; k < 3; k++) C.lo[k] = boxes[(size_t)l * 6 + k], C.hi[k] = boxes[(size_t)l * 6 + 3 + k];
                C.finite = has_box[(size_t)l] && s->leaves[l].kind != RTAMD_CONE;
                take_in(C);
            }
            B.pair = b - a == 1 && B.finite && (s->leaves[a].kind == RTAMD_SPHERE || s->leaves[a].kind == RTAMD_CUBE);
            return B;
        }
        const rtamd_csg_node& N = s->csg_nodes[node];
        const CsgBox L = csg_range_box(root, N.leaf_lo, N.leaf_mid, node), Rt = csg_range_box(root, N.leaf_mid, N.leaf_hi, node);
        // (`left.includes`, src/shapes/csg.rs:64, goes by value: a leaf of the right child that equals one of the left child
        // counts as a hit of the left child wherever the groups above it put it -- then the leaves' bounds together)
        const int n_left = N.leaf_mid - N.leaf_lo;
        const bool by_place = n_left >= 64 || (N.left_mask >> n_left) == 0;
        if (by_place && N.op == RTAMD_CSG_DIFFERENCE && L.pair) {
            B = L;
        } else if (by_place && N.op == RTAMD_CSG_INTERSECTION && (L.pair || Rt.pair)) {
            B = L.pair ? L : Rt;
            if (L.pair && Rt.pair) {
                CsgBox C = B;
                for (int k = 0; k < 3; k++) C.lo[k] = std::max(L.lo[k], Rt.lo[k]), C.hi[k] = std::min(L.hi[k], Rt.hi[k]);
                if (C.lo[0] <= C.hi[0] && C.lo[1] <= C.hi[1] && C.lo[2] <= C.hi[2]) B = C;  // (boxes apart: nothing is kept; either box will do)
            }
        } else {
            take_in(L), take_in(Rt);
        }
        B.pair = false;
        return B;
    };
    auto csg_record = [&](DVisit& D, int root) {
        const rtamd_csg_root& R = s->csg_roots[root];
        const CsgBox B = getenv("RTAMD_CSG_LOOSE_BOX") ? csg_range_box(root, R.leaf_lo, R.leaf_hi, R.node_start)  // (diagnostic: the leaves' bounds together)
                                                      : csg_range_box(root, R.leaf_lo, R.leaf_hi, R.node_start + R.node_count);
        for (int k = 0; k < 3; k++) D.w2o[k] = B.finite ? B.lo[k] : -INFINITY, D.w2o[3 + k] = B.finite ? B.hi[k] : INFINITY;
    };
    auto leaf_record = [&](DVisit& D, int leaf, const Run* run) {  // run: of a MESH leaf, which of its runs the record is for
        const DLeaf& L = dl[leaf];
        std::memcpy(D.w2o, L.w2o, sizeof D.w2o);
        if (L.flags & DLEAF_DIAG) {
            D.w2o[1] = L.inv_diag[0], D.w2o[2] = L.inv_diag[1], D.w2o[4] = L.inv_diag[2];
            D.w2o[6] = -L.w2o[3] * L.inv_diag[0], D.w2o[8] = -L.w2o[7] * L.inv_diag[1], D.w2o[9] = -L.w2o[11] * L.inv_diag[2];
            if (L.kind == RTAMD_CUBE)
                for (int k = 0; k < 3; k++) diag_thr = std::max(diag_thr, 1e-5 * std::fabs(L.inv_diag[k]) * (1.0 + 1e-9));
        }
        D.p0 = L.p0, D.p1 = L.p1;
        if (L.kind == RTAMD_MESH) {
            int32_t tr[2] = {run->start, run->count};  // in tri_sorted / tri_key
            std::memcpy(&D.p0, tr, sizeof tr);
            // the traversal needs the bounds for every run and the matrix only for the runs a ray comes near:
            // the bounds take the matrix's place in the record, the matrix stays in the leaf table
            std::memcpy(D.w2o, run->box, 6 * sizeof(double));
            if (brute)
                for (int k = 0; k < 3; k++) D.w2o[k] = -INFINITY, D.w2o[3 + k] = INFINITY;
            for (int k = 6; k < 12; k++) D.w2o[k] = 0.0;
        }
        D.kind_flags = (uint32_t)L.kind | (L.flags << 8);
    };
    dv.clear();
    for (int i = 0; i < s->n_visits; i++) {
        const rtamd_visit& V = s->visits[i];
        class_mat[V.eq_class] = V.class_material;
        DVisit D;
        std::memset(&D, 0, sizeof D);
        D.unit = V.unit;
        D.vflags_class = (V.flags & 0xffu) | ((uint32_t)V.eq_class << 8);
        if (V.flags & RTAMD_VISIT_CSG) {
            csg_record(D, V.unit);
            dv.push_back(D);
        } else if (dl[(size_t)V.unit].kind != RTAMD_MESH) {
            leaf_record(D, V.unit, nullptr);
            dv.push_back(D);
        } else if (runs_of[(size_t)V.unit].empty()) {
            // a leaf of a pool other than its first: the pool's records stand at the first leaf's place; what this visit
            // says about the end of the class goes to the last of them (the pool's visits stand side by side)
            if ((V.flags & RTAMD_VISIT_CLASS_END) && !dv.empty()) dv.back().vflags_class |= RTAMD_VISIT_CLASS_END;
        } else {
            const std::vector<Run>& rs = runs_of[(size_t)V.unit];
            for (size_t q = 0; q < rs.size(); q++) {
                DVisit R = D;
                // (only the last record of a leaf may close the class)
                if (q + 1 < rs.size()) R.vflags_class &= ~(uint32_t)RTAMD_VISIT_CLASS_END;
                leaf_record(R, V.unit, &rs[q]);
                dv.push_back(R);
            }
        }
    }
    // the same units once each, in depth-first order (what traversals without a container pass walk)
    std::vector<int> root_of((size_t)s->n_leaves, -1);
    for (int r = 0; r < s->n_csg_roots; r++)
        for (int l = s->csg_roots[r].leaf_lo; l < s->csg_roots[r].leaf_hi; l++) root_of[l] = r;
    std::vector<DVisit> dv_dfs;
    for (int i = 0; i < s->n_leaves; i++) {
        DVisit D;
        std::memset(&D, 0, sizeof D);
        if (root_of[i] < 0) {
            D.unit = i;
            D.vflags_class = RTAMD_VISIT_FIRST | ((uint32_t)dl[i].eq_class << 8);
            if (dl[(size_t)i].kind == RTAMD_MESH) {
                for (const Run& r : runs_of[(size_t)i]) {  // (none for a pooled leaf that is not the pool's first)
                    DVisit R = D;
                    leaf_record(R, i, &r);
                    dv_dfs.push_back(R);
                }
                continue;
            }
            leaf_record(D, i, nullptr);
        } else {
            if (i > 0 && root_of[i - 1] == root_of[i]) continue;
            D.unit = root_of[i];
            D.vflags_class = RTAMD_VISIT_FIRST | RTAMD_VISIT_CSG;
            csg_record(D, D.unit);
        }
        dv_dfs.push_back(D);
    }
    // Runs of records of bounded leaves get bounds records in front, as a binary tree over the run in the order it has
    // (the order decides ties and groups the equality classes, so it stays): every node is cut where the two sides have
    // the least box surface x cost, down to pairs of records, and a node gets a bounds record when its box is clearly
    // smaller than its parent's and the records below it cost several times the test.  A bounds record
    // carries the union of the bounds below it and the number of records (nested bounds records included) a wave may
    // skip when none of its rays meets them.  A record that closes an equality class is skipped only when it is that whole class.
    // A ray nearly parallel to a cone's side gets an intersection that may lie outside the cone's bounds (see the kernel).
    // "Nearly parallel" is a quadratic form of the world-space direction d: with M the 3x3 part of the cone's world ->
    // object matrix, a = (M d)_x^2 - (M d)_y^2 + (M d)_z^2 = d^T Q d, Q = M^T diag(1, -1, 1) M.  The scene's distinct Q
    // (cones that differ by a turn about their own axis, or point the other way, share one) are numbered; a ray works
    // its 64-bit set of risky orientations out once per traversal, and a bounds record carries the set of those below it
    // (orientation k -> bit k % 64).  Per orientation: Qxx, Qyy, Qzz, 2 Qxy, 2 Qxz, 2 Qyz, the Frobenius norm of Q, one spare double.
    std::vector<double> cone_quads;
    auto cone_bit = [&](const double* m) {
        const double M[3][3] = {{m[0], m[1], m[2]}, {m[4], m[5], m[6]}, {m[8], m[9], m[10]}};
        double Q[3][3];
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) Q[i][j] = M[0][i] * M[0][j] - M[1][i] * M[1][j] + M[2][i] * M[2][j];
        const double q6[6] = {Q[0][0], Q[1][1], Q[2][2], 2.0 * Q[0][1], 2.0 * Q[0][2], 2.0 * Q[1][2]};
        double big = 0.0;
        for (double v : q6) big = std::max(big, std::fabs(v));
        size_t k = 0;
        for (; k < cone_quads.size() / 8; k++) {  // (the kernel's margin is 4 EPSILON against the cone test's own EPSILON: 1e-9 is lost in it)
            bool same = true;
            for (int i = 0; i < 6; i++) same = same && std::fabs(cone_quads[k * 8 + (size_t)i] - q6[i]) <= 1e-9 * big;
            if (same) break;
        }
        if (k == cone_quads.size() / 8) {
            cone_quads.insert(cone_quads.end(), q6, q6 + 6);
            double fro = 0.0;  // Frobenius norm of Q (>= its largest eigenvalue), rounded up
            for (int i = 0; i < 3; i++)
                for (int j = 0; j < 3; j++) fro += Q[i][j] * Q[i][j];
            cone_quads.push_back(std::sqrt(fro) * (1.0 + 1e-9));
            cone_quads.push_back(0.0);
        }
        return (uint64_t)1 << (k % 64);
    };
    auto with_bounds = [&](const std::vector<DVisit>& in) {
        if (brute) return in;
        // A record that closes a class may be skipped only when it is the whole class (the record before it
        // closed the previous one): the container pass then has nothing pending when it is passed over.
        std::vector<char> groupable(in.size(), 0);
        for (size_t i = 0; i < in.size(); i++) {
            const DVisit& v = in[i];
            const bool mesh = !(v.vflags_class & RTAMD_VISIT_CSG) && ((v.kind_flags & 0xffu) == RTAMD_MESH || bounded[(size_t)v.unit]);
            const bool closes = (v.vflags_class & RTAMD_VISIT_CLASS_END) != 0;
            const bool alone = closes && (i == 0 || (in[i - 1].vflags_class & RTAMD_VISIT_CLASS_END));
            groupable[i] = mesh && (!closes || alone);
        }
        auto is_mesh = [&](const DVisit& v) { return groupable[(size_t)(&v - in.data())] != 0; };
        auto bounds_record = [&](const std::vector<DVisit>& body) {
            DVisit B;
            std::memset(&B, 0, sizeof B);
            double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
            // the kernel looks closer before it passes over a cone (rtamd_kernel.inc): which of the scene's cone orientations
            // (cone_quads, one bit each) lie under this record
            uint64_t cone_mask = 0;
            for (const DVisit& v : body) {
                if ((v.kind_flags & 0xffu) == RTAMD_CONE) {
                    cone_mask |= cone_bit(dl[(size_t)v.unit].w2o);
                } else if ((v.kind_flags & 0xffu) == DVISIT_BOUNDS && ((v.kind_flags >> 8) & DBOUNDS_HAS_CONE)) {
                    uint64_t below;
                    std::memcpy(&below, &v.p0, sizeof below);
                    cone_mask |= below;
                }
                const uint32_t vk = v.kind_flags & 0xffu;  // bounds records and runs of triangles hold their box themselves
                const double* b = vk == DVISIT_BOUNDS || vk == RTAMD_MESH ? v.w2o : &boxes[(size_t)v.unit * 6];
                for (int k = 0; k < 3; k++) lo[k] = std::min(lo[k], b[k]), hi[k] = std::max(hi[k], b[3 + k]);
            }
            for (int k = 0; k < 3; k++) B.w2o[k] = lo[k], B.w2o[3 + k] = hi[k];
            // the same box as centre and half extent (what the kernel tests: three fused operations per axis), rounded outwards
            for (int k = 0; k < 3; k++) {
                const double c = 0.5 * lo[k] + 0.5 * hi[k];
                B.w2o[6 + k] = c;
                B.w2o[9 + k] = std::max(c - lo[k], hi[k] - c) * (1.0 + 0x1p-48) + 0x1p-1000;
            }
            B.kind_flags = DVISIT_BOUNDS;
            if (cone_mask != 0) {
                B.kind_flags |= DBOUNDS_HAS_CONE << 8;
                std::memcpy(&B.p0, &cone_mask, sizeof cone_mask);
            }
            B.unit = (int32_t)body.size();
            return B;
        };
        // A bounds record costs about as much as two tests of an axis-aligned cube; it is worth having when
        // what it can save is about that.  Rough costs in instructions per wave.
        // A wave passes over the records only if all of its rays miss the box; the share of the frame outside the
        // box's picture says how often that can happen for the rays that start at the camera.
        const double kBoundsCost = 45.0;
        // (1.0 would be break-even for the rays that start at the camera; measured on one GPU: 0.75 and 1.0 beat 2.0 by 5 % on
        // cover.yaml, whose cubes then get records over clusters of 2 to 11, 0.75 beats 1.0 by 5 % on the tables scene, and
        // the other scenes do not care; 0 -- a record at every node -- 1.25 and more lose)
        // A frame with fewer groups of pixels than the GPU has wave slots waits for its longest chain of dependent rays, and
        // those rays live among the objects: they meet the boxes, and a record they meet is one more step in the chain.
        // Records then have to save more to be worth a place (reflect_refract 400x200: 0.39 -> 0.35 ms, as with none at all;
        // a mesh without records is 2.7 times slower).
        const bool small_frame = (int64_t)((C.hsize + 7) / 8) * ((C.vsize + 7) / 8) <= (int64_t)16 * ctx->num_cus;
        const double kWorthIt = getenv("RTAMD_BOUNDS_WORTH") ? atof(getenv("RTAMD_BOUNDS_WORTH")) : small_frame ? 6.0 : 0.75;  // (diagnostic override)
        auto outside = [&](const DVisit& B) {
            double r[4];
            if (!screen_rect(B.w2o, r)) return 0.0;
            const double w = std::max(0.0, std::min(r[1], (double)C.hsize) - std::max(r[0], 0.0));
            const double h = std::max(0.0, std::min(r[3], (double)C.vsize) - std::max(r[2], 0.0));
            return 1.0 - w * h / ((double)C.hsize * (double)C.vsize);
        };
        auto cost = [&](const DVisit& v) {
            const uint32_t kind = v.kind_flags & 0xffu;
            if (kind == DVISIT_BOUNDS) return 0.0;
            if (kind == RTAMD_MESH) return 150.0;
            if (kind == RTAMD_CUBE && ((v.kind_flags >> 8) & DLEAF_DIAG)) return 25.0;
            return 120.0;
        };
        auto total = [&](const std::vector<DVisit>& body) {
            double c = 0.0;
            for (const DVisit& v : body) c += cost(v);
            return c;
        };
        std::vector<DVisit> out;
        size_t i = 0;
        while (i < in.size()) {
            if (!is_mesh(in[i])) {
                out.push_back(in[i++]);
                continue;
            }
            // a run of triangle records, or a run of other bounded leaves: not mixed, the triangles of one object
            // lie together and share little with the shapes around them
            const bool tri_run = (in[i].kind_flags & 0xffu) == RTAMD_MESH;
            size_t j = i;
            while (j < in.size() && is_mesh(in[j]) && ((in[j].kind_flags & 0xffu) == RTAMD_MESH) == tri_run) j++;
            {
                auto box_of = [&](const DVisit& v) {
                    const uint32_t vk = v.kind_flags & 0xffu;
                    return vk == DVISIT_BOUNDS || vk == RTAMD_MESH ? v.w2o : &boxes[(size_t)v.unit * 6];
                };
                auto area = [](const double* lo, const double* hi) {
                    const double dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
                    return 2.0 * (dx * dy + dy * dz + dz * dx);
                };
                // (measured over the mesh, group, table and cylinder scenes: leaves of 1 to 3 records and thresholds of
                // 0.7 to 0.9 are within 2 % of each other; leaves of 8 or a threshold of 0.5 lose 10 to 30 %)
                const size_t leaf_max = 2;
                const double shrink = 0.8;
                std::function<std::vector<DVisit>(size_t, size_t, double)> build = [&](size_t lo_i, size_t hi_i, double parent_area) {
                    std::vector<DVisit> body;
                    const size_t n = hi_i - lo_i;
                    double node_area = parent_area;
                    {
                        double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
                        for (size_t q = lo_i; q < hi_i; q++) {
                            const double* b = box_of(in[q]);
                            for (int k = 0; k < 3; k++) lo[k] = std::min(lo[k], b[k]), hi[k] = std::max(hi[k], b[3 + k]);
                        }
                        node_area = area(lo, hi);
                    }
                    if (n <= leaf_max) {
                        // (a box of its own for a single costly leaf -- a cone, a sphere -- was measured: the cone tests of the
                        // groups scene went from 1.10 to 0.80 per ray, its box tests from 4.8 to 5.8, its frame time nowhere;
                        // two nested glass spheres got a box each and their frame 13 % slower.  Not kept.)
                        // A cylinder or cone that stands alone between unbounded records (no run to share a record with)
                        // gets a box of its own when the rule below finds it worth one: its test costs five times the
                        // box's (the CSG synthetic scene's holder, tested by every ray: 17.4 -> 16.6 ms at 7680x4320).
                        // Inside runs the same was measured and changes nothing: groups 3.58 / 3.57 ms, cylinders 0.63 / 0.63.
                        const uint32_t vk = in[lo_i].kind_flags & 0xffu;
                        if (n == 1 && parent_area == INFINITY && (vk == RTAMD_CYLINDER || vk == RTAMD_CONE)) {
                            const DVisit B = bounds_record(std::vector<DVisit>{in[lo_i]});
                            if (cost(in[lo_i]) * outside(B) >= kWorthIt * kBoundsCost) body.push_back(B);
                        }
                        body.insert(body.end(), in.begin() + (long)lo_i, in.begin() + (long)hi_i);
                    } else {
                        // the cut that leaves the least box surface x cost on its two sides
                        std::vector<double> left_cost(n, 0.0), right_cost(n, 0.0);
                        double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY}, c = 0.0;
                        for (size_t q = 0; q + 1 < n; q++) {
                            const double* b = box_of(in[lo_i + q]);
                            for (int k = 0; k < 3; k++) lo[k] = std::min(lo[k], b[k]), hi[k] = std::max(hi[k], b[3 + k]);
                            c += cost(in[lo_i + q]);
                            left_cost[q + 1] = area(lo, hi) * c;
                        }
                        for (int k = 0; k < 3; k++) lo[k] = INFINITY, hi[k] = -INFINITY;
                        c = 0.0;
                        for (size_t q = n; q-- > 1;) {
                            const double* b = box_of(in[lo_i + q]);
                            for (int k = 0; k < 3; k++) lo[k] = std::min(lo[k], b[k]), hi[k] = std::max(hi[k], b[3 + k]);
                            c += cost(in[lo_i + q]);
                            right_cost[q] = area(lo, hi) * c;
                        }
                        size_t best = n / 2;
                        double best_cost = INFINITY;
                        // (cuts near the ends only when the run is short: the depth of the tree stays logarithmic for long runs)
                        const size_t margin = n > 256 ? n / 4 : 1;
                        for (size_t q = margin; q + margin <= n; q++)
                            if (left_cost[q] + right_cost[q] < best_cost) best_cost = left_cost[q] + right_cost[q], best = q;
                        std::vector<DVisit> l = build(lo_i, lo_i + best, node_area), r = build(lo_i + best, hi_i, node_area);
                        body = std::move(l);
                        body.insert(body.end(), r.begin(), r.end());
                    }
                    if (n > 1 && node_area <= shrink * parent_area) {
                        const DVisit B = bounds_record(body);
                        if (total(body) * outside(B) >= kWorthIt * kBoundsCost) body.insert(body.begin(), B);
                    }
                    return body;
                };
                const std::vector<DVisit> tree = build(i, j, INFINITY);
                out.insert(out.end(), tree.begin(), tree.end());
            }
            i = j;
        }
        return out;
    };
    dv_dfs = with_bounds(dv_dfs);
    if (getenv("RTAMD_UPLOAD_DUMP")) {  // diagnostic: the depth-first list, B<n> = bounds record over the next n records
        fprintf(stderr, "[rtamd upload] depth-first list:");
        for (const DVisit& v : dv_dfs) {
            const uint32_t k = v.kind_flags & 0xffu;
            if (k == DVISIT_BOUNDS)
                fprintf(stderr, " B%d", v.unit);
            else
                fprintf(stderr, " %c", k == RTAMD_MESH ? 'm' : k == RTAMD_CUBE ? 'c' : k == RTAMD_SPHERE ? 's' : k == RTAMD_PLANE ? 'p' : 'q');
        }
        fprintf(stderr, "\n");
    }
    // runs of axis-aligned cubes with the same shadow flag: every record carries the length of the rest of its run
    for (size_t i = dv_dfs.size(); i-- > 0;) {
        const uint32_t want = (uint32_t)RTAMD_CUBE | (DLEAF_DIAG << 8), mask = 0xffu | (DLEAF_DIAG << 8);
        DVisit& D = dv_dfs[i];
        if ((D.vflags_class & RTAMD_VISIT_CSG) || (D.kind_flags & mask) != want) continue;
        const bool joins = i + 1 < dv_dfs.size() && dv_dfs[i + 1].run > 0 && (dv_dfs[i + 1].kind_flags & mask) == want &&
                           ((dv_dfs[i + 1].kind_flags ^ D.kind_flags) & (RTAMD_LEAF_SHADOW << 8)) == 0;
        D.run = joins ? dv_dfs[i + 1].run + 1 : 1;
    }
    // runs of planes with the same shadow flag (the walls of a room): the kernels for scenes with CSG have a loop of their own for them
    for (size_t i = dv_dfs.size(); i-- > 0;) {
        DVisit& D = dv_dfs[i];
        // (not the planes below groups: the loop reads the record's matrix alone, and strict arithmetic takes their rays level by level)
        const uint32_t mask = 0xffu | (DLEAF_LEVELS << 8);
        if ((D.vflags_class & RTAMD_VISIT_CSG) || (D.kind_flags & mask) != (uint32_t)RTAMD_PLANE) continue;
        const bool joins = i + 1 < dv_dfs.size() && dv_dfs[i + 1].run > 0 && (dv_dfs[i + 1].kind_flags & mask) == (uint32_t)RTAMD_PLANE &&
                           !(dv_dfs[i + 1].vflags_class & RTAMD_VISIT_CSG) &&
                           ((dv_dfs[i + 1].kind_flags ^ D.kind_flags) & (RTAMD_LEAF_SHADOW << 8)) == 0;
        D.run = joins ? dv_dfs[i + 1].run + 1 : 1;
    }
    // the same in the class-major list, for cubes that are a class of their own (marked before the bounds records go in)
    for (size_t i = 0; i < dv.size(); i++) {
        const uint32_t want = (uint32_t)RTAMD_CUBE | (DLEAF_DIAG << 8), mask = 0xffu | (DLEAF_DIAG << 8);
        DVisit& D = dv[i];
        const bool own_class = (D.vflags_class & RTAMD_VISIT_CLASS_END) && (i == 0 || (dv[i - 1].vflags_class & RTAMD_VISIT_CLASS_END));
        D.run = !(D.vflags_class & RTAMD_VISIT_CSG) && (D.kind_flags & mask) == want && own_class ? 1 : 0;
    }
    dv = with_bounds(dv);
    for (size_t i = dv.size(); i-- > 0;) {
        DVisit& D = dv[i];
        if ((D.kind_flags & 0xffu) == DVISIT_BOUNDS || D.run == 0) continue;
        const bool joins = i + 1 < dv.size() && (dv[i + 1].kind_flags & 0xffu) != DVISIT_BOUNDS && dv[i + 1].run > 0 &&
                           ((dv[i + 1].kind_flags ^ D.kind_flags) & (RTAMD_LEAF_SHADOW << 8)) == 0;
        D.run = joins ? dv[i + 1].run + 1 : 1;
    }
    dv.push_back(DVisit{});  // spare record, as in the depth-first list
    d.n_visits = (int)dv.size() - 1;
    d.n_visits_dfs = (int)dv_dfs.size();
    d.diag_thr = diag_thr;
    d.n_cone_quads = (int32_t)(cone_quads.size() / 8);
    if ((rc = upload(ctx, cone_quads.data(), cone_quads.size(), &d.cone_quads))) return rc;
    dv_dfs.push_back(DVisit{});  // spare record: the loop over a run of cubes reads one record past the run
    if ((rc = upload(ctx, dv_dfs.data(), dv_dfs.size(), &d.visits_dfs))) return rc;
    if ((rc = upload(ctx, class_mat.data(), class_mat.size(), &d.class_material))) return rc;
    if ((rc = upload(ctx, dv.data(), dv.size(), &d.visits))) return rc;
    if ((rc = ensure_frames(ctx, std::max(1, s->max_remaining)))) return rc;
    // ---- where on the screen the costly leaves are (empty list: no preference)
    ctx->costly.clear();
    for (auto& t : ctx->orders) t.key = {{-1, -1, -1, -1, -1, -1, -1, -1}}, t.used = false;
    {
        bool usable = cam_ok;
        for (int i = 0; usable && i < s->n_leaves; i++) {
            const rtamd_leaf& L = s->leaves[i];
            const rtamd_material& M = s->materials[L.material];
            if (M.reflective == 0.0 || M.transparency == 0.0) continue;
            // a shape without bounds makes the whole screen costly, i.e. nothing is preferred
            double r[4];
            if (!has_box[(size_t)i] || !screen_rect(&boxes[(size_t)i * 6], r)) {
                usable = false;
                break;
            }
            auto pix = [](double v, int n) { return (int)std::min<double>(std::max<double>(v, -1.0), (double)n); };
            std::array<int, 4> q{{pix(std::floor(r[0]) - 1, C.hsize), pix(std::ceil(r[1]) + 1, C.hsize), pix(std::floor(r[2]) - 1, C.vsize),
                                  pix(std::ceil(r[3]) + 1, C.vsize)}};
            if (q[1] >= 0 && q[0] < C.hsize && q[3] >= 0 && q[2] < C.vsize) ctx->costly.push_back(q);
        }
        if (!usable) ctx->costly.clear();
    }
    ctx->ds = d;
    ctx->has_csg = s->n_csg_roots > 0;
    bool has_bounds = false;
    ctx->variant = ctx->has_csg ? (csg_small ? 3 : 2) : (s->n_tris > 0 ? 1 : 0);
    ctx->has_levels = false;
    for (const DLeaf& L : dl) ctx->has_levels = ctx->has_levels || (L.flags & DLEAF_LEVELS) != 0;
    if (ctx->has_csg && s->n_tris == 0) ctx->variant |= 32;  // the CSG kernels without the code for triangles
    if (!ctx->has_csg)  // the CSG kernels are built without the slab code
        for (int i = 0; i < s->n_leaves; i++) {
            if (dl[i].kind == RTAMD_CUBE && (dl[i].flags & DLEAF_DIAG)) ctx->variant |= 4;
            if (dl[i].kind == RTAMD_CYLINDER || dl[i].kind == RTAMD_CONE) ctx->variant |= 8;
        }
    if (!ctx->has_csg) {
        for (const DVisit& v : dv) has_bounds |= (v.kind_flags & 0xffu) == DVISIT_BOUNDS;
        for (const DVisit& v : dv_dfs) has_bounds |= (v.kind_flags & 0xffu) == DVISIT_BOUNDS;
        if (has_bounds) ctx->variant |= 16;
    }
    ctx->have_scene = true;
    guard.done = true;
    // counters of the scene before mean nothing for this one (an overflow of its CSG lists least of all)
    HIPCHK(ctx, hipMemset(ctx->d_stats, 0, (size_t)ctx->stat_blocks * RTAMD_STAT_STRIDE * sizeof(unsigned long long)));
    ctx->ovf_seen = 0;
    return RTAMD_OK;
}

// A stride near n / golden ratio that is coprime to n: g -> (g * stride) % n visits every group once,
// neighbours in the queue far apart in the frame.
static int64_t coprime_stride(int64_t n) {
    if (n <= 2) return 1;
    int64_t s = (int64_t)((double)n * 0.6180339887498949);
    if (s < 1) s = 1;
    auto gcd = [](int64_t a, int64_t b) {
        while (b) {
            int64_t t = a % b;
            a = b, b = t;
        }
        return a;
    };
    while (gcd(s, n) != 1) s++;
    return s % n == 0 ? 1 : s;
}

static int64_t part_rows(int vsize, const rtamd_rows* r) {
    if (!r || r->tile_rows <= 0 || r->n_parts <= 0 || r->part < 0 || r->part >= r->n_parts || r->span < 0) return -1;
    const int64_t span = std::max(1, r->span);
    if (r->part + span > r->n_parts) return -1;
    int64_t rows = 0;
    for (int64_t first = r->part; first * r->tile_rows < vsize; first += r->n_parts)
        for (int64_t tile = first; tile < first + span && tile * r->tile_rows < vsize; tile++)
            rows += std::min<int64_t>(r->tile_rows, vsize - tile * r->tile_rows);
    return rows;
}

int64_t rtamd_part_rows(const rtamd_ctx* ctx, const rtamd_rows* rows) {
    if (!ctx || !ctx->have_scene) return -1;
    return part_rows(ctx->ds.cam.vsize, rows);
}

// static_scratch: the work-queue counters and the frame stacks live in this device's copy of the file-scope `scratch`
// array instead of the context's own buffers, and nothing on the way allocates, frees or waits for the device.
// Everything a launch needs short of touching the stream: the grid, the scratch, the order of the work queue (a table
// that is built -- host work, an allocation and a blocking copy -- the first time a partition of the frame is used).
static int prepare(rtamd_ctx* ctx, int mode, DWork& w, bool static_scratch, int* blocks_out) {
    if (mode != RTAMD_MODE_STRICT && mode != RTAMD_MODE_FAST) return fail(ctx, RTAMD_E_INVALID, "unknown mode");
    const int variant = ctx->variant | (ctx->count_work ? 64 : 0);
    int blocks = (int)std::min<int64_t>(ctx->max_blocks[mode][variant], (w.n_pixels + 255) / 256);
    blocks = std::max(blocks, 1);
    w.stats = ctx->d_stats;
    if (static_scratch) {
        // The frame stacks of one block: levels x doubles per frame x 256 lanes.  The grid is cut down to what the
        // array holds; below one block per CU the call is refused instead of crawling.
        const int64_t levels = std::max(1, w.remaining);
        const int64_t per_block = levels * (mode == RTAMD_MODE_STRICT ? RTAMD_FRAME_F64_STRICT : RTAMD_FRAME_F64_FAST) * 256 * (int64_t)sizeof(double);
        const int64_t fit = (int64_t)RTAMD_STATIC_FRAME_BYTES / per_block;
        if (fit < std::min<int64_t>(ctx->num_cus, blocks))
            return fail(ctx, RTAMD_E_LIMIT,
                        "rtamd_render_rows_static: the frame stacks for recursion budget " + std::to_string(w.remaining) + " in " +
                            (mode == RTAMD_MODE_STRICT ? "strict" : "fast") + " arithmetic need " + std::to_string(per_block) +
                            " bytes per block; fewer than one block per CU fit the static scratch of " +
                            std::to_string((long long)RTAMD_STATIC_FRAME_BYTES) + " bytes (use rtamd_render_rows)");
        blocks = (int)std::min<int64_t>(blocks, fit);
        w.queue_head = reinterpret_cast<unsigned long long*>(ctx->static_scratch);
        w.frames = reinterpret_cast<double*>(ctx->static_scratch + RTAMD_STATIC_QUEUE_BYTES);
        w.total_lanes = (int64_t)blocks * 256;
    } else {
        int rc = ensure_frames(ctx, std::max(1, w.remaining));
        if (rc) return rc;
        w.queue_head = ctx->d_queue;
        w.frames = ctx->d_frames;
        w.total_lanes = ctx->total_lanes;
    }
    // Camera rays: blocks of RTAMD_BLOCK_W x RTAMD_BLOCK_H pixels when the row tiles are whole multiples of the block height (no block then straddles
    // two tiles, which lie far apart in the frame); strips of 64 pixels otherwise and for caller-supplied rays.
    w.gcols = 0, w.n_rows = 0;
    if (!w.custom_rays && w.tile_rows % RTAMD_BLOCK_H == 0 && w.hsize >= RTAMD_BLOCK_W) {
        w.n_rows = (int32_t)(w.n_pixels / w.hsize);
        w.gcols = (w.hsize + RTAMD_BLOCK_W - 1) / RTAMD_BLOCK_W;
        w.n_groups = (int64_t)w.gcols * ((w.n_rows + RTAMD_BLOCK_H - 1) / RTAMD_BLOCK_H);
    } else {
        w.n_groups = (w.n_pixels + 63) / 64;
    }
    w.group_stride = coprime_stride(w.n_groups);
    w.order = nullptr;
    // A wave finishes its group of pixels before it asks for the next one: its lanes then go through the phases of
    // a pixel more or less together (fewer phases to execute per iteration, rays of one kind side by side), which
    // is worth more than the lanes that idle towards the end of a group.
    w.refill_min = 64;
    if (const char* e = getenv("RTAMD_REFILL_MIN")) w.refill_min = atoi(e);  // diagnostic
    w.mesh_few = 32;
    if (const char* e = getenv("RTAMD_MESH_FEW")) w.mesh_few = atoi(e);  // diagnostic: 0 = always the loop, 64 = never
    if (getenv("RTAMD_BRUTE_FORCE")) w.mesh_few = 0;
    // Teams of lanes per pixel for the groups the table below marks (FAST kernels): they halve the chains of dependent rays at
    // the price of idle lanes, which pays when the frame (or this rank's part of it) is too small to fill the GPU many times
    // over -- then the longest chain, not the throughput, is what the frame waits for.  Same pixels either way.
    int teams = 1, teams_rule = 0;  // rule: 0 never, 1 when every piece of work gets a wave at once, 2 whenever allowed
    if (mode == RTAMD_MODE_FAST && !static_scratch && !ctx->count_work && ctx->ds.n_lights >= 1 && ctx->ds.n_lights <= 3) {
        const char* e = getenv("RTAMD_TEAMS");  // diagnostic: 0 never, 1 whenever the scene allows
        teams_rule = e ? (atoi(e) != 0 ? 2 : 0) : 1;
        teams = ctx->ds.n_lights == 1 ? 2 : 4;
    }
    const int64_t wave_slots = (int64_t)ctx->max_blocks[2][variant & 63] * 4;  // (of the kernels with teams)
    w.team = 1;
    w.n_queue = w.n_groups;
    // (the table is memory the context allocates and fills with a blocking copy: not for static_scratch, whose
    // pixels are the same bits in another order of work)
    if (!static_scratch && !w.custom_rays && !ctx->costly.empty() && w.n_groups > 1 && w.n_groups < (int64_t)1 << 31) {
        // Groups that touch a costly rectangle go first, the rest after them, each set in the strided order.
        // The table depends on the partition of the frame only: it is kept until that changes.
        const std::array<int64_t, 8> key{{w.hsize, w.vsize, w.tile_rows, w.part, w.n_parts, w.span, teams_rule ? teams : 1, teams_rule == 1 ? wave_slots : 0}};
        rtamd_ctx::OrderTable* found = nullptr;
        for (auto& t : ctx->orders)
            if (t.key == key) found = &t;
        const bool build = found == nullptr;
        if (build) found = &ctx->orders[ctx->order_next], ctx->order_next = (ctx->order_next + 1) % rtamd_ctx::kOrderTables;
        rtamd_ctx::OrderTable& T = *found;
        if (build) {
            T.used = false;
            std::vector<int32_t> first, rest;
            first.reserve((size_t)w.n_groups), rest.reserve((size_t)w.n_groups);
            for (int64_t j = 0; j < w.n_groups; j++) {
                const int64_t grp = (j * w.group_stride) % w.n_groups;
                bool hot = false;
                auto touches = [&](int64_t lrow, int64_t xa, int64_t xb) {  // pixels xa..xb of row lrow of this part
                    const int64_t tile = lrow / w.tile_rows, cyc = tile / w.span;
                    const int64_t py = (cyc * w.n_parts + w.part + (tile - cyc * w.span)) * w.tile_rows + (lrow - tile * w.tile_rows);
                    for (const auto& r : ctx->costly) hot |= py >= r[2] && py <= r[3] && xb >= r[0] && xa <= r[1];
                };
                if (w.gcols) {
                    const int64_t gy = grp / w.gcols, gx = grp - gy * w.gcols;
                    for (int64_t lrow = gy * RTAMD_BLOCK_H; lrow < std::min<int64_t>(gy * RTAMD_BLOCK_H + RTAMD_BLOCK_H, w.n_rows) && !hot; lrow++)
                        touches(lrow, gx * RTAMD_BLOCK_W, std::min<int64_t>(gx * RTAMD_BLOCK_W + RTAMD_BLOCK_W - 1, w.hsize - 1));
                } else {
                    const int64_t p0 = grp * 64, p1 = std::min<int64_t>(p0 + 63, w.n_pixels - 1);
                    for (int64_t lrow = p0 / w.hsize; lrow <= p1 / w.hsize && !hot; lrow++)
                        touches(lrow, std::max<int64_t>(p0 - lrow * w.hsize, 0), std::min<int64_t>(p1 - lrow * w.hsize, w.hsize - 1));
                }
                (hot ? first : rest).push_back((int32_t)grp);
            }
            if (!first.empty() && w.n_groups < ((int64_t)1 << 29)) {
                // the groups a kernel may give a team of lanes per pixel: marked in the sign bit; with teams such a group
                // is handed out in `team` pieces (bits 29-30), one per queue entry, so that its pixels spread over as many waves
                T.hot = first.size();
                if (teams_rule == 0 || (teams_rule == 1 && w.n_groups + (int64_t)first.size() * (teams - 1) > wave_slots)) teams = 1;
                T.teams = teams;
                if (teams > 1) {
                    std::vector<int32_t> pieces;
                    pieces.reserve(first.size() * (size_t)teams);
                    for (int32_t g : first)
                        for (int k = 0; k < teams; k++) pieces.push_back((int32_t)(0x80000000u | ((uint32_t)k << 29) | (uint32_t)g));
                    first.swap(pieces);
                }
                first.insert(first.end(), rest.begin(), rest.end());
                if (first.size() > T.cap) {
                    HIPCHK(ctx, hipDeviceSynchronize());
                    if (T.d) HIPCHK(ctx, hipFree(T.d));
                    T.d = nullptr, T.cap = 0;
                    HIPCHK(ctx, hipMalloc(&T.d, first.size() * sizeof(int32_t)));
                    T.cap = first.size();
                }
                if (getenv("RTAMD_UPLOAD_DUMP")) {  // diagnostic
                    fprintf(stderr, "[rtamd order] groups %lld hot %zu teams %d entries %zu:", (long long)w.n_groups, T.hot, teams, first.size());
                    for (size_t q = 0; q < std::min<size_t>(first.size(), 40); q++) fprintf(stderr, " %08x", (unsigned)first[q]);
                    fprintf(stderr, "\n");
                }
                HIPCHK(ctx, hipDeviceSynchronize());  // an earlier launch may still read the table
                HIPCHK(ctx, hipMemcpy(T.d, first.data(), first.size() * sizeof(int32_t), hipMemcpyHostToDevice));
                T.used = true;
                T.len = first.size();
            }
            T.key = key;
        }
        if (T.used) w.order = T.d, w.n_queue = (int64_t)T.len, w.team = T.teams;
    }
    if (w.team > 1)  // (pieces of groups are units of work like groups: a wave each)
        blocks = (int)std::min<int64_t>(ctx->max_blocks[2][variant & 63], (w.n_queue + 3) / 4);
    *blocks_out = blocks;
    return RTAMD_OK;
}

// The counters of the work queue back to zero (RTAMD_QUEUES of them, RTAMD_QUEUE_STRIDE words apart).
__global__ void rtamd_zero_queue(unsigned long long* head) {
    for (unsigned i = threadIdx.x; i < RTAMD_QUEUES * RTAMD_QUEUE_STRIDE; i += blockDim.x) head[i] = 0;
}

static int launch(rtamd_ctx* ctx, int mode, DWork& w, hipStream_t stream, bool static_scratch = false) {
    int blocks = 0;
    const int prc = prepare(ctx, mode, w, static_scratch, &blocks);
    if (prc != RTAMD_OK) return prc;
    const int variant = ctx->variant | (ctx->count_work ? 64 : 0);
    if (static_scratch) {
        // (a kernel, not hipMemsetAsync: captured into a graph, the memset node of the second graph a process builds was seen
        // not to run when that graph is replayed, and the render kernel then finds the counters where the frame before left them)
        hipLaunchKernelGGL(rtamd_zero_queue, dim3(1), dim3(256), 0, stream, w.queue_head);
        HIPCHK(ctx, hipGetLastError());
    } else {
        HIPCHK(ctx, hipMemsetAsync(w.queue_head, 0, RTAMD_QUEUES * RTAMD_QUEUE_STRIDE * sizeof(unsigned long long), stream));
    }
    // The static entry point records no timing events: its caller may be capturing the stream into a graph, and an event
    // recorded during capture can be neither waited for nor read afterwards (kernel_ms then stays 0).
    const bool time_it = !static_scratch;
    if (time_it && !ctx->skip_ev0) HIPCHK(ctx, hipEventRecord(ctx->ev0, stream));
    hipError_t e = mode == RTAMD_MODE_STRICT ? (ctx->has_levels ? rtamd_launch_levels(ctx->ds, w, variant, blocks, stream)
                                                                : rtamd_launch_strict(ctx->ds, w, variant, blocks, stream))
                   : w.team > 1              ? ((w.n_queue + 3) / 4 <= ctx->max_blocks[3][variant & 63]  // every piece gets a wave at 3 waves per SIMD too
                                                    ? rtamd_launch_teams3(ctx->ds, w, variant & 63, blocks, stream)
                                                    : rtamd_launch_teams(ctx->ds, w, variant & 63, blocks, stream))
                                             : rtamd_launch_fast(ctx->ds, w, variant, blocks, stream);
    if (e != hipSuccess) return fail(ctx, RTAMD_E_HIP, std::string("kernel launch: ") + hipGetErrorString(e));
    if (time_it && !ctx->skip_ev1) HIPCHK(ctx, hipEventRecord(ctx->ev1, stream));
    ctx->timed = time_it;
    return RTAMD_OK;
}

// what: 0 render, 1 only prepare (rtamd_prepare_rows)
static int render_rows(rtamd_ctx* ctx, const rtamd_rows* rows, int mode, double* d_rgb, uint8_t* d_rgb8, void* stream, bool static_scratch, int what = 0) {
    if (!ctx) return RTAMD_E_INVALID;
    if (!ctx->have_scene) return fail(ctx, RTAMD_E_NO_SCENE, "no scene uploaded");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int64_t nrows = part_rows(ctx->ds.cam.vsize, rows);
    if (nrows < 0) return fail(ctx, RTAMD_E_INVALID, "bad rows specification");
    if (nrows == 0) return RTAMD_OK;
    DWork w{};
    w.hsize = ctx->ds.cam.hsize, w.vsize = ctx->ds.cam.vsize;
    w.tile_rows = rows->tile_rows, w.part = rows->part, w.n_parts = rows->n_parts, w.span = std::max(1, rows->span);
    w.remaining = ctx->ds.max_remaining;
    w.n_pixels = nrows * ctx->ds.cam.hsize;
    w.custom_rays = nullptr;
    w.out_rgb = d_rgb, w.out_rgb8 = d_rgb8;
    if (what == 1) {
        int blocks = 0;
        return prepare(ctx, mode, w, false, &blocks);
    }
    return launch(ctx, mode, w, (hipStream_t)stream, static_scratch);
}

int rtamd_prepare_rows(rtamd_ctx* ctx, const rtamd_rows* rows, int mode) {
    return render_rows(ctx, rows, mode, nullptr, nullptr, nullptr, false, 1);
}

int rtamd_render_rows(rtamd_ctx* ctx, const rtamd_rows* rows, int mode, double* d_rgb, uint8_t* d_rgb8, void* stream) {
    return render_rows(ctx, rows, mode, d_rgb, d_rgb8, stream, false);
}

int rtamd_render_rows_static(rtamd_ctx* ctx, const rtamd_rows* rows, int mode, double* d_rgb, uint8_t* d_rgb8, void* stream) {
    return render_rows(ctx, rows, mode, d_rgb, d_rgb8, stream, true);
}

int rtamd_render(rtamd_ctx* ctx, int mode, double* h_rgb, uint8_t* h_rgb8) {
    if (!ctx) return RTAMD_E_INVALID;
    if (!ctx->have_scene) return fail(ctx, RTAMD_E_NO_SCENE, "no scene uploaded");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    size_t npx = (size_t)ctx->ds.cam.hsize * ctx->ds.cam.vsize;
    if (h_rgb && ctx->out_cap < npx * 3 * sizeof(double)) {
        HIPCHK(ctx, hipDeviceSynchronize());
        if (ctx->d_out) HIPCHK(ctx, hipFree(ctx->d_out));
        ctx->d_out = nullptr, ctx->out_cap = 0;
        HIPCHK(ctx, hipMalloc(&ctx->d_out, npx * 3 * sizeof(double)));
        ctx->out_cap = npx * 3 * sizeof(double);
    }
    if (h_rgb8 && ctx->out8_cap < npx * 3) {
        HIPCHK(ctx, hipDeviceSynchronize());
        if (ctx->d_out8) HIPCHK(ctx, hipFree(ctx->d_out8));
        ctx->d_out8 = nullptr, ctx->out8_cap = 0;
        HIPCHK(ctx, hipMalloc(&ctx->d_out8, npx * 3));
        ctx->out8_cap = npx * 3;
    }
    double* d_rgb = h_rgb ? ctx->d_out : nullptr;
    uint8_t* d_u8 = h_rgb8 ? ctx->d_out8 : nullptr;
    int rc = RTAMD_OK;
    const int64_t hsize = ctx->ds.cam.hsize, vsize = ctx->ds.cam.vsize;
    // A large frame takes longer to reach the host than to render (199 MB at 3840x2160: 3.5 ms into page-locked memory,
    // 1.5 ms of kernel).  It is rendered as a few chunks of rows, all launched at once on one stream; each chunk is
    // copied on a second stream as soon as it is complete, while the chunks after it render.
    int64_t chunk_rows = vsize;
    if ((h_rgb || h_rgb8) && npx >= ((size_t)3 << 19) && !getenv("RTAMD_RENDER_WHOLE")) chunk_rows = ((vsize + 3) / 4 + 7) / 8 * 8;
    const int n_chunks = (int)((vsize + chunk_rows - 1) / chunk_rows);
    if (n_chunks <= 1) {
        rtamd_rows all{ctx->ds.cam.vsize, 0, 1, 1};
        rc = rtamd_render_rows(ctx, &all, mode, d_rgb, d_u8, nullptr);
        if (rc == RTAMD_OK) {
            hipError_t e = hipDeviceSynchronize();
            if (e == hipSuccess && h_rgb) e = hipMemcpy(h_rgb, d_rgb, npx * 3 * sizeof(double), hipMemcpyDeviceToHost);
            if (e == hipSuccess && h_rgb8) e = hipMemcpy(h_rgb8, d_u8, npx * 3, hipMemcpyDeviceToHost);
            if (e != hipSuccess) rc = fail(ctx, RTAMD_E_HIP, std::string("render: ") + hipGetErrorString(e));
        }
    } else {
        if (!ctx->s_render) HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->s_render, hipStreamNonBlocking));
        if (!ctx->s_copy) HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->s_copy, hipStreamNonBlocking));
        for (auto& e : ctx->ev_chunk)
            if (!e) HIPCHK(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        HIPCHK(ctx, hipDeviceSynchronize());  // (work of earlier calls on other streams)
        for (int i = 0; i < n_chunks && rc == RTAMD_OK; i++) {
            const size_t first_px = (size_t)i * (size_t)chunk_rows * (size_t)hsize;
            rtamd_rows rows{(int32_t)chunk_rows, i, n_chunks, 1};  // (tiles of chunk_rows rows, one tile per part)
            ctx->skip_ev0 = i != 0, ctx->skip_ev1 = i != n_chunks - 1;  // the timing events span all the chunks
            rc = rtamd_render_rows(ctx, &rows, mode, d_rgb ? d_rgb + first_px * 3 : nullptr, d_u8 ? d_u8 + first_px * 3 : nullptr, ctx->s_render);
            ctx->skip_ev0 = ctx->skip_ev1 = false;
            if (rc == RTAMD_OK) HIPCHK(ctx, hipEventRecord(ctx->ev_chunk[i], ctx->s_render));
        }
        hipError_t e = hipSuccess;
        for (int i = 0; i < n_chunks && rc == RTAMD_OK && e == hipSuccess; i++) {
            const size_t first_px = (size_t)i * (size_t)chunk_rows * (size_t)hsize;
            const size_t n_px = (size_t)std::min<int64_t>(chunk_rows, vsize - (int64_t)i * chunk_rows) * (size_t)hsize;
            e = hipStreamWaitEvent(ctx->s_copy, ctx->ev_chunk[i], 0);
            if (e == hipSuccess && h_rgb) e = hipMemcpyAsync(h_rgb + first_px * 3, d_rgb + first_px * 3, n_px * 3 * sizeof(double), hipMemcpyDeviceToHost, ctx->s_copy);
            if (e == hipSuccess && h_rgb8) e = hipMemcpyAsync(h_rgb8 + first_px * 3, d_u8 + first_px * 3, n_px * 3, hipMemcpyDeviceToHost, ctx->s_copy);
        }
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->s_render);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->s_copy);
        if (rc == RTAMD_OK && e != hipSuccess) rc = fail(ctx, RTAMD_E_HIP, std::string("render: ") + hipGetErrorString(e));
        if (rc != RTAMD_OK) (void)hipDeviceSynchronize();
    }
    if (rc == RTAMD_OK) {
        // the counter is cumulative (rtamd_reset_stats, rtamd_scene_upload clear it): only what has been added
        // since the last look is this frame's -- or that of a rtamd_render_rows call since then
        rtamd_stats st;
        if (ctx->has_csg && rtamd_get_stats(ctx, &st) == RTAMD_OK) {
            if (st.csg_overflows > ctx->ovf_seen)
                rc = fail(ctx, RTAMD_E_OVERFLOW, "a per-lane CSG intersection list overflowed (RTAMD_CSG_LIST_CAP)");
            ctx->ovf_seen = st.csg_overflows;
        }
    }
    return rc;
}

void* rtamd_host_alloc(size_t bytes) {
    void* p = nullptr;
    return hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) == hipSuccess ? p : nullptr;
}

void rtamd_host_free(void* p) {
    if (p) (void)hipHostFree(p);
}

int rtamd_color_at(rtamd_ctx* ctx, int mode, const double* h_rays, int64_t n, int remaining, double* h_rgb) {
    if (!ctx || !h_rays || !h_rgb || n < 0) return RTAMD_E_INVALID;
    if (!ctx->have_scene) return fail(ctx, RTAMD_E_NO_SCENE, "no scene uploaded");
    if (remaining < 0 || remaining > RTAMD_MAX_REMAINING) return fail(ctx, RTAMD_E_LIMIT, "remaining out of range");
    if (n == 0) return RTAMD_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    double *d_rays = nullptr, *d_rgb = nullptr;
    hipError_t e0 = hipMalloc(&d_rays, (size_t)n * 6 * sizeof(double));
    if (e0 == hipSuccess) e0 = hipMalloc(&d_rgb, (size_t)n * 3 * sizeof(double));
    if (e0 == hipSuccess) e0 = hipMemcpy(d_rays, h_rays, (size_t)n * 6 * sizeof(double), hipMemcpyHostToDevice);
    if (e0 != hipSuccess) {
        if (d_rays) (void)hipFree(d_rays);
        if (d_rgb) (void)hipFree(d_rgb);
        return fail(ctx, RTAMD_E_HIP, std::string("color_at: ") + hipGetErrorString(e0));
    }
    DWork w{};
    w.hsize = ctx->ds.cam.hsize, w.vsize = ctx->ds.cam.vsize;
    w.tile_rows = 1, w.part = 0, w.n_parts = 1, w.span = 1;
    w.remaining = remaining;
    w.n_pixels = n;
    w.custom_rays = d_rays;
    w.out_rgb = d_rgb, w.out_rgb8 = nullptr;
    int rc = launch(ctx, mode, w, nullptr);
    if (rc == RTAMD_OK) {
        hipError_t e = hipDeviceSynchronize();
        if (e == hipSuccess) e = hipMemcpy(h_rgb, d_rgb, (size_t)n * 3 * sizeof(double), hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = fail(ctx, RTAMD_E_HIP, std::string("color_at: ") + hipGetErrorString(e));
    }
    (void)hipFree(d_rays);
    (void)hipFree(d_rgb);
    return rc;
}

int rtamd_get_stats(rtamd_ctx* ctx, rtamd_stats* out) {
    if (!ctx || !out) return RTAMD_E_INVALID;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    std::memset(out, 0, sizeof *out);
    if (ctx->timed) {
        HIPCHK(ctx, hipEventSynchronize(ctx->ev1));
        float ms = 0;
        HIPCHK(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
        out->kernel_ms = ms;
    }
    std::vector<unsigned long long> slots((size_t)ctx->stat_blocks * RTAMD_STAT_STRIDE);
    HIPCHK(ctx, hipMemcpy(slots.data(), ctx->d_stats, slots.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    unsigned long long h[ST_COUNT] = {};
    for (int b = 0; b < ctx->stat_blocks; b++)
        for (int k = 0; k < ST_COUNT; k++) h[k] += slots[(size_t)b * RTAMD_STAT_STRIDE + k];
    out->primary_rays = h[ST_PRIMARY], out->shadow_rays = h[ST_SHADOW_RAYS], out->reflect_rays = h[ST_REFLECT];
    out->refract_rays = h[ST_REFRACT], out->container_passes = h[ST_CONTAINER];
    out->rays = h[ST_PRIMARY] + h[ST_SHADOW_RAYS] + h[ST_REFLECT] + h[ST_REFRACT] + h[ST_CONTAINER];
    out->shade_points = h[ST_SHADE_POINTS], out->light_samples = h[ST_LIGHT_SAMPLES], out->csg_overflows = h[ST_CSG_OVERFLOW];
    out->tests_sphere = h[ST_T_SPHERE], out->tests_plane = h[ST_T_PLANE], out->tests_cube = h[ST_T_CUBE];
    out->tests_slab2 = h[ST_T_SLAB2], out->tests_slab3 = h[ST_T_SLAB3], out->tests_cylinder = h[ST_T_CYLINDER];
    out->tests_cone = h[ST_T_CONE], out->tests_caps = h[ST_T_CAPS], out->tests_tri_first = h[ST_T_TRI1];
    out->tests_tri_second = h[ST_T_TRI2], out->tests_box = h[ST_T_BOX], out->tests_run_transform = h[ST_T_RUNX];
    out->team_shade_points = h[ST_TEAM_POINTS];
    if (getenv("RTAMD_PROFILE_DUMP")) {  // diagnostic builds (-DRTAMD_PROFILE) only
        const unsigned long long* p = slots.data() + ST_COUNT;
        if (p[3])
            fprintf(stderr, "[rtamd profile] wave-iterations %llu: refill %.0f, traversal %.0f (record loads %.0f), post %.0f cycles per iteration\n",
                    p[3], (double)p[0] / p[3], (double)p[1] / p[3], (double)p[4] / p[3], (double)p[2] / p[3]);
        const unsigned long long* q = slots.data() + RTAMD_STAT_STRIDE + ST_COUNT;
        if (p[3] && ctx->stat_blocks > 1)
            fprintf(stderr, "[rtamd profile]   post: hit/containers %.0f, setup %.0f, lighting %.0f, return %.0f\n", (double)q[0] / p[3],
                    (double)q[1] / p[3], (double)q[2] / p[3], (double)q[3] / p[3]);
        if (p[3] && ctx->stat_blocks > 1 && q[4])
            fprintf(stderr, "[rtamd profile]   traversals with a container pass pending: %llu (%.1f %%), %.0f cycles each; the others %.0f\n", q[4],
                    100.0 * q[4] / p[3], (double)q[5] / q[4], (double)(p[1] - q[5]) / (double)(p[3] - q[4]));
        if (p[3] && ctx->stat_blocks > 2) {
            // how full the phases after the traversal run: a wave executes a phase when any of its lanes needs it
            const unsigned long long* n = slots.data() + 2 * RTAMD_STAT_STRIDE + ST_COUNT;
            const char* names[4] = {"hit", "setup", "lighting", "return"};
            fprintf(stderr, "[rtamd profile]   phases after the traversal:");
            for (int k = 0; k < 4; k++)
                fprintf(stderr, " %s in %.1f %% of the iterations for %.1f lanes each;", names[k], 100.0 * n[2 * k] / p[3], n[2 * k] ? (double)n[2 * k + 1] / n[2 * k] : 0.0);
            fprintf(stderr, "\n");
        }
        if (p[3] && ctx->stat_blocks > 1)
            fprintf(stderr, "[rtamd profile]   most iterations in one wave %llu, longest-running wave %llu cycles\n", q[6], q[7]);
    }
    return RTAMD_OK;
}

int rtamd_count_work(rtamd_ctx* ctx, int on) {
    if (!ctx) return RTAMD_E_INVALID;
    ctx->count_work = on != 0;
    return RTAMD_OK;
}

int rtamd_reset_stats(rtamd_ctx* ctx) {
    if (!ctx) return RTAMD_E_INVALID;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipDeviceSynchronize());
    HIPCHK(ctx, hipMemset(ctx->d_stats, 0, (size_t)ctx->stat_blocks * RTAMD_STAT_STRIDE * sizeof(unsigned long long)));
    ctx->ovf_seen = 0;
    return RTAMD_OK;
}

