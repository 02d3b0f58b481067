// Several GPUs, one frame, below any host language: the part of include/rtamd.h that starts at rtamd_rows_layout.
// One context per entry of the device list (an entry may repeat: two contexts on one GPU take the same path, which is
// how a one-GPU machine tests it), every context renders its interleaved row tiles into a buffer on its own GPU and
// pushes them -- one strided copy per rank, over xGMI when the GPUs differ -- into their rows of the frame on the
// first device.  Everything is queued from the calling thread on one stream per context; nothing here needs a
// collective library: the exchange is N - 1 point-to-point writes to one GPU, each over its own link.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/rtamd.h"

struct rtamd_multi {
    struct Rank {
        int device = 0;
        rtamd_ctx* ctx = nullptr;
        hipStream_t stream = nullptr;
        hipEvent_t ev[3] = {nullptr, nullptr, nullptr};  // start, rendered, in place
        double* part = nullptr;                          // this rank's rows, compact
        uint8_t* part8 = nullptr;
        size_t part_cap = 0, part8_cap = 0;
        float render_ms = 0, place_ms = 0;
        uint64_t ovf_seen = 0;  // CSG list overflows of this rank counted up to the last rtamd_multi_render
    };
    std::vector<Rank> ranks;
    std::string err;
    int hsize = 0, vsize = 0;
    bool have_scene = false, has_csg = false;
    double* frame = nullptr;  // on ranks[0].device
    uint8_t* frame8 = nullptr;
    size_t frame_cap = 0, frame8_cap = 0;
};

static std::string g_multi_err;

static int mfail(rtamd_multi* m, int code, const std::string& msg) {
    m->err = msg;
    return code;
}

#define MHIP(m, call)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess) return mfail(m, RTAMD_E_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)

int rtamd_rows_layout(int vsize, const rtamd_rows* r, rtamd_layout* out) {
    if (!r || !out || vsize < 0 || r->tile_rows <= 0 || r->n_parts <= 0 || r->part < 0 || r->span < 0) return RTAMD_E_INVALID;
    const int64_t span = std::max(1, r->span);
    if (r->part + span > r->n_parts) return RTAMD_E_INVALID;
    out->first_row = (int64_t)r->part * r->tile_rows;
    out->chunk_rows = span * r->tile_rows;
    out->stride_rows = (int64_t)r->n_parts * r->tile_rows;
    out->n_chunks = 0, out->tail_rows = 0;
    for (int64_t start = out->first_row; start < vsize; start += out->stride_rows) {
        const int64_t have = std::min<int64_t>(out->chunk_rows, vsize - start);
        if (have == out->chunk_rows)
            out->n_chunks++;
        else
            out->tail_rows = have;  // only the last cycle can be cut short
    }
    return RTAMD_OK;
}

const char* rtamd_multi_last_error(const rtamd_multi* m) { return m ? m->err.c_str() : g_multi_err.c_str(); }

void rtamd_multi_destroy(rtamd_multi* m) {
    if (!m) return;
    for (auto& r : m->ranks) {
        if (!r.ctx) continue;  // (a rank whose creation failed: its device index may not even exist)
        (void)hipSetDevice(r.device);
        if (r.stream) (void)hipStreamSynchronize(r.stream);
    }
    if (!m->ranks.empty() && m->ranks[0].ctx) {
        (void)hipSetDevice(m->ranks[0].device);
        if (m->frame) (void)hipFree(m->frame);
        if (m->frame8) (void)hipFree(m->frame8);
    }
    for (auto& r : m->ranks) {
        if (!r.ctx) continue;
        (void)hipSetDevice(r.device);
        if (r.part) (void)hipFree(r.part);
        if (r.part8) (void)hipFree(r.part8);
        for (hipEvent_t e : r.ev)
            if (e) (void)hipEventDestroy(e);
        if (r.stream) (void)hipStreamDestroy(r.stream);
        if (r.ctx) rtamd_ctx_destroy(r.ctx);
    }
    delete m;
}

int rtamd_multi_create(const int* devices, int n_devices, rtamd_multi** out) {
    if (!out) {
        g_multi_err = "rtamd_multi_create: out is NULL";
        return RTAMD_E_INVALID;
    }
    *out = nullptr;
    if (!devices || n_devices < 1 || n_devices > 64) {
        g_multi_err = "rtamd_multi_create: need 1 to 64 devices";
        return RTAMD_E_INVALID;
    }
    rtamd_multi* m = new (std::nothrow) rtamd_multi();
    if (!m) {
        g_multi_err = "out of host memory";
        return RTAMD_E_LIMIT;
    }
    try {
        m->ranks.resize((size_t)n_devices);
    } catch (...) {
        delete m;
        g_multi_err = "out of host memory";
        return RTAMD_E_LIMIT;
    }
    auto bail = [&](int code, const std::string& msg) {
        g_multi_err = msg;
        rtamd_multi_destroy(m);
        return code;
    };
    for (int i = 0; i < n_devices; i++) {
        auto& r = m->ranks[(size_t)i];
        r.device = devices[i];
        int rc = rtamd_ctx_create(devices[i], &r.ctx);
        if (rc != RTAMD_OK) return bail(rc, std::string("device ") + std::to_string(devices[i]) + ": " + rtamd_last_error(nullptr));
        hipError_t e = hipSetDevice(r.device);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&r.stream, hipStreamNonBlocking);
        for (int k = 0; k < 3 && e == hipSuccess; k++) e = hipEventCreate(&r.ev[k]);
        if (e != hipSuccess) return bail(RTAMD_E_HIP, std::string("stream / events: ") + hipGetErrorString(e));
        // the rows of every other GPU are written straight into the frame on the first one
        if (r.device != m->ranks[0].device) {
            int can = 0;
            e = hipDeviceCanAccessPeer(&can, r.device, m->ranks[0].device);
            if (e != hipSuccess || !can)
                return bail(RTAMD_E_HIP, "device " + std::to_string(r.device) + " cannot write to device " + std::to_string(m->ranks[0].device) + " (no peer access)");
            e = hipDeviceEnablePeerAccess(m->ranks[0].device, 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) return bail(RTAMD_E_HIP, std::string("hipDeviceEnablePeerAccess: ") + hipGetErrorString(e));
            (void)hipGetLastError();
        }
    }
    *out = m;
    return RTAMD_OK;
}

int rtamd_multi_upload(rtamd_multi* m, const rtamd_scene* scene) {
    if (!m || !scene) return RTAMD_E_INVALID;
    for (auto& r : m->ranks) {
        int rc = rtamd_scene_upload(r.ctx, scene);
        if (rc != RTAMD_OK) {
            m->have_scene = false;  // the ranks may now hold different scenes
            return mfail(m, rc, std::string("device ") + std::to_string(r.device) + ": " + rtamd_last_error(r.ctx));
        }
    }
    m->hsize = scene->camera.hsize, m->vsize = scene->camera.vsize;
    m->have_scene = true;
    m->has_csg = scene->n_csg_roots > 0;
    for (auto& r : m->ranks) r.ovf_seen = 0;  // (rtamd_scene_upload cleared the counters)
    return RTAMD_OK;
}

// Waits for everything queued on every rank's stream (a failed call may have left work in flight on some of them).
static void drain(rtamd_multi* m) {
    for (auto& r : m->ranks) {
        if (!r.stream) continue;
        (void)hipSetDevice(r.device);
        (void)hipStreamSynchronize(r.stream);
    }
}

template <class T>
static int ensure(rtamd_multi* m, int device, T** p, size_t* cap, size_t bytes) {
    if (bytes <= *cap) return RTAMD_OK;
    drain(m);  // the frame on the first device is written from the other devices' streams
    MHIP(m, hipSetDevice(device));
    MHIP(m, hipDeviceSynchronize());
    if (*p) MHIP(m, hipFree(*p));
    *p = nullptr, *cap = 0;
    MHIP(m, hipMalloc((void**)p, bytes));
    *cap = bytes;
    return RTAMD_OK;
}

static int multi_render(rtamd_multi* m, int mode, int tile_rows, int first_parts, int other_parts, double* h_rgb, uint8_t* h_rgb8, int want);

int rtamd_multi_render(rtamd_multi* m, int mode, int tile_rows, int first_parts, int other_parts, double* h_rgb, uint8_t* h_rgb8, int want) {
    if (!m) return RTAMD_E_INVALID;
    const int rc = multi_render(m, mode, tile_rows, first_parts, other_parts, h_rgb, h_rgb8, want);
    // A failure part of the way leaves kernels and copies of the ranks queued before it in flight, some of them writing
    // into the frame on the first device: nothing may outlive the call that the next call could free under it.
    if (rc != RTAMD_OK) drain(m);
    return rc;
}

static int multi_render(rtamd_multi* m, int mode, int tile_rows, int first_parts, int other_parts, double* h_rgb, uint8_t* h_rgb8, int want) {
    if (!m->have_scene) return mfail(m, RTAMD_E_NO_SCENE, "no scene uploaded");
    if (tile_rows < 1 || first_parts < 1 || other_parts < 1) return mfail(m, RTAMD_E_INVALID, "tile_rows and the shares must be positive");
    const bool want64 = h_rgb != nullptr || (want & RTAMD_WANT_F64), want8 = h_rgb8 != nullptr || (want & RTAMD_WANT_U8);
    if (!want64 && !want8) return mfail(m, RTAMD_E_INVALID, "neither the f64 nor the 8-bit frame was asked for");
    const int n = (int)m->ranks.size();
    const int n_parts = first_parts + (n - 1) * other_parts;
    const size_t px_row = (size_t)m->hsize, npx = px_row * (size_t)m->vsize;
    int rc;
    if (want64 && (rc = ensure(m, m->ranks[0].device, &m->frame, &m->frame_cap, npx * 3 * sizeof(double)))) return rc;
    if (want8 && (rc = ensure(m, m->ranks[0].device, &m->frame8, &m->frame8_cap, npx * 3))) return rc;
    std::vector<rtamd_rows> rows((size_t)n);
    std::vector<int64_t> nrows((size_t)n);
    for (int i = 0; i < n; i++) {
        auto& r = m->ranks[(size_t)i];
        rows[(size_t)i] = rtamd_rows{tile_rows, i == 0 ? 0 : first_parts + (i - 1) * other_parts, n_parts, i == 0 ? first_parts : other_parts};
        nrows[(size_t)i] = rtamd_part_rows(r.ctx, &rows[(size_t)i]);
        if (nrows[(size_t)i] < 0) return mfail(m, RTAMD_E_INVALID, "bad rows specification");
        if (want64 && (rc = ensure(m, r.device, &r.part, &r.part_cap, (size_t)nrows[(size_t)i] * px_row * 3 * sizeof(double)))) return rc;
        if (want8 && (rc = ensure(m, r.device, &r.part8, &r.part8_cap, (size_t)nrows[(size_t)i] * px_row * 3))) return rc;
    }
    // What a rank's first render of a partition does on the host and with blocking copies (the order of its work queue)
    // is done for every rank before any kernel is queued: the ranks' kernels then start together, in the first frame too.
    for (int i = 0; i < n; i++) {
        if (nrows[(size_t)i] == 0) continue;
        rc = rtamd_prepare_rows(m->ranks[(size_t)i].ctx, &rows[(size_t)i], mode);
        if (rc != RTAMD_OK) return mfail(m, rc, std::string("device ") + std::to_string(m->ranks[(size_t)i].device) + ": " + rtamd_last_error(m->ranks[(size_t)i].ctx));
    }
    // queue everything: render, then this rank's rows into their places of the frame, on the rank's own stream
    for (int i = 0; i < n; i++) {
        auto& r = m->ranks[(size_t)i];
        if (nrows[(size_t)i] == 0) continue;
        MHIP(m, hipSetDevice(r.device));
        MHIP(m, hipEventRecord(r.ev[0], r.stream));
        rc = rtamd_render_rows(r.ctx, &rows[(size_t)i], mode, want64 ? r.part : nullptr, want8 ? r.part8 : nullptr, r.stream);
        if (rc != RTAMD_OK) return mfail(m, rc, std::string("device ") + std::to_string(r.device) + ": " + rtamd_last_error(r.ctx));
        MHIP(m, hipEventRecord(r.ev[1], r.stream));
        rtamd_layout L;
        (void)rtamd_rows_layout(m->vsize, &rows[(size_t)i], &L);
        auto place = [&](char* frame, const char* part, size_t row_bytes) -> hipError_t {
            hipError_t e = hipSuccess;
            if (L.n_chunks > 0)
                e = hipMemcpy2DAsync(frame + (size_t)L.first_row * row_bytes, (size_t)L.stride_rows * row_bytes, part, (size_t)L.chunk_rows * row_bytes,
                                     (size_t)L.chunk_rows * row_bytes, (size_t)L.n_chunks, hipMemcpyDeviceToDevice, r.stream);
            if (e == hipSuccess && L.tail_rows > 0)
                e = hipMemcpyAsync(frame + (size_t)(L.first_row + L.n_chunks * L.stride_rows) * row_bytes, part + (size_t)(L.n_chunks * L.chunk_rows) * row_bytes,
                                   (size_t)L.tail_rows * row_bytes, hipMemcpyDeviceToDevice, r.stream);
            return e;
        };
        if (want64) MHIP(m, place(reinterpret_cast<char*>(m->frame), reinterpret_cast<const char*>(r.part), px_row * 3 * sizeof(double)));
        if (want8) MHIP(m, place(reinterpret_cast<char*>(m->frame8), reinterpret_cast<const char*>(r.part8), px_row * 3));
        MHIP(m, hipEventRecord(r.ev[2], r.stream));
    }
    for (int i = 0; i < n; i++) {
        auto& r = m->ranks[(size_t)i];
        r.render_ms = r.place_ms = 0;
        if (nrows[(size_t)i] == 0) continue;
        MHIP(m, hipSetDevice(r.device));
        MHIP(m, hipStreamSynchronize(r.stream));
        MHIP(m, hipEventElapsedTime(&r.render_ms, r.ev[0], r.ev[1]));
        MHIP(m, hipEventElapsedTime(&r.place_ms, r.ev[1], r.ev[2]));
    }
    // Same pixels as rtamd_render, and the same refusal: a per-lane CSG intersection list that overflowed on any rank
    // left pixels that mean nothing (the counters are cumulative: what was added since the last look is this frame's).
    if (m->has_csg) {
        bool overflowed = false;
        for (auto& r : m->ranks) {
            rtamd_stats st;
            rc = rtamd_get_stats(r.ctx, &st);
            if (rc != RTAMD_OK) return mfail(m, rc, std::string("device ") + std::to_string(r.device) + ": " + rtamd_last_error(r.ctx));
            overflowed = overflowed || st.csg_overflows > r.ovf_seen;
            r.ovf_seen = st.csg_overflows;
        }
        if (overflowed) return mfail(m, RTAMD_E_OVERFLOW, "a per-lane CSG intersection list overflowed (RTAMD_CSG_LIST_CAP)");
    }
    MHIP(m, hipSetDevice(m->ranks[0].device));
    if (h_rgb) MHIP(m, hipMemcpy(h_rgb, m->frame, npx * 3 * sizeof(double), hipMemcpyDeviceToHost));
    if (h_rgb8) MHIP(m, hipMemcpy(h_rgb8, m->frame8, npx * 3, hipMemcpyDeviceToHost));
    return RTAMD_OK;
}

int rtamd_multi_frame(rtamd_multi* m, double** d_rgb, uint8_t** d_rgb8) {
    if (!m) return RTAMD_E_INVALID;
    if (d_rgb) *d_rgb = m->frame;
    if (d_rgb8) *d_rgb8 = m->frame8;
    return RTAMD_OK;
}

int rtamd_multi_get_stats(rtamd_multi* m, rtamd_stats* total, double* render_ms, double* place_ms) {
    if (!m) return RTAMD_E_INVALID;
    rtamd_stats sum;
    std::memset(&sum, 0, sizeof sum);
    const size_t n_counters = offsetof(rtamd_stats, kernel_ms) / sizeof(uint64_t);
    const size_t n_tests = (sizeof(rtamd_stats) - offsetof(rtamd_stats, tests_sphere)) / sizeof(uint64_t);
    for (size_t i = 0; i < m->ranks.size(); i++) {
        auto& r = m->ranks[i];
        rtamd_stats st;
        int rc = rtamd_get_stats(r.ctx, &st);
        if (rc != RTAMD_OK) return mfail(m, rc, rtamd_last_error(r.ctx));
        const uint64_t* a = reinterpret_cast<const uint64_t*>(&st);
        uint64_t* b = reinterpret_cast<uint64_t*>(&sum);
        for (size_t k = 0; k < n_counters; k++) b[k] += a[k];
        for (size_t k = 0; k < n_tests; k++) (&sum.tests_sphere)[k] += (&st.tests_sphere)[k];
        sum.kernel_ms = std::max(sum.kernel_ms, st.kernel_ms);  // the slowest rank's kernel
        if (render_ms) render_ms[i] = r.render_ms;
        if (place_ms) place_ms[i] = r.place_ms;
    }
    if (total) *total = sum;
    return RTAMD_OK;
}

int rtamd_multi_reset_stats(rtamd_multi* m) {
    if (!m) return RTAMD_E_INVALID;
    for (auto& r : m->ranks) {
        int rc = rtamd_reset_stats(r.ctx);
        if (rc != RTAMD_OK) return mfail(m, rc, rtamd_last_error(r.ctx));
        r.ovf_seen = 0;
    }
    return RTAMD_OK;
}
