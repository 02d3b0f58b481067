// C entry points of the host library, for callers without a C++ toolchain (ctypes in the tests
// and the bench). Everything here is a thin wrapper over the C++ API of rt_scene.hpp.
#include <cstring>

#include "../../include/rtamd_host.h"
#include "rt_scene.hpp"

namespace rt {
std::unique_ptr<Scene> builtin_reflect_refract(const std::string& teapot_obj);
std::unique_ptr<Scene> builtin_teapot(const std::string& teapot_obj);
std::unique_ptr<Scene> builtin_csg();
std::unique_ptr<Scene> builtin_default_world();
}  // namespace rt

static thread_local std::string g_err;
void rth_set_error(const std::string& msg) { g_err = msg; }  // (rt_builder.cpp reports through the same string)

template <class F>
static int guard(F&& f) {
    try {
        f();
        return 0;
    } catch (std::exception& e) {
        g_err = e.what();
        return -1;
    }
}

extern "C" {

const char* rth_last_error() { return g_err.c_str(); }

void* rth_load_yaml(const char* path) {
    rt::Scene* s = nullptr;
    guard([&] { s = rt::load_yaml_scene(path).release(); });
    return s;
}
void* rth_load_yaml_str(const char* text) {
    rt::Scene* s = nullptr;
    guard([&] { s = rt::load_yaml_scene_str(text).release(); });
    return s;
}
void* rth_load_sg(const char* path) {
    rt::Scene* s = nullptr;
    guard([&] { s = rt::load_sg(path).release(); });
    return s;
}
// name: "reflect_refract" | "teapot_room" | "teapot" | "csg" | "default_world"; arg: OBJ path where needed
void* rth_builtin_scene(const char* name, const char* arg) {
    rt::Scene* s = nullptr;
    guard([&] {
        std::string n = name, a = arg ? arg : "";
        if (n == "reflect_refract")
            s = rt::builtin_reflect_refract("").release();
        else if (n == "teapot_room")
            s = rt::builtin_reflect_refract(a).release();
        else if (n == "teapot")
            s = rt::builtin_teapot(a).release();
        else if (n == "csg")
            s = rt::builtin_csg().release();
        else if (n == "default_world")
            s = rt::builtin_default_world().release();
        else
            throw rt::Error("unknown builtin scene " + n);
    });
    return s;
}
int rth_save_sg(void* scene, const char* path) {
    return guard([&] { rt::save_sg(*(rt::Scene*)scene, path); });
}
void rth_free_scene(void* scene) { delete (rt::Scene*)scene; }

int rth_scene_info(void* scene, int* hsize, int* vsize, int* n_objects, int* n_lights) {
    auto* s = (rt::Scene*)scene;
    if (!s || !s->camera) {
        g_err = "scene has no camera";
        return -1;
    }
    *hsize = (int)s->camera->hsize, *vsize = (int)s->camera->vsize;
    *n_objects = (int)s->world.objects.size(), *n_lights = (int)s->world.lights.size();
    return 0;
}
// New Camera::new(hsize, vsize, same fov) with the same transform.
int rth_set_resolution(void* scene, int hsize, int vsize) {
    return guard([&] {
        auto* s = (rt::Scene*)scene;
        if (!s->camera) throw rt::Error("scene has no camera");
        if (hsize <= 0 || vsize <= 0) throw rt::Error("resolution must be positive");
        auto c = std::make_unique<rt::Camera>((size_t)hsize, (size_t)vsize, s->camera->field_of_view);
        c->transform = s->camera->transform;
        s->camera = std::move(c);
    });
}

void* rth_flatten(void* scene, int max_remaining) {
    rt::FlatScene* f = nullptr;
    guard([&] {
        auto* s = (rt::Scene*)scene;
        if (!s->camera) throw rt::Error("scene has no camera");
        f = rt::flatten(s->world, *s->camera, max_remaining).release();
    });
    return f;
}
const rtamd_scene* rth_flat_view(void* flat) { return &((rt::FlatScene*)flat)->view; }
void rth_free_flat(void* flat) { delete (rt::FlatScene*)flat; }

// Camera::render into caller memory: rgb = hsize*vsize*3 doubles, rgb8 the 8-bit channels (either may be NULL, not
// both: with rgb NULL only 3 bytes per pixel leave the device). stats (optional): rtamd_stats of the render.
int rth_render(void* scene, int mode, double* rgb, uint8_t* rgb8, rtamd_stats* stats) {
    return guard([&] {
        auto* s = (rt::Scene*)scene;
        if (!s || !s->camera) throw rt::Error("scene has no camera");
        if (!rgb && !rgb8) throw rt::Error("render: neither the f64 pixels nor the 8-bit channels were asked for");
        rt::RenderStats rs;
        rt::Canvas img = s->camera->render(s->world, mode, &rs, (rgb ? RTAMD_WANT_F64 : 0) | (rgb8 ? RTAMD_WANT_U8 : 0));
        if (rgb) std::memcpy(rgb, (const void*)img.pixels().data(), img.pixels().size() * sizeof(rt::Color));
        if (rgb8) std::memcpy(rgb8, img.rgb8().data(), img.rgb8().size());
        if (stats) *stats = rs.gpu;
    });
}
int rth_save_png8(const uint8_t* rgb8, int w, int h, const char* path) {
    return guard([&] {
        if (!rgb8 || w <= 0 || h <= 0 || !path) throw rt::Error("save_png8: bad argument");
        rt::Canvas::write_png8(path, rgb8, (size_t)w, (size_t)h);
    });
}

// Canvas helpers on caller pixels (rgb = w*h*3 doubles)
int rth_canvas_save(const double* rgb, int w, int h, const char* path, int as_ppm) {
    return guard([&] {
        rt::Canvas c((size_t)w, (size_t)h);
        std::memcpy((void*)c.pixels().data(), rgb, (size_t)w * h * sizeof(rt::Color));
        if (as_ppm)
            c.save_ppm(path);
        else
            c.save_png(path);
    });
}
int rth_canvas_quantize(const double* rgb, int64_t n_pixels, uint8_t* out) {
    for (int64_t i = 0; i < n_pixels; i++) rt::Canvas::quantize({rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2]}, out + 3 * i);
    return 0;
}
// returns a malloc'ed PPM string (free with rth_free_str)
char* rth_canvas_ppm(const double* rgb, int w, int h) {
    char* out = nullptr;
    guard([&] {
        rt::Canvas c((size_t)w, (size_t)h);
        std::memcpy((void*)c.pixels().data(), rgb, (size_t)w * h * sizeof(rt::Color));
        std::string s = c.to_ppm();
        out = (char*)std::malloc(s.size() + 1);
        std::memcpy(out, s.c_str(), s.size() + 1);
    });
    return out;
}
void rth_free_str(char* s) { std::free(s); }

// OBJ parser facts for tests: triangles per group etc.
int rth_obj_info(const char* text, int* n_vertices, int* n_normals, int* n_default_tris, int* n_groups, int* n_group_tris,
                 int* lines_ignored, int* n_smooth) {
    return guard([&] {
        rt::ObjParser p = rt::ObjParser::from_str(text);
        *n_vertices = (int)p.vertices.size() - 1, *n_normals = (int)p.normals.size() - 1;
        *n_default_tris = (int)p.default_group.children.size();
        *n_groups = (int)p.groups.size();
        *lines_ignored = (int)p.lines_ignored;
        int gt = 0, sm = 0;
        auto count = [&](rt::Group& g) {
            for (auto& c : g.children)
                if (static_cast<rt::Triangle&>(*c).is_smooth) sm++;
        };
        count(p.default_group);
        for (auto& kv : p.groups) gt += (int)kv.second.children.size(), count(kv.second);
        *n_group_tris = gt, *n_smooth = sm;
    });
}
// Triangle k of the default group: 9 doubles p1,p2,p3 then 12 doubles n1,n2,n3 (xyzw); returns is_smooth or -1
int rth_obj_triangle(const char* text, int k, double* out21) {
    int smooth = -1;
    guard([&] {
        rt::ObjParser p = rt::ObjParser::from_str(text);
        if (k < 0 || k >= (int)p.default_group.children.size()) throw rt::Error("triangle index out of range");
        auto& t = static_cast<rt::Triangle&>(*p.default_group.children[k]);
        const rt::Tuple ps[3] = {t.p1, t.p2, t.p3}, ns[3] = {t.n1, t.n2, t.n3};
        for (int i = 0; i < 3; i++) out21[3 * i] = ps[i].x, out21[3 * i + 1] = ps[i].y, out21[3 * i + 2] = ps[i].z;
        for (int i = 0; i < 3; i++)
            out21[9 + 4 * i] = ns[i].x, out21[10 + 4 * i] = ns[i].y, out21[11 + 4 * i] = ns[i].z, out21[12 + 4 * i] = ns[i].w;
        smooth = t.is_smooth ? 1 : 0;
    });
    return smooth;
}

}  // extern "C"
