// Scene builder of the host library's C ABI (include/rtamd_host.h, rth_world_* / rth_add_* / rth_*_begin / rth_*_end):
// a host in any language walks its own object tree -- the reference's World {objects: Vec<Box<dyn Shape>>, lights}
// (src/scene.rs:6-9), shapes with Props (src/shape.rs:5-12), Group {children} (src/shapes/group.rs:8-11), Csg {op, left,
// right} (src/shapes/csg.rs:5-15) -- with one call per node, and gets the same rt::Scene the loaders of this library
// build; rth_flatten then does for it what it does for them.
#include <cmath>
#include <cstring>
#include <new>

#include "../../include/rtamd_host.h"
#include "rt_scene.hpp"

void rth_set_error(const std::string& msg);  // rt_capi.cpp: the thread's rth_last_error

struct rth_world {
    rt::Scene scene;
    // the groups and CSG nodes that are open, outermost first
    struct Open {
        bool is_csg = false;
        rt::Op op = rt::Op::Union;
        rt::Props props;
        std::vector<rt::ShapePtr> children;
    };
    std::vector<Open> open;
};

namespace {

rt::Matrix4x4 matrix_of(const double* m16) {
    if (!m16) return rt::Matrix4x4::identity();
    rt::Matrix4x4 m;
    std::memcpy(m.m, m16, sizeof m.m);
    return m;
}

rt::Material material_of(const rth_material* in) {
    rt::Material mt;  // Material::new(), src/material.rs:20-33
    if (!in) return mt;
    mt.color = {in->color[0], in->color[1], in->color[2]};
    mt.ambient = in->ambient, mt.diffuse = in->diffuse, mt.specular = in->specular, mt.shininess = in->shininess;
    mt.reflective = in->reflective, mt.transparency = in->transparency, mt.refractive_index = in->refractive_index;
    if (in->pattern != RTH_PAT_NONE) {
        if (in->pattern < RTH_PAT_STRIPE || in->pattern > RTH_PAT_TEST) throw rt::Error("material: pattern kind out of range");
        rt::Pattern p;
        p.design = (rt::PatternDesign)(in->pattern - RTH_PAT_STRIPE);
        p.a = {in->pat_a[0], in->pat_a[1], in->pat_a[2]}, p.b = {in->pat_b[0], in->pat_b[1], in->pat_b[2]};
        p.transform = matrix_of(in->pat_transform);
        mt.set_pattern(p);
    }
    return mt;
}

void place(rth_world* w, rt::ShapePtr s) {
    if (w->open.empty()) {
        w->scene.world.objects.push_back(std::move(s));
        return;
    }
    rth_world::Open& top = w->open.back();
    if (top.is_csg && top.children.size() >= 2) throw rt::Error("a CSG node has exactly two children (left, right)");
    top.children.push_back(std::move(s));
}

template <class S>
void add_leaf(rth_world* w, S shape, const double* transform, const rth_material* material, int shadow) {
    shape.props.transform = matrix_of(transform);
    shape.props.material = material_of(material);
    shape.props.shadow = shadow != 0;
    place(w, shape.shape());
}

rt::Tuple tuple_at(const double* p, int k) { return rt::Tuple{p[4 * k], p[4 * k + 1], p[4 * k + 2], p[4 * k + 3]}; }

template <class F>
int guarded(rth_world* w, F&& f) {
    if (!w) {
        rth_set_error("world is NULL");
        return -1;
    }
    try {
        f();
        return 0;
    } catch (const std::bad_alloc&) {
        rth_set_error("out of host memory");
    } catch (const std::exception& e) {
        rth_set_error(e.what());
    } catch (...) {
        rth_set_error("unexpected failure");
    }
    return -1;
}

}  // namespace

extern "C" {

void rth_material_default(rth_material* out) {
    if (!out) return;
    std::memset(out, 0, sizeof *out);
    const rt::Material mt;
    out->color[0] = mt.color.red, out->color[1] = mt.color.green, out->color[2] = mt.color.blue;
    out->ambient = mt.ambient, out->diffuse = mt.diffuse, out->specular = mt.specular, out->shininess = mt.shininess;
    out->reflective = mt.reflective, out->transparency = mt.transparency, out->refractive_index = mt.refractive_index;
    out->pattern = RTH_PAT_NONE;
    const rt::Matrix4x4 id = rt::Matrix4x4::identity();
    std::memcpy(out->pat_transform, id.m, sizeof id.m);
}

rth_world* rth_world_new(void) {
    rth_world* w = new (std::nothrow) rth_world();
    if (!w) rth_set_error("out of host memory");
    return w;
}
void rth_world_free(rth_world* w) { delete w; }

int rth_add_sphere(rth_world* w, const double* transform, const rth_material* material, int shadow) {
    return guarded(w, [&] { add_leaf(w, rt::Sphere(), transform, material, shadow); });
}
int rth_add_plane(rth_world* w, const double* transform, const rth_material* material, int shadow) {
    return guarded(w, [&] { add_leaf(w, rt::Plane(), transform, material, shadow); });
}
int rth_add_cube(rth_world* w, const double* transform, const rth_material* material, int shadow) {
    return guarded(w, [&] { add_leaf(w, rt::Cube(), transform, material, shadow); });
}
int rth_add_cylinder(rth_world* w, const double* transform, const rth_material* material, int shadow, double min, double max, int closed) {
    return guarded(w, [&] { add_leaf(w, rt::Cylinder(min, max, closed != 0), transform, material, shadow); });
}
int rth_add_cone(rth_world* w, const double* transform, const rth_material* material, int shadow, double min, double max, int closed) {
    return guarded(w, [&] { add_leaf(w, rt::Cone(min, max, closed != 0), transform, material, shadow); });
}
int rth_add_triangle(rth_world* w, const double* transform, const rth_material* material, int shadow, const double* points) {
    return guarded(w, [&] {
        if (!points) throw rt::Error("triangle: points is NULL");
        add_leaf(w, rt::Triangle(tuple_at(points, 0), tuple_at(points, 1), tuple_at(points, 2)), transform, material, shadow);
    });
}
int rth_add_smooth_triangle(rth_world* w, const double* transform, const rth_material* material, int shadow, const double* points,
                            const double* normals) {
    return guarded(w, [&] {
        if (!points || !normals) throw rt::Error("smooth triangle: points or normals is NULL");
        add_leaf(w, rt::Triangle::smooth(tuple_at(points, 0), tuple_at(points, 1), tuple_at(points, 2), tuple_at(normals, 0), tuple_at(normals, 1), tuple_at(normals, 2)),
                 transform, material, shadow);
    });
}

int rth_group_begin(rth_world* w, const double* transform, const rth_material* material, int shadow) {
    return guarded(w, [&] {
        if (!w->open.empty() && w->open.back().is_csg && w->open.back().children.size() >= 2)
            throw rt::Error("a CSG node has exactly two children (left, right)");
        rth_world::Open o;
        o.props.transform = matrix_of(transform), o.props.material = material_of(material), o.props.shadow = shadow != 0;
        w->open.push_back(std::move(o));
    });
}
int rth_group_end(rth_world* w) {
    return guarded(w, [&] {
        if (w->open.empty() || w->open.back().is_csg) throw rt::Error("rth_group_end without an open group");
        rth_world::Open o = std::move(w->open.back());
        w->open.pop_back();
        rt::Group g;
        g.props = o.props;
        for (auto& c : o.children) g.add_child(std::move(c));
        place(w, g.shape());
    });
}
int rth_csg_begin(rth_world* w, int op, const double* transform, const rth_material* material, int shadow) {
    return guarded(w, [&] {
        if (op != RTAMD_CSG_UNION && op != RTAMD_CSG_INTERSECTION && op != RTAMD_CSG_DIFFERENCE) throw rt::Error("csg: operation out of range");
        if (!w->open.empty() && w->open.back().is_csg && w->open.back().children.size() >= 2)
            throw rt::Error("a CSG node has exactly two children (left, right)");
        rth_world::Open o;
        o.is_csg = true;
        o.op = op == RTAMD_CSG_UNION ? rt::Op::Union : op == RTAMD_CSG_INTERSECTION ? rt::Op::Intersection : rt::Op::Difference;
        o.props.transform = matrix_of(transform), o.props.material = material_of(material), o.props.shadow = shadow != 0;
        w->open.push_back(std::move(o));
    });
}
int rth_csg_end(rth_world* w) {
    return guarded(w, [&] {
        if (w->open.empty() || !w->open.back().is_csg) throw rt::Error("rth_csg_end without an open CSG node");
        if (w->open.back().children.size() != 2) throw rt::Error("a CSG node has exactly two children (left, right)");
        rth_world::Open o = std::move(w->open.back());
        w->open.pop_back();
        rt::Csg c(o.op, std::move(o.children[0]), std::move(o.children[1]));
        c.props = o.props;
        place(w, c.shape());
    });
}

int rth_add_light(rth_world* w, const double position[3], const double intensity[3]) {
    return guarded(w, [&] {
        if (!position || !intensity) throw rt::Error("light: position or intensity is NULL");
        rt::PointLight l;
        l.position = rt::pt(position[0], position[1], position[2]);
        l.intensity = {intensity[0], intensity[1], intensity[2]};
        w->scene.world.lights.push_back(l);
    });
}
int rth_set_camera(rth_world* w, int hsize, int vsize, double field_of_view, const double* transform) {
    return guarded(w, [&] {
        if (hsize <= 0 || vsize <= 0) throw rt::Error("camera: the size must be positive");
        auto c = std::make_unique<rt::Camera>((size_t)hsize, (size_t)vsize, field_of_view);
        c->transform = matrix_of(transform);
        w->scene.camera = std::move(c);
    });
}

void* rth_world_scene(rth_world* w) {
    if (!w) {
        rth_set_error("world is NULL");
        return nullptr;
    }
    if (!w->open.empty()) {
        rth_set_error("a group or CSG node is still open");
        return nullptr;
    }
    return &w->scene;
}

}  // extern "C"
