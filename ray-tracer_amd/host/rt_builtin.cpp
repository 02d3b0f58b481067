// Scenes the reference builds in code rather than in YAML, restated with the host API.
// Each function names the reference binary that is its specification.
#include "rt_scene.hpp"

namespace rt {

static const double PI = 3.14159265358979323846;

static Material wall_material() {  // src/bin/reflect_refract.rs:10-18
    Material wm;
    Pattern wp = stripe_pattern({0.45, 0.45, 0.45}, {0.55, 0.55, 0.55});
    wp.transform = Matrix4x4::scaling(0.25, 0.25, 0.25) * Matrix4x4::rotation_y(1.5708);
    wm.set_pattern(wp);
    wm.ambient = 0.0, wm.diffuse = 0.4, wm.specular = 0.0, wm.reflective = 0.3;
    return wm;
}
static void add_room(World& world) {  // src/bin/reflect_refract.rs:20-69
    Material fm;
    fm.set_pattern(checkers_pattern({0.35, 0.35, 0.35}, {0.65, 0.65, 0.65}));
    fm.specular = 0.0, fm.reflective = 0.4;
    world.objects.push_back(Plane().transform(Matrix4x4::rotation_y(0.31415)).material(fm).shape());
    Material cm;
    cm.color = {0.8, 0.8, 0.8};
    cm.ambient = 0.3, cm.specular = 0.0;
    world.objects.push_back(Plane().transform(Matrix4x4::translation(0, 5, 0)).material(cm).shape());
    Material wm = wall_material();
    world.objects.push_back(Plane().transform(id().rotate_y(1.5708).rotate_z(1.5708).translate(-5, 0, 0)).material(wm).shape());
    world.objects.push_back(Plane().transform(id().rotate_y(1.5708).rotate_z(1.5708).translate(5, 0, 0)).material(wm).shape());
    world.objects.push_back(Plane().transform(id().rotate_x(1.5708).translate(0, 0, 5)).material(wm).shape());
    world.objects.push_back(Plane().transform(id().rotate_x(1.5708).translate(0, 0, -5)).material(wm).shape());
}
static Material shiny(double r, double g, double b) {
    Material x;
    x.color = {r, g, b};
    x.shininess = 50.0;
    return x;
}
static Material glass(double r, double g, double b) {  // src/bin/reflect_refract.rs:101-110
    Material x;
    x.color = {r, g, b};
    x.ambient = 0.0, x.diffuse = 0.4, x.specular = 0.9, x.shininess = 300.0;
    x.reflective = 0.9, x.transparency = 0.9, x.refractive_index = 1.5;
    return x;
}

// src/bin/reflect_refract.rs:4-144. With `teapot_obj` non-empty the red sphere is replaced by
// the OBJ group exactly as the commented block :89-97 does (the render the reference's author
// timed at "2+ hours").
std::unique_ptr<Scene> builtin_reflect_refract(const std::string& teapot_obj) {
    auto sc = std::make_unique<Scene>();
    sc->camera = std::make_unique<Camera>(400, 200, 1.152);
    sc->camera->transform = view_transform(pt(-2.6, 1.5, -3.9), pt(-0.6, 1.0, -0.8), v(0, 1, 0));
    World& w = sc->world;
    w.lights.push_back({pt(-4.9, 4.9, -1.0), {1, 1, 1}});
    add_room(w);
    w.objects.push_back(Sphere().transform(id().scale(0.4, 0.4, 0.4).translate(4.6, 0.4, 1.0)).material(shiny(0.8, 0.5, 0.3)).shape());
    w.objects.push_back(Sphere().transform(id().scale(0.3, 0.3, 0.3).translate(4.7, 0.3, 0.4)).material(shiny(0.9, 0.4, 0.5)).shape());
    w.objects.push_back(Sphere().transform(id().scale(0.5, 0.5, 0.5).translate(-1.0, 0.5, 4.5)).material(shiny(0.4, 0.9, 0.6)).shape());
    w.objects.push_back(Sphere().transform(id().scale(0.3, 0.3, 0.3).translate(-1.7, 0.3, 4.7)).material(shiny(0.4, 0.6, 0.9)).shape());
    if (teapot_obj.empty()) {
        Material red;
        red.color = {1.0, 0.3, 0.2};
        red.specular = 0.4, red.shininess = 5.0;
        w.objects.push_back(Sphere().transform(id().translate(-0.6, 1.0, 0.6)).material(red).shape());
    } else {
        Group teapot = ObjParser::from_file(teapot_obj).into_group();
        teapot.transform(id().rotate_x(-PI / 2.0).scale(0.07, 0.07, 0.07).translate(-0.6, 0.0, 0.6));
        w.objects.push_back(teapot.shape());
    }
    w.objects.push_back(Sphere().transform(id().scale(0.7, 0.7, 0.7).translate(0.6, 0.7, -0.6)).material(glass(0.0, 0.0, 0.2)).shape());
    w.objects.push_back(Sphere().transform(id().scale(0.5, 0.5, 0.5).translate(-0.7, 0.5, -0.8)).material(glass(0.0, 0.2, 0.0)).shape());
    return sc;
}

// src/bin/teapot.rs:5-20
std::unique_ptr<Scene> builtin_teapot(const std::string& teapot_obj) {
    auto sc = std::make_unique<Scene>();
    sc->camera = std::make_unique<Camera>(100, 100, PI / 2.0);
    sc->camera->transform = view_transform(pt(0, 14, -30), pt(0, 0, 0), v(0, 1, 0));
    sc->world.lights.push_back({pt(10, 20, -40), {1, 1, 1}});
    Group g = ObjParser::from_file(teapot_obj).into_group();
    g.transform(Matrix4x4::rotation_x(-PI / 2.0));
    sc->world.objects.push_back(g.shape());
    return sc;
}

// CSG divergence-stress scene. Core = src/bin/csg.rs:5-91 (a red cube minus a green closed
// cylinder, next to a green "holder" cylinder; same camera and light); added for depth: an
// Intersection(Sphere, Cube) lens of glass and the reflective room of
// src/bin/reflect_refract.rs:20-69. CSG nodes keep identity transforms, as in csg.rs.
std::unique_ptr<Scene> builtin_csg() {
    auto sc = std::make_unique<Scene>();
    sc->camera = std::make_unique<Camera>(400, 200, 1.152);
    sc->camera->transform = view_transform(pt(-2.6, 1.5, -3.9), pt(-0.6, 1.0, -0.8), v(0, 1, 0));
    World& w = sc->world;
    w.lights.push_back({pt(-4.9, 4.9, -1.0), {1, 1, 1}});
    add_room(w);
    Material red = Material().rgb(1, 0, 0), green = Material().rgb(0, 1, 0);
    ShapePtr cube = Cube().material(red).shape();
    ShapePtr cyl = Cylinder(-2.0, 2.0, true).material(green).shape();
    w.objects.push_back(Csg::difference(std::move(cube), std::move(cyl)).shape());
    w.objects.push_back(Cylinder(-4.0, 4.0, true).transform(id().scale(0.5, 0.5, 0.5).rotate_x(PI / 2.0)).material(green).shape());
    Material g = glass(0.05, 0.05, 0.2);
    ShapePtr ls = Sphere().transform(id().translate(-2.2, 1.0, -1.0)).material(g).shape();
    ShapePtr lc = Cube().transform(id().scale(0.7, 0.7, 0.7).rotate_y(0.5).translate(-2.2, 1.0, -1.0)).material(g).shape();
    w.objects.push_back(Csg::intersection(std::move(ls), std::move(lc)).shape());
    return sc;
}

// The default world looked at by the 11x11 camera of the reference's `camera_render` test
// (src/scene.rs:139-155, 458-468).
std::unique_ptr<Scene> builtin_default_world() {
    auto sc = std::make_unique<Scene>();
    sc->camera = std::make_unique<Camera>(11, 11, PI / 2.0);
    sc->camera->transform = view_transform(pt(0, 0, -5), pt(0, 0, 0), v(0, 1, 0));
    sc->world = World::default_world();
    return sc;
}

}  // namespace rt
