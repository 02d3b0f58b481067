// TEST INFRASTRUCTURE ONLY -- CPU oracle for the per-pixel render loop.
//
// A plain, single-threaded C++ (f64) restatement of the algorithm of the Rust reference
// (joshuabenuck/ray-tracer).  It exists so that the HIP path can be checked against it and
// so that a CPU baseline can be timed.  Nothing in the shipped library may include, link or
// call this file; only tests/, the smoke check and the bench's cpu_baseline leg do.
//
// Every function cites the reference file:line it follows (paths relative to the reference
// repository root).  Arithmetic keeps the reference's operand order and its 4-component
// tuples (w takes part in dot/magnitude/normalize) so that, compiled with
// -ffp-contract=off, the result is what IEEE f64 evaluation of the Rust source gives.
//
// Parity pinning: the known-answer tests of the reference's own unit tests are transcribed
// in kat_tests.cpp, and whole frames are compared with the renders the reference's author
// committed (tests/golden/ref_images).  The reference itself (Rust) cannot be built in this
// environment: there is no Rust toolchain.
//
// Two evaluation modes, bit-identical in their output:
//   faithful: re-inverts every transform at every use, re-evaluates reflection and
//             refraction once per light -- the work the reference really does; used as the
//             timed CPU baseline.
//   cached:   memoises each matrix inverse and evaluates the children of a shade point
//             once -- used to produce goldens quickly.
#pragma once
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

namespace orc {

constexpr double EPSILON = 0.00001;  // src/lib.rs:26

struct Tuple {
    double x, y, z, w;
};
struct Color {
    double r, g, b;
};
struct M4 {
    double m[4][4];
};

Tuple pt(double x, double y, double z);
Tuple vec(double x, double y, double z);
Tuple operator+(Tuple a, Tuple b);
Tuple operator-(Tuple a, Tuple b);
Tuple operator-(Tuple a);
Tuple operator*(Tuple a, double s);
Tuple cross(Tuple a, Tuple b);
double dot(Tuple a, Tuple b);
double magnitude(Tuple a);
Tuple normalize(Tuple a);
Tuple reflect(Tuple v, Tuple n);
bool tuple_eq(Tuple a, Tuple b);
bool color_eq(Color a, Color b);

M4 identity();
M4 translation(double x, double y, double z);
M4 scaling(double x, double y, double z);
M4 rotation_x(double r);
M4 rotation_y(double r);
M4 rotation_z(double r);
M4 shearing(double xy, double xz, double yx, double yz, double zx, double zy);
M4 mul(const M4& a, const M4& b);
Tuple mul(const M4& a, Tuple t);
M4 transpose(const M4& a);
double determinant(const M4& a);
bool inverse(const M4& a, M4* out);  // false when singular
bool m4_eq(const M4& a, const M4& b);
M4 view_transform(Tuple from, Tuple to, Tuple up);

enum PatternKind { PAT_NONE = 0, PAT_STRIPE, PAT_GRADIENT, PAT_RING, PAT_CHECKERS, PAT_TEST };
struct Pattern {
    PatternKind kind = PAT_NONE;
    Color a{0, 0, 0}, b{0, 0, 0};
    M4 transform = identity();
};
struct Material {
    Color color{1, 1, 1};
    Pattern pattern;
    double ambient = 0.1, diffuse = 0.9, specular = 0.9, shininess = 200.0;
    double reflective = 0.0, transparency = 0.0, refractive_index = 1.0;
};
bool material_eq(const Material& a, const Material& b);

enum ShapeKind { SPHERE = 0, PLANE, CUBE, CYLINDER, CONE, TRIANGLE, GROUP, CSG };
enum CsgOp { OP_UNION = 0, OP_INTERSECTION, OP_DIFFERENCE };

struct Ray {
    Tuple origin, direction;
};
struct PointLight {
    Tuple position;
    Color intensity;
};

struct Shape {
    ShapeKind kind = SPHERE;
    M4 transform = identity();
    Material material;
    std::vector<M4> parent_transforms;
    bool shadow = true;
    // cylinder / cone
    double minv = 0, maxv = 0;
    bool closed = false;
    // triangle
    Tuple p1{}, p2{}, p3{}, e1{}, e2{}, normal{};
    bool smooth = false;
    Tuple n1{}, n2{}, n3{};
    // group
    std::vector<std::unique_ptr<Shape>> children;
    // csg
    CsgOp op = OP_UNION;
    std::unique_ptr<Shape> left, right;
};

std::unique_ptr<Shape> make_shape(ShapeKind k);
std::unique_ptr<Shape> make_cyl(ShapeKind k, double mn, double mx, bool closed);
std::unique_ptr<Shape> make_triangle(Tuple p1, Tuple p2, Tuple p3);
std::unique_ptr<Shape> make_smooth_triangle(Tuple p1, Tuple p2, Tuple p3, Tuple n1, Tuple n2, Tuple n3);
std::unique_ptr<Shape> make_csg(CsgOp op, std::unique_ptr<Shape> l, std::unique_ptr<Shape> r);
std::unique_ptr<Shape> glass_sphere();
bool shape_eq(const Shape& a, const Shape& b);
bool includes(const Shape& a, const Shape& other);
void refresh_parents(Shape& s);

struct Intersection {
    double t;
    const Shape* object;
    bool has_uv;
    double u, v;
};
typedef std::vector<Intersection> Xs;
Intersection isect(double t, const Shape* s);
Intersection isect_uv(double t, const Shape* s, double u, double v);
const Intersection* hit(const Xs& xs);

struct Comps {
    double t;
    const Shape* object;
    Tuple point, eyev, normalv, reflectv;
    bool inside;
    Tuple over_point, under_point;
    double n1, n2;
};

struct Stats {
    uint64_t rays = 0;          // World::intersect calls actually executed
    uint64_t shade_points = 0;  // shade_hit calls
    uint64_t leaf_tests = 0;    // local_intersect calls on primitives
    uint64_t inversions = 0;    // 4x4 inversions actually computed
};

struct Ctx {
    bool faithful = false;
    Stats stats;
};

Xs local_intersect(Ctx& c, const Shape& s, const Ray& r);
Xs intersect(Ctx& c, const Shape& s, const Ray& r);
Tuple local_normal_at(const Shape& s, Tuple p, const Intersection& i);
Tuple world_to_object(Ctx& c, const Shape& s, Tuple p);
Tuple normal_to_world(Ctx& c, const Shape& s, Tuple n);
Tuple normal_at(Ctx& c, const Shape& s, Tuple p, const Intersection& i);
Comps prepare_computations(Ctx& c, const Intersection& i, const Ray& r, const Xs& xs);
double schlick(const Comps& c);
Color pattern_at(const Pattern& p, Tuple point);
Color pattern_at_object(Ctx& c, const Pattern& p, const Shape& s, Tuple world_point);
Color lighting(Ctx& c, const Material& m, const Shape& object, const PointLight& l, Tuple point, Tuple eyev,
               Tuple normalv, bool in_shadow);
bool csg_intersection_allowed(CsgOp op, bool hit_left, bool inl, bool inr);
Xs csg_filter(const Shape& csg, const Xs& xs);
Ray ray_transform(const Ray& r, const M4& m);
Tuple ray_position(const Ray& r, double t);

struct World {
    std::vector<std::unique_ptr<Shape>> objects;
    std::vector<PointLight> lights;
};
World default_world();
Xs world_intersect(Ctx& c, const World& w, const Ray& r);
Color shade_hit(Ctx& c, const World& w, const Comps& comps, int remaining);
Color color_at(Ctx& c, const World& w, const Ray& r, int remaining);
bool is_shadowed(Ctx& c, const World& w, const PointLight& l, Tuple p);
Color reflected_color(Ctx& c, const World& w, const Comps& comps, int remaining);
Color refracted_color(Ctx& c, const World& w, const Comps& comps, int remaining);
void world_refresh_parents(World& w);

struct Camera {
    int hsize = 0, vsize = 0;
    double field_of_view = 0;
    M4 transform = identity();
    double pixel_size = 0, half_width = 0, half_height = 0;
};
Camera make_camera(int hsize, int vsize, double fov);
Ray ray_for_pixel(Ctx& c, const Camera& cam, int px, int py);
void color_to_u8(Color c, uint8_t out[3]);

struct Scene {
    Camera camera;
    World world;
};
// Reads the scene-graph text format (".sg"); throws std::runtime_error on malformed input.
std::unique_ptr<Scene> load_sg(const std::string& path);

}  // namespace orc
