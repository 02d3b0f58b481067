// TEST INFRASTRUCTURE ONLY -- see rt_oracle.hpp for the role of this file.
// Build with -O2 -ffp-contract=off (Rust never fuses a multiply and an add).
#include "rt_oracle.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <limits>
#include <map>
#include <unordered_map>
#include <sstream>
#include <stdexcept>

namespace orc {

// ------------------------------------------------------------------ tuples (src/tuple.rs)
Tuple pt(double x, double y, double z) { return {x, y, z, 1.0}; }   // src/tuple.rs:28-30
Tuple vec(double x, double y, double z) { return {x, y, z, 0.0}; }  // src/tuple.rs:32-34
Tuple operator+(Tuple a, Tuple b) { return {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }  // :102-113
Tuple operator-(Tuple a, Tuple b) { return {a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w}; }  // :115-126
Tuple operator-(Tuple a) { return {-a.x, -a.y, -a.z, -a.w}; }                                // :128-139
Tuple operator*(Tuple a, double s) { return {a.x * s, a.y * s, a.z * s, a.w * s}; }          // :144-155
// src/tuple.rs:167-180 -- the cross product keeps the left operand's w
Tuple cross(Tuple a, Tuple b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x, a.w}; }
double dot(Tuple a, Tuple b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }  // :70-72
double magnitude(Tuple a) { return std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w); }  // :47-49
Tuple normalize(Tuple a) {  // src/tuple.rs:51-59
    double m = magnitude(a);
    return {a.x / m, a.y / m, a.z / m, a.w / m};
}
Tuple reflect(Tuple v, Tuple n) { return v - n * 2.0 * dot(v, n); }  // src/tuple.rs:74-76
static bool feq(double a, double b) { return std::fabs(a - b) < EPSILON; }  // src/lib.rs:28-30
bool tuple_eq(Tuple a, Tuple b) { return feq(a.x, b.x) && feq(a.y, b.y) && feq(a.z, b.z) && feq(a.w, b.w); }
bool color_eq(Color a, Color b) { return feq(a.r, b.r) && feq(a.g, b.g) && feq(a.b, b.b); }  // :239-245
static Color operator+(Color a, Color b) { return {a.r + b.r, a.g + b.g, a.b + b.b}; }
static Color operator-(Color a, Color b) { return {a.r - b.r, a.g - b.g, a.b - b.b}; }
static Color operator*(Color a, double s) { return {a.r * s, a.g * s, a.b * s}; }
static Color operator*(Color a, Color b) { return {a.r * b.r, a.g * b.g, a.b * b.b}; }  // Hadamard :295-307

// ------------------------------------------------------------------ matrices (src/matrix.rs)
M4 identity() {
    M4 r{};
    for (int i = 0; i < 4; i++) r.m[i][i] = 1.0;
    return r;
}
M4 translation(double x, double y, double z) {  // :99-105
    M4 r = identity();
    r.m[0][3] = x, r.m[1][3] = y, r.m[2][3] = z;
    return r;
}
M4 scaling(double x, double y, double z) {  // :107-113
    M4 r = identity();
    r.m[0][0] = x, r.m[1][1] = y, r.m[2][2] = z;
    return r;
}
M4 rotation_x(double a) {  // :115-122
    M4 r = identity();
    r.m[1][1] = std::cos(a), r.m[1][2] = -std::sin(a), r.m[2][1] = std::sin(a), r.m[2][2] = std::cos(a);
    return r;
}
M4 rotation_y(double a) {  // :124-131
    M4 r = identity();
    r.m[0][0] = std::cos(a), r.m[0][2] = std::sin(a), r.m[2][0] = -std::sin(a), r.m[2][2] = std::cos(a);
    return r;
}
M4 rotation_z(double a) {  // :133-140
    M4 r = identity();
    r.m[0][0] = std::cos(a), r.m[0][1] = -std::sin(a), r.m[1][0] = std::sin(a), r.m[1][1] = std::cos(a);
    return r;
}
M4 shearing(double xy, double xz, double yx, double yz, double zx, double zy) {  // :142-151
    M4 r = identity();
    r.m[0][1] = xy, r.m[0][2] = xz, r.m[1][0] = yx, r.m[1][2] = yz, r.m[2][0] = zx, r.m[2][1] = zy;
    return r;
}
M4 mul(const M4& a, const M4& b) {  // :296-311
    M4 r;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++)
            r.m[i][j] = a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j] + a.m[i][2] * b.m[2][j] + a.m[i][3] * b.m[3][j];
    return r;
}
Tuple mul(const M4& a, Tuple t) {  // :313-326
    double b[4] = {t.x, t.y, t.z, t.w}, m[4];
    for (int r = 0; r < 4; r++) m[r] = a.m[r][0] * b[0] + a.m[r][1] * b[1] + a.m[r][2] * b[2] + a.m[r][3] * b[3];
    return {m[0], m[1], m[2], m[3]};
}
M4 transpose(const M4& a) {  // :174-183
    M4 r;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) r.m[i][j] = a.m[j][i];
    return r;
}
// 2x2 / 3x3 / 4x4 determinants by first-row cofactor expansion, src/matrix.rs:8-10,24-71,185-225
static double det2(const double s[2][2]) { return s[0][0] * s[1][1] - s[0][1] * s[1][0]; }
static double minor3(const double a[3][3], int row, int col) {
    double s[2][2];
    int sr = 0;
    for (int r = 0; r < 3; r++) {
        if (r == row) continue;
        int sc = 0;
        for (int c = 0; c < 3; c++) {
            if (c == col) continue;
            s[sr][sc++] = a[r][c];
        }
        sr++;
    }
    return det2(s);
}
static double cofactor3(const double a[3][3], int row, int col) {
    double sign = ((row + col) % 2 == 0) ? 1.0 : -1.0;
    return minor3(a, row, col) * sign;
}
static double det3(const double a[3][3]) {
    double d = 0.0;
    for (int c = 0; c < 3; c++) d += a[0][c] * cofactor3(a, 0, c);
    return d;
}
static double minor4(const M4& a, int row, int col) {
    double s[3][3];
    int sr = 0;
    for (int r = 0; r < 4; r++) {
        if (r == row) continue;
        int sc = 0;
        for (int c = 0; c < 4; c++) {
            if (c == col) continue;
            s[sr][sc++] = a.m[r][c];
        }
        sr++;
    }
    return det3(s);
}
static double cofactor4(const M4& a, int row, int col) {
    double sign = ((row + col) % 2 == 0) ? 1.0 : -1.0;
    return minor4(a, row, col) * sign;
}
double determinant(const M4& a) {  // :217-225
    double d = 0.0;
    for (int c = 0; c < 4; c++) d += a.m[0][c] * cofactor4(a, 0, c);
    return d;
}
bool inverse(const M4& a, M4* out) {  // :231-250
    double d = determinant(a);
    if (d == 0.0) return false;
    double cof[4][4];
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) cof[r][c] = cofactor4(a, r, c);
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) out->m[c][r] = cof[r][c] / d;
    return true;
}
bool m4_eq(const M4& a, const M4& b) {  // :265-279
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++)
            if (!feq(a.m[r][c], b.m[r][c])) return false;
    return true;
}
M4 view_transform(Tuple from, Tuple to, Tuple up) {  // src/scene.rs:157-169
    Tuple forward = normalize(to - from);
    Tuple upn = normalize(up);
    Tuple left = cross(forward, upn);
    Tuple true_up = cross(left, forward);
    M4 o = identity();
    o.m[0][0] = left.x, o.m[0][1] = left.y, o.m[0][2] = left.z;
    o.m[1][0] = true_up.x, o.m[1][1] = true_up.y, o.m[1][2] = true_up.z;
    o.m[2][0] = -forward.x, o.m[2][1] = -forward.y, o.m[2][2] = -forward.z;
    return mul(o, translation(-from.x, -from.y, -from.z));
}

// The reference panics on a singular transform (`inverse().unwrap()`, src/shape.rs:40).
static M4 inv_or_throw(Ctx& c, const M4& m) {
    M4 r;
    c.stats.inversions++;
    if (!inverse(m, &r)) throw std::runtime_error("singular transform (reference panics: inverse().unwrap())");
    return r;
}

// ------------------------------------------------------------------ materials / equality
static bool pattern_eq(const Pattern& a, const Pattern& b) {  // derived PartialEq, src/pattern.rs:3-16
    if (a.kind != b.kind) return false;
    if (a.kind == PAT_NONE) return true;
    if (a.kind != PAT_TEST && !(color_eq(a.a, b.a) && color_eq(a.b, b.b))) return false;
    return m4_eq(a.transform, b.transform);
}
bool material_eq(const Material& a, const Material& b) {  // derived PartialEq, src/material.rs:7-18
    return color_eq(a.color, b.color) && pattern_eq(a.pattern, b.pattern) && a.ambient == b.ambient &&
           a.diffuse == b.diffuse && a.specular == b.specular && a.shininess == b.shininess &&
           a.reflective == b.reflective && a.transparency == b.transparency &&
           a.refractive_index == b.refractive_index;
}
// `PartialEq for dyn Shape`: Props (transform, material) and the concrete type only,
// src/shape.rs:23-27,119-123; Group also compares the child count (src/shapes/group.rs:67-74),
// Csg compares its children recursively (src/shapes/csg.rs:119-124).
bool shape_eq(const Shape& a, const Shape& b) {
    if (!(m4_eq(a.transform, b.transform) && material_eq(a.material, b.material))) return false;
    if (a.kind != b.kind) return false;
    if (a.kind == GROUP) return a.children.size() == b.children.size();
    if (a.kind == CSG) return shape_eq(*a.left, *b.left) && shape_eq(*a.right, *b.right);
    return true;
}
bool includes(const Shape& a, const Shape& other) {
    if (a.kind == GROUP) {  // src/shapes/group.rs:91-98
        for (auto& ch : a.children)
            if (includes(*ch, other)) return true;
        return false;
    }
    if (a.kind == CSG) return includes(*a.left, other) || includes(*a.right, other);  // csg.rs:126-128
    return shape_eq(a, other);  // e.g. src/shapes/sphere.rs:84-86
}
void refresh_parents(Shape& s) {  // src/shapes/group.rs:76-89; every other shape: no-op (src/shape.rs:98)
    if (s.kind != GROUP) return;
    Material material = s.material;
    std::vector<M4> child_transforms = s.parent_transforms;
    child_transforms.push_back(s.transform);
    Material dflt;
    for (auto& ch : s.children) {
        if (material_eq(ch->material, dflt)) ch->material = material;
        ch->parent_transforms = child_transforms;
        if (ch->kind == GROUP) refresh_parents(*ch);
    }
}

std::unique_ptr<Shape> make_shape(ShapeKind k) {
    auto s = std::make_unique<Shape>();
    s->kind = k;
    return s;
}
std::unique_ptr<Shape> make_cyl(ShapeKind k, double mn, double mx, bool closed) {
    auto s = make_shape(k);
    s->minv = mn, s->maxv = mx, s->closed = closed;
    return s;
}
std::unique_ptr<Shape> make_triangle(Tuple p1, Tuple p2, Tuple p3) {  // src/shapes/triangle.rs:22-35
    auto s = make_shape(TRIANGLE);
    s->p1 = p1, s->p2 = p2, s->p3 = p3;
    s->e1 = p2 - p1;
    s->e2 = p3 - p1;
    s->normal = normalize(cross(s->e2, s->e1));
    return s;
}
std::unique_ptr<Shape> make_smooth_triangle(Tuple p1, Tuple p2, Tuple p3, Tuple n1, Tuple n2, Tuple n3) {  // :37-49
    auto s = make_shape(TRIANGLE);
    s->p1 = p1, s->p2 = p2, s->p3 = p3;
    s->e1 = p2 - p1;
    s->e2 = p3 - p1;
    s->smooth = true;
    s->n1 = n1, s->n2 = n2, s->n3 = n3;
    return s;
}
std::unique_ptr<Shape> make_csg(CsgOp op, std::unique_ptr<Shape> l, std::unique_ptr<Shape> r) {
    auto s = make_shape(CSG);
    s->op = op;
    s->left = std::move(l);
    s->right = std::move(r);
    return s;
}
std::unique_ptr<Shape> glass_sphere() {  // src/shapes/sphere.rs:16-21
    auto s = make_shape(SPHERE);
    s->material.transparency = 1.0;
    s->material.refractive_index = 1.5;
    return s;
}

// ------------------------------------------------------------------ rays / intersections
Ray ray_transform(const Ray& r, const M4& m) { return {mul(m, r.origin), mul(m, r.direction)}; }  // src/ray.rs:18-23
Tuple ray_position(const Ray& r, double t) { return r.origin + r.direction * t; }                   // src/ray.rs:14-16
Intersection isect(double t, const Shape* s) { return {t, s, false, 0, 0}; }
Intersection isect_uv(double t, const Shape* s, double u, double v) { return {t, s, true, u, v}; }

// Rust's sort_by is a stable merge sort; `partial_cmp().unwrap()` panics on NaN (src/scene.rs:24).
static void sort_xs(Xs& xs) {
    for (auto& i : xs)
        if (std::isnan(i.t)) throw std::runtime_error("NaN t in intersection list (reference panics in sort_by)");
    std::stable_sort(xs.begin(), xs.end(), [](const Intersection& a, const Intersection& b) { return a.t < b.t; });
}

const Intersection* hit(const Xs& xs) {  // src/intersection.rs:135-147
    for (auto& i : xs)
        if (i.t > 0.0) return &i;
    return nullptr;
}

static double powi2(double x) { return x * x; }

// src/shapes/cube.rs:31-49
static void check_axis(double origin, double direction, double* tmin, double* tmax) {
    double tmin_n = -1.0 - origin;
    double tmax_n = 1.0 - origin;
    double a, b;
    if (std::fabs(direction) >= EPSILON) {
        a = tmin_n / direction;
        b = tmax_n / direction;
    } else {
        a = tmin_n * std::numeric_limits<double>::infinity();
        b = tmax_n * std::numeric_limits<double>::infinity();
    }
    if (a > b) std::swap(a, b);
    *tmin = a;
    *tmax = b;
}

// src/shapes/cylinder.rs:47-74 (cone == false) and src/shapes/cone.rs:47-74 (cone == true)
static void intersect_caps(const Shape& s, const Ray& ray, Xs& xs, bool cone) {
    double dy = cone ? std::fabs(ray.direction.y) : ray.direction.y;
    if (!s.closed || feq(dy, 0.0)) return;
    for (int k = 0; k < 2; k++) {
        double lim = k == 0 ? s.minv : s.maxv;
        double t = (lim - ray.origin.y) / ray.direction.y;
        double x = ray.origin.x + t * ray.direction.x;
        double z = ray.origin.z + t * ray.direction.z;
        double rad2 = cone ? powi2(std::fabs(lim)) : 1.0;
        if (powi2(x) + powi2(z) <= rad2) xs.push_back(isect(t, &s));
    }
}

Xs local_intersect(Ctx& c, const Shape& s, const Ray& ray) {
    Xs xs;
    if (s.kind != GROUP && s.kind != CSG) c.stats.leaf_tests++;
    switch (s.kind) {
        case SPHERE: {  // src/shapes/sphere.rs:39-55
            Tuple s2r = ray.origin - pt(0, 0, 0);
            double a = dot(ray.direction, ray.direction);
            double b = 2.0 * dot(ray.direction, s2r);
            double cc = dot(s2r, s2r) - 1.0;
            double disc = (b * b) - 4.0 * a * cc;
            if (disc < 0.0) return xs;
            double t1 = (-b - std::sqrt(disc)) / (2.0 * a);
            double t2 = (-b + std::sqrt(disc)) / (2.0 * a);
            xs.push_back(isect(t1, &s));
            xs.push_back(isect(t2, &s));
            return xs;
        }
        case PLANE: {  // src/shapes/plane.rs:32-39
            if (std::fabs(ray.direction.y) < EPSILON) return xs;
            xs.push_back(isect(-ray.origin.y / ray.direction.y, &s));
            return xs;
        }
        case CUBE: {  // src/shapes/cube.rs:52-66; f64::max/min ignore a NaN operand, as fmax/fmin do
            double xtmin, xtmax, ytmin, ytmax, ztmin, ztmax;
            check_axis(ray.origin.x, ray.direction.x, &xtmin, &xtmax);
            check_axis(ray.origin.y, ray.direction.y, &ytmin, &ytmax);
            check_axis(ray.origin.z, ray.direction.z, &ztmin, &ztmax);
            double tmin = std::fmax(std::fmax(xtmin, ytmin), ztmin);
            double tmax = std::fmin(std::fmin(xtmax, ytmax), ztmax);
            if (tmin > tmax) return xs;
            xs.push_back(isect(tmin, &s));
            xs.push_back(isect(tmax, &s));
            return xs;
        }
        case CYLINDER: {  // src/shapes/cylinder.rs:77-116
            double a = powi2(ray.direction.x) + powi2(ray.direction.z);
            if (feq(a, 0.0)) {
                intersect_caps(s, ray, xs, false);
                return xs;
            }
            double b = 2.0 * ray.origin.x * ray.direction.x + 2.0 * ray.origin.z * ray.direction.z;
            double cc = powi2(ray.origin.x) + powi2(ray.origin.z) - 1.0;
            double disc = powi2(b) - 4.0 * a * cc;
            if (disc < 0.0) return xs;
            double t0 = (-b - std::sqrt(disc)) / (2.0 * a);
            double t1 = (-b + std::sqrt(disc)) / (2.0 * a);
            if (t0 > t1) std::swap(t0, t1);
            double y0 = ray.origin.y + t0 * ray.direction.y;
            if (s.minv < y0 && y0 < s.maxv) xs.push_back(isect(t0, &s));
            double y1 = ray.origin.y + t1 * ray.direction.y;
            if (s.minv < y1 && y1 < s.maxv) xs.push_back(isect(t1, &s));
            intersect_caps(s, ray, xs, false);
            return xs;
        }
        case CONE: {  // src/shapes/cone.rs:77-121
            const Tuple &o = ray.origin, &d = ray.direction;
            double a = powi2(d.x) - powi2(d.y) + powi2(d.z);
            double b = 2.0 * o.x * d.x - 2.0 * o.y * d.y + 2.0 * o.z * d.z;
            double cc = powi2(o.x) - powi2(o.y) + powi2(o.z);
            if (feq(std::fabs(a), 0.0)) {
                if (feq(std::fabs(b), 0.0)) return xs;
                double t = -cc / (2.0 * b);
                xs.push_back(isect(t, &s));  // pushed without the min/max check, as the reference does
                intersect_caps(s, ray, xs, true);
                return xs;
            }
            double disc = powi2(b) - 4.0 * a * cc;
            if (disc < 0.0) return xs;
            double t0 = (-b - std::sqrt(disc)) / (2.0 * a);
            double t1 = (-b + std::sqrt(disc)) / (2.0 * a);
            if (t0 > t1) std::swap(t0, t1);
            double y0 = o.y + t0 * d.y;
            if (s.minv < y0 && y0 < s.maxv) xs.push_back(isect(t0, &s));
            double y1 = o.y + t1 * d.y;
            if (s.minv < y1 && y1 < s.maxv) xs.push_back(isect(t1, &s));
            intersect_caps(s, ray, xs, true);
            return xs;
        }
        case TRIANGLE: {  // src/shapes/triangle.rs:57-80
            Tuple dce2 = cross(ray.direction, s.e2);
            double det = dot(s.e1, dce2);
            if (std::fabs(det) < EPSILON) return xs;
            double f = 1.0 / det;
            Tuple p1o = ray.origin - s.p1;
            double u = f * dot(p1o, dce2);
            if (u < 0.0 || u > 1.0) return xs;
            Tuple oce1 = cross(p1o, s.e1);
            double v = f * dot(ray.direction, oce1);
            if (v < 0.0 || (u + v) > 1.0) return xs;
            double t = f * dot(s.e2, oce1);
            xs.push_back(isect_uv(t, &s, u, v));
            return xs;
        }
        case GROUP: {  // src/shapes/group.rs:37-45
            for (auto& ch : s.children) {
                Xs cx = intersect(c, *ch, ray);
                xs.insert(xs.end(), cx.begin(), cx.end());
            }
            sort_xs(xs);
            return xs;
        }
        case CSG: {  // src/shapes/csg.rs:90-97
            xs = intersect(c, *s.left, ray);
            Xs rx = intersect(c, *s.right, ray);
            xs.insert(xs.end(), rx.begin(), rx.end());
            sort_xs(xs);
            return csg_filter(s, xs);
        }
    }
    return xs;
}

// Cached inverses (cached mode only): same routine on the same input => same bits.
struct InvCache {
    std::unordered_map<const M4*, M4> m;
};
static thread_local InvCache* g_cache = nullptr;
static M4 inv_of(Ctx& c, const M4& m) {
    if (c.faithful || !g_cache) return inv_or_throw(c, m);
    auto it = g_cache->m.find(&m);
    if (it != g_cache->m.end()) return it->second;
    M4 r = inv_or_throw(c, m);
    g_cache->m.emplace(&m, r);
    return r;
}

Xs intersect(Ctx& c, const Shape& s, const Ray& ray) {  // src/shape.rs:39-42
    Ray local = ray_transform(ray, inv_of(c, s.transform));
    return local_intersect(c, s, local);
}

bool csg_intersection_allowed(CsgOp op, bool hit_left, bool inl, bool inr) {  // src/shapes/csg.rs:46-52
    switch (op) {
        case OP_UNION: return (hit_left && !inr) || (!hit_left && !inl);
        case OP_INTERSECTION: return (hit_left && inr) || (!hit_left && inl);
        case OP_DIFFERENCE: return (hit_left && !inr) || (!hit_left && inl);
    }
    return false;
}
Xs csg_filter(const Shape& csg, const Xs& xs) {  // src/shapes/csg.rs:54-82
    bool inl = false, inr = false;
    Xs result;
    for (auto& i : xs) {
        bool hit_left = includes(*csg.left, *i.object);
        if (csg_intersection_allowed(csg.op, hit_left, inl, inr)) result.push_back(i);
        if (hit_left)
            inl = !inl;
        else
            inr = !inr;
    }
    return result;
}

Tuple local_normal_at(const Shape& s, Tuple p, const Intersection& i) {
    switch (s.kind) {
        case SPHERE: return p - pt(0, 0, 0);  // src/shapes/sphere.rs:57-59
        case PLANE: return vec(0, 1, 0);      // src/shapes/plane.rs:41-43
        case CUBE: {                          // src/shapes/cube.rs:68-82
            double maxc = std::fmax(std::fmax(std::fabs(p.x), std::fabs(p.y)), std::fabs(p.z));
            if (maxc == std::fabs(p.x)) return vec(p.x, 0, 0);
            if (maxc == std::fabs(p.y)) return vec(0, p.y, 0);
            return vec(0, 0, p.z);
        }
        case CYLINDER: {  // src/shapes/cylinder.rs:118-128
            double dist = powi2(p.x) + powi2(p.z);
            if (dist < 1.0 && p.y >= s.maxv - EPSILON) return vec(0, 1, 0);
            if (dist < 1.0 && p.y <= s.minv + EPSILON) return vec(0, -1, 0);
            return vec(p.x, 0, p.z);
        }
        case CONE: {  // src/shapes/cone.rs:123-138 (the cap test really compares with 1.0)
            double dist = powi2(p.x) + powi2(p.z);
            if (dist < 1.0 && p.y >= s.maxv - EPSILON) return vec(0, 1, 0);
            if (dist < 1.0 && p.y <= s.minv + EPSILON) return vec(0, -1, 0);
            double y = std::sqrt(dist);
            if (p.y > 0.0) y = -y;
            return vec(p.x, y, p.z);
        }
        case TRIANGLE: {  // src/shapes/triangle.rs:82-93
            if (!s.smooth) return s.normal;
            if (!i.has_uv) throw std::runtime_error("uv not set on intersection!");
            return s.n2 * i.u + s.n3 * i.v + s.n1 * (1.0 - i.u - i.v);
        }
        default: throw std::runtime_error("local_normal_at on a group/csg (reference: unreachable!)");
    }
}
Tuple world_to_object(Ctx& c, const Shape& s, Tuple p) {  // src/shape.rs:49-55
    for (auto& t : s.parent_transforms) p = mul(inv_of(c, t), p);
    return mul(inv_of(c, s.transform), p);
}
static Tuple compute_normal(Ctx& c, const M4& transform, Tuple n) {  // src/shape.rs:58-63
    n = mul(transpose(inv_of(c, transform)), n);
    n.w = 0.0;
    return normalize(n);
}
Tuple normal_to_world(Ctx& c, const Shape& s, Tuple n) {  // src/shape.rs:56-70
    n = compute_normal(c, s.transform, n);
    for (auto it = s.parent_transforms.rbegin(); it != s.parent_transforms.rend(); ++it) n = compute_normal(c, *it, n);
    return n;
}
Tuple normal_at(Ctx& c, const Shape& s, Tuple wp, const Intersection& i) {  // src/shape.rs:44-48
    Tuple op = world_to_object(c, s, wp);
    Tuple on = local_normal_at(s, op, i);
    return normal_to_world(c, s, on);
}

Comps prepare_computations(Ctx& c, const Intersection& self, const Ray& ray, const Xs& xs) {  // src/intersection.rs:70-128
    Comps k;
    k.t = self.t;
    k.object = self.object;
    k.point = ray_position(ray, k.t);
    k.eyev = -ray.direction;
    Tuple n = normal_at(c, *k.object, k.point, self);
    k.inside = dot(n, k.eyev) < 0.0;
    k.normalv = k.inside ? -n : n;
    k.reflectv = reflect(ray.direction, k.normalv);
    k.over_point = k.point + k.normalv * EPSILON;
    k.under_point = k.point - k.normalv * EPSILON;
    k.n1 = 1.0, k.n2 = 1.0;
    std::vector<const Shape*> containers;
    for (auto& i : xs) {
        bool is_self = (i.t == self.t) && shape_eq(*i.object, *self.object);  // :47-51
        if (is_self) k.n1 = containers.empty() ? 1.0 : containers.back()->material.refractive_index;
        size_t pos = containers.size();
        for (size_t q = 0; q < containers.size(); q++)
            if (shape_eq(*containers[q], *i.object)) {
                pos = q;
                break;
            }
        if (pos < containers.size())
            containers.erase(containers.begin() + pos);
        else
            containers.push_back(i.object);
        if (is_self) {
            k.n2 = containers.empty() ? 1.0 : containers.back()->material.refractive_index;
            break;
        }
    }
    return k;
}

// compiler-rt's __powidf2 / LLVM's powi expansion for n = 5: x * ((x*x)*(x*x))
static double powi5(double x) {
    double x2 = x * x;
    double x4 = x2 * x2;
    return x * x4;
}
double schlick(const Comps& k) {  // src/intersection.rs:3-24
    double cosv = dot(k.eyev, k.normalv);
    if (k.n1 > k.n2) {
        double n = k.n1 / k.n2;
        double sin2_t = powi2(n) * (1.0 - powi2(cosv));
        if (sin2_t > 1.0) return 1.0;
        cosv = std::sqrt(1.0 - sin2_t);
    }
    double r0 = powi2((k.n1 - k.n2) / (k.n1 + k.n2));
    return r0 + (1.0 - r0) * powi5(1.0 - cosv);
}

// Rust `f64 as isize` saturates and maps NaN to 0; only the parity is used afterwards.
static bool even_as_isize(double v) {
    if (std::isnan(v)) return true;
    if (v >= 9223372036854775807.0) return false;   // isize::MAX is odd
    if (v <= -9223372036854775808.0) return true;   // isize::MIN is even
    long long q = (long long)v;  // truncation toward zero, as the Rust cast does
    return q % 2 == 0;
}
Color pattern_at(const Pattern& p, Tuple q) {  // src/pattern.rs:19-55
    switch (p.kind) {
        case PAT_STRIPE: return even_as_isize(std::floor(q.x)) ? p.a : p.b;
        case PAT_GRADIENT: {
            Color distance = p.b - p.a;
            double fraction = q.x - std::floor(q.x);
            return p.a + distance * fraction;
        }
        case PAT_RING: {
            double x2 = q.x * q.x, z2 = q.z * q.z;
            return even_as_isize(std::sqrt(x2 + z2)) ? p.a : p.b;
        }
        case PAT_CHECKERS: return even_as_isize(std::floor(q.x) + std::floor(q.y) + std::floor(q.z)) ? p.a : p.b;
        case PAT_TEST: return {q.x, q.y, q.z};
        default: return {0, 0, 0};
    }
}
Color pattern_at_object(Ctx& c, const Pattern& p, const Shape& s, Tuple wp) {  // src/pattern.rs:57-61
    Tuple op = world_to_object(c, s, wp);
    Tuple pp = mul(inv_of(c, p.transform), op);
    return pattern_at(p, pp);
}

Color lighting(Ctx& c, const Material& m, const Shape& object, const PointLight& light, Tuple point, Tuple eyev,
               Tuple normalv, bool in_shadow) {  // src/material.rs:81-133
    Color color = m.pattern.kind != PAT_NONE ? pattern_at_object(c, m.pattern, object, point) : m.color;
    Color effective = color * light.intensity;
    Tuple lightv = normalize(light.position - point);
    Color ambient = effective * m.ambient;
    if (in_shadow) return ambient;
    double ldn = dot(lightv, normalv);
    Color diffuse{0, 0, 0}, specular{0, 0, 0};
    if (!(ldn < 0.0)) {
        diffuse = effective * m.diffuse * ldn;
        Tuple reflectv = -reflect(lightv, normalv);
        double rde = dot(reflectv, eyev);
        if (!(rde <= 0.0)) {
            double factor = std::pow(rde, m.shininess);
            specular = light.intensity * m.specular * factor;
        }
    }
    return ambient + diffuse + specular;
}

// ------------------------------------------------------------------ world (src/scene.rs)
World default_world() {  // src/scene.rs:139-155
    World w;
    w.lights.push_back({pt(-10, 10, -10), {1, 1, 1}});
    auto s1 = make_shape(SPHERE);
    s1->material.color = {0.8, 1.0, 0.6};
    s1->material.diffuse = 0.7;
    s1->material.specular = 0.2;
    auto s2 = make_shape(SPHERE);
    s2->transform = scaling(0.5, 0.5, 0.5);
    w.objects.push_back(std::move(s1));
    w.objects.push_back(std::move(s2));
    return w;
}
void world_refresh_parents(World& w) {  // src/scene.rs:132-136
    for (auto& o : w.objects) refresh_parents(*o);
}
Xs world_intersect(Ctx& c, const World& w, const Ray& r) {  // src/scene.rs:19-26
    c.stats.rays++;
    Xs xs;
    for (auto& o : w.objects) {
        Xs ox = intersect(c, *o, r);
        xs.insert(xs.end(), ox.begin(), ox.end());
    }
    sort_xs(xs);
    return xs;
}
bool is_shadowed(Ctx& c, const World& w, const PointLight& light, Tuple point) {  // src/scene.rs:65-84
    Tuple v = light.position - point;
    double distance = magnitude(v);
    Tuple direction = normalize(v);
    Ray r{point, direction};
    Xs xs = world_intersect(c, w, r);
    Xs filtered;
    for (auto& i : xs)
        if (i.object->shadow) filtered.push_back(i);
    const Intersection* h = hit(filtered);
    return h && h->t < distance;
}
Color reflected_color(Ctx& c, const World& w, const Comps& k, int remaining) {  // src/scene.rs:86-96
    if (remaining == 0) return {0, 0, 0};
    if (k.object->material.reflective == 0.0) return {0, 0, 0};
    Ray rr{k.over_point, k.reflectv};
    Color col = color_at(c, w, rr, remaining - 1);
    return col * k.object->material.reflective;
}
Color refracted_color(Ctx& c, const World& w, const Comps& k, int remaining) {  // src/scene.rs:98-130
    if (remaining == 0) return {0, 0, 0};
    if (k.object->material.transparency == 0.0) return {0, 0, 0};
    double n_ratio = k.n1 / k.n2;
    double cos_i = dot(k.eyev, k.normalv);
    double sin2_t = n_ratio * n_ratio * (1.0 - cos_i * cos_i);
    if (sin2_t > 1.0) return {0, 0, 0};
    double cos_t = std::sqrt(1.0 - sin2_t);
    Tuple direction = k.normalv * (n_ratio * cos_i - cos_t) - k.eyev * n_ratio;
    Ray rr{k.under_point, direction};
    return color_at(c, w, rr, remaining - 1) * k.object->material.transparency;
}
Color shade_hit(Ctx& c, const World& w, const Comps& k, int remaining) {  // src/scene.rs:28-51
    c.stats.shade_points++;
    const Material& material = k.object->material;
    Color sum{0, 0, 0};
    Color reflected{0, 0, 0}, refracted{0, 0, 0};
    bool have_children = false;
    for (auto& l : w.lights) {
        Color surface =
            lighting(c, material, *k.object, l, k.over_point, k.eyev, k.normalv, is_shadowed(c, w, l, k.over_point));
        // The reference evaluates both children inside the per-light closure; they do not
        // depend on the light, so cached mode evaluates them once (same bits).
        if (c.faithful || !have_children) {
            reflected = reflected_color(c, w, k, remaining);
            refracted = refracted_color(c, w, k, remaining);
            have_children = true;
        }
        Color cl;
        if (material.reflective > 0.0 && material.transparency > 0.0) {
            double reflectance = schlick(k);
            cl = surface + reflected * reflectance + refracted * (1.0 - reflectance);
        } else {
            cl = surface + reflected + refracted;
        }
        sum = sum + cl;  // Color::sum: left fold from black, src/tuple.rs:259-267
    }
    return sum;
}
Color color_at(Ctx& c, const World& w, const Ray& r, int remaining) {  // src/scene.rs:53-63
    Xs xs = world_intersect(c, w, r);
    const Intersection* h = hit(xs);
    if (!h) return {0, 0, 0};
    Intersection i = *h;
    Comps k = prepare_computations(c, i, r, xs);
    return shade_hit(c, w, k, remaining);
}

// ------------------------------------------------------------------ camera (src/scene.rs:171-249)
Camera make_camera(int hsize, int vsize, double fov) {  // :182-209
    Camera cam;
    double h = (double)hsize, v = (double)vsize;
    double half_view = std::tan(fov / 2.0);
    double aspect = h / v;
    if (aspect >= 1.0) {
        cam.half_width = half_view;
        cam.half_height = half_view / aspect;
    } else {
        cam.half_width = half_view * aspect;
        cam.half_height = half_view;
    }
    cam.pixel_size = (cam.half_width * 2.0) / h;
    cam.hsize = hsize, cam.vsize = vsize, cam.field_of_view = fov;
    return cam;
}
Ray ray_for_pixel(Ctx& c, const Camera& cam, int px, int py) {  // :215-234
    double xoffset = ((double)px + 0.5) * cam.pixel_size;
    double yoffset = ((double)py + 0.5) * cam.pixel_size;
    double world_x = cam.half_width - xoffset;
    double world_y = cam.half_height - yoffset;
    Tuple pixel = mul(inv_of(c, cam.transform), pt(world_x, world_y, -1.0));
    Tuple origin = mul(inv_of(c, cam.transform), pt(0, 0, 0));
    Tuple direction = normalize(pixel - origin);
    return {origin, direction};
}
void color_to_u8(Color c, uint8_t out[3]) {  // src/tuple.rs:226-237
    double ch[3] = {c.r, c.g, c.b};
    for (int k = 0; k < 3; k++) {
        double r = std::round(ch[k] * 255.0);  // half away from zero, like f64::round
        long long q;
        if (std::isnan(r))
            q = 0;
        else if (r >= 9.2e18)
            q = 255;
        else if (r <= -9.2e18)
            q = 0;
        else
            q = (long long)r;
        out[k] = (uint8_t)std::min(255LL, std::max(0LL, q));
    }
}

// ------------------------------------------------------------------ .sg reader
namespace {
struct Tok {
    std::vector<std::string> t;
    size_t i = 0;
    const std::string& next() {
        if (i >= t.size()) throw std::runtime_error("sg: unexpected end of line");
        return t[i++];
    }
    double num() { return std::strtod(next().c_str(), nullptr); }
    int integer() { return std::atoi(next().c_str()); }
    void expect(const char* s) {
        if (next() != s) throw std::runtime_error(std::string("sg: expected '") + s + "'");
    }
};
struct SgReader {
    std::ifstream in;
    std::map<int, Material> mats;
    std::map<int, M4> xfs;
    bool read_line(Tok& tk) {
        std::string line;
        while (std::getline(in, line)) {
            size_t h = line.find('#');
            if (h != std::string::npos) line.resize(h);
            std::istringstream ss(line);
            tk.t.clear();
            tk.i = 0;
            std::string w;
            while (ss >> w) tk.t.push_back(w);
            if (!tk.t.empty()) return true;
        }
        return false;
    }
    const M4& xf(int id) {
        auto it = xfs.find(id);
        if (it == xfs.end()) throw std::runtime_error("sg: unknown xf id");
        return it->second;
    }
    std::unique_ptr<Shape> read_node() {
        Tok tk;
        if (!read_line(tk)) throw std::runtime_error("sg: missing node");
        tk.expect("node");
        std::string kind = tk.next();
        tk.expect("mat");
        int mid = tk.integer();
        tk.expect("xf");
        int xid = tk.integer();
        tk.expect("shadow");
        int shadow = tk.integer();
        std::unique_ptr<Shape> s;
        auto p3 = [&]() {
            double x = tk.num(), y = tk.num(), z = tk.num();
            return pt(x, y, z);
        };
        auto t4 = [&]() {
            double x = tk.num(), y = tk.num(), z = tk.num(), w = tk.num();
            return Tuple{x, y, z, w};
        };
        if (kind == "sphere")
            s = make_shape(SPHERE);
        else if (kind == "plane")
            s = make_shape(PLANE);
        else if (kind == "cube")
            s = make_shape(CUBE);
        else if (kind == "cylinder" || kind == "cone") {
            double mn = tk.num(), mx = tk.num();
            int closed = tk.integer();
            s = make_cyl(kind == "cone" ? CONE : CYLINDER, mn, mx, closed != 0);
        } else if (kind == "tri") {
            Tuple a = p3(), b = p3(), c = p3();
            s = make_triangle(a, b, c);
        } else if (kind == "stri") {
            Tuple a = p3(), b = p3(), c = p3();
            Tuple n1 = t4(), n2 = t4(), n3 = t4();
            s = make_smooth_triangle(a, b, c, n1, n2, n3);
        } else if (kind == "group") {
            int n = tk.integer();
            s = make_shape(GROUP);
            for (int k = 0; k < n; k++) s->children.push_back(read_node());
        } else if (kind == "csg") {
            std::string op = tk.next();
            CsgOp o = op == "union" ? OP_UNION : op == "intersection" ? OP_INTERSECTION : OP_DIFFERENCE;
            if (op != "union" && op != "intersection" && op != "difference") throw std::runtime_error("sg: bad csg op");
            auto l = read_node();
            auto r = read_node();
            s = make_csg(o, std::move(l), std::move(r));
        } else
            throw std::runtime_error("sg: unknown node kind " + kind);
        auto mi = mats.find(mid);
        if (mi == mats.end()) throw std::runtime_error("sg: unknown material id");
        s->material = mi->second;
        s->transform = xf(xid);
        s->shadow = shadow != 0;
        return s;
    }
};
}  // namespace

std::unique_ptr<Scene> load_sg(const std::string& path) {
    SgReader rd;
    rd.in.open(path);
    if (!rd.in) throw std::runtime_error("sg: cannot open " + path);
    rd.xfs[0] = identity();
    auto sc = std::make_unique<Scene>();
    Tok tk;
    if (!rd.read_line(tk) || tk.next() != "rtsg") throw std::runtime_error("sg: bad magic");
    bool have_cam = false;
    while (rd.read_line(tk)) {
        std::string cmd = tk.next();
        if (cmd == "camera") {
            int h = tk.integer(), v = tk.integer();
            double fov = tk.num();
            int xid = tk.integer();
            sc->camera = make_camera(h, v, fov);
            sc->camera.transform = rd.xf(xid);
            have_cam = true;
        } else if (cmd == "light") {
            PointLight l;
            double x = tk.num(), y = tk.num(), z = tk.num();
            l.position = pt(x, y, z);
            l.intensity.r = tk.num(), l.intensity.g = tk.num(), l.intensity.b = tk.num();
            sc->world.lights.push_back(l);
        } else if (cmd == "xf") {
            int id = tk.integer();
            M4 m;
            for (int r = 0; r < 4; r++)
                for (int c = 0; c < 4; c++) m.m[r][c] = tk.num();
            rd.xfs[id] = m;
        } else if (cmd == "mat") {
            int id = tk.integer();
            Material m;
            m.color.r = tk.num(), m.color.g = tk.num(), m.color.b = tk.num();
            m.ambient = tk.num(), m.diffuse = tk.num(), m.specular = tk.num(), m.shininess = tk.num();
            m.reflective = tk.num(), m.transparency = tk.num(), m.refractive_index = tk.num();
            std::string pk = tk.next();
            static const char* names[] = {"none", "stripe", "gradient", "ring", "checkers", "test"};
            int kind = -1;
            for (int k = 0; k < 6; k++)
                if (pk == names[k]) kind = k;
            if (kind < 0) throw std::runtime_error("sg: bad pattern kind");
            m.pattern.kind = (PatternKind)kind;
            if (kind != PAT_NONE) {
                m.pattern.a.r = tk.num(), m.pattern.a.g = tk.num(), m.pattern.a.b = tk.num();
                m.pattern.b.r = tk.num(), m.pattern.b.g = tk.num(), m.pattern.b.b = tk.num();
                m.pattern.transform = rd.xf(tk.integer());
            }
            rd.mats[id] = m;
        } else if (cmd == "objects") {
            int n = tk.integer();
            for (int k = 0; k < n; k++) sc->world.objects.push_back(rd.read_node());
        } else if (cmd == "end") {
            break;
        } else
            throw std::runtime_error("sg: unknown record " + cmd);
    }
    if (!have_cam) throw std::runtime_error("sg: no camera");
    return sc;
}

// ------------------------------------------------------------------ C entry points (tests, bench baseline)
void set_thread_cache(void* p) { g_cache = (InvCache*)p; }
void* new_thread_cache() { return new InvCache(); }
void free_thread_cache(void* p) { delete (InvCache*)p; }

}  // namespace orc

#include <thread>

extern "C" {

struct orc_stats {
    uint64_t rays, shade_points, leaf_tests, inversions;
};

static thread_local std::string g_err;
const char* orc_last_error() { return g_err.c_str(); }

void* orc_load_sg(const char* path) {
    try {
        return orc::load_sg(path).release();
    } catch (std::exception& e) {
        g_err = e.what();
        return nullptr;
    }
}
void orc_free_scene(void* s) { delete (orc::Scene*)s; }
void orc_scene_size(void* s, int* hsize, int* vsize) {
    auto* sc = (orc::Scene*)s;
    *hsize = sc->camera.hsize;
    *vsize = sc->camera.vsize;
}

// Render the pixels pix[0..n) (index = x + y*hsize of an hsize x vsize frame; hsize/vsize <= 0
// keep the scene's own resolution) exactly as Camera::render would (src/scene.rs:236-248,
// recursion budget 4).  out_rgb: 3 doubles per listed pixel; out_u8 (optional): 3 bytes each.
// mode 0 = cached, 1 = faithful.  threads > 1 splits the list (cached mode goldens only).
// Returns 0, or -1 with orc_last_error() set (singular transform, NaN t: the reference panics).
int orc_render_pixels(void* s, int mode, int hsize, int vsize, const int64_t* pix, int64_t n, double* out_rgb,
                      uint8_t* out_u8, int threads, orc_stats* st) {
    auto* sc = (orc::Scene*)s;
    orc::Camera cam = sc->camera;
    if (hsize > 0 && vsize > 0) {
        orc::M4 t = cam.transform;
        cam = orc::make_camera(hsize, vsize, cam.field_of_view);
        cam.transform = t;
    }
    orc::world_refresh_parents(sc->world);
    if (threads < 1) threads = 1;
    std::vector<orc::Stats> stats(threads);
    std::vector<std::string> errs(threads);
    auto work = [&](int tid) {
        orc::Ctx c;
        c.faithful = mode == 1;
        void* cache = orc::new_thread_cache();
        orc::set_thread_cache(cache);
        try {
            for (int64_t k = tid; k < n; k += threads) {
                int px = (int)(pix[k] % cam.hsize), py = (int)(pix[k] / cam.hsize);
                orc::Ray r = orc::ray_for_pixel(c, cam, px, py);
                orc::Color col = orc::color_at(c, sc->world, r, 4);  // src/scene.rs:243
                out_rgb[3 * k + 0] = col.r, out_rgb[3 * k + 1] = col.g, out_rgb[3 * k + 2] = col.b;
                if (out_u8) orc::color_to_u8(col, out_u8 + 3 * k);
            }
        } catch (std::exception& e) {
            errs[tid] = e.what();
        }
        orc::set_thread_cache(nullptr);
        orc::free_thread_cache(cache);
        stats[tid] = c.stats;
    };
    if (threads == 1)
        work(0);
    else {
        std::vector<std::thread> th;
        for (int t = 0; t < threads; t++) th.emplace_back(work, t);
        for (auto& t : th) t.join();
    }
    for (auto& e : errs)
        if (!e.empty()) {
            g_err = e;
            return -1;
        }
    if (st) {
        *st = {0, 0, 0, 0};
        for (auto& x : stats) {
            st->rays += x.rays, st->shade_points += x.shade_points, st->leaf_tests += x.leaf_tests;
            st->inversions += x.inversions;
        }
    }
    return 0;
}

// World::color_at(ray, remaining) (src/scene.rs:53-63) for n caller-supplied rays: rays = n x 6
// doubles (origin xyz, direction xyz), out_rgb = n x 3.  mode as above.  Returns 0 or -1.
int orc_color_at(void* s, int mode, const double* rays, int64_t n, int remaining, double* out_rgb) {
    auto* sc = (orc::Scene*)s;
    orc::world_refresh_parents(sc->world);
    orc::Ctx c;
    c.faithful = mode == 1;
    void* cache = orc::new_thread_cache();
    orc::set_thread_cache(cache);
    int rc = 0;
    try {
        for (int64_t k = 0; k < n; k++) {
            const double* q = rays + 6 * k;
            orc::Ray r{orc::pt(q[0], q[1], q[2]), orc::vec(q[3], q[4], q[5])};
            orc::Color col = orc::color_at(c, sc->world, r, remaining);
            out_rgb[3 * k + 0] = col.r, out_rgb[3 * k + 1] = col.g, out_rgb[3 * k + 2] = col.b;
        }
    } catch (std::exception& e) {
        g_err = e.what();
        rc = -1;
    }
    orc::set_thread_cache(nullptr);
    orc::free_thread_cache(cache);
    return rc;
}

// Diagnostic for tests: the first hit of one ray as Intersection::prepare_computations (src/intersection.rs:70-128)
// describes it.  out = t, point xyz, over_point xyz, normalv xyz, reflectv xyz, inside (0/1), kind of the shape hit,
// whether each light is shadowed at over_point (bit k of out[15] as a double) -- 16 doubles.  Returns 1 hit, 0 miss, -1 error.
int orc_first_hit(void* s, const double* ray, double* out) {
    auto* sc = (orc::Scene*)s;
    orc::world_refresh_parents(sc->world);
    orc::Ctx c;
    void* cache = orc::new_thread_cache();
    orc::set_thread_cache(cache);
    int rc = 0;
    try {
        orc::Ray r{orc::pt(ray[0], ray[1], ray[2]), orc::vec(ray[3], ray[4], ray[5])};
        orc::Xs xs = orc::world_intersect(c, sc->world, r);
        const orc::Intersection* h = orc::hit(xs);
        if (h) {
            orc::Intersection i = *h;
            orc::Comps k = orc::prepare_computations(c, i, r, xs);
            out[0] = k.t;
            out[1] = k.point.x, out[2] = k.point.y, out[3] = k.point.z;
            out[4] = k.over_point.x, out[5] = k.over_point.y, out[6] = k.over_point.z;
            out[7] = k.normalv.x, out[8] = k.normalv.y, out[9] = k.normalv.z;
            out[10] = k.reflectv.x, out[11] = k.reflectv.y, out[12] = k.reflectv.z;
            out[13] = k.inside ? 1.0 : 0.0;
            out[14] = (double)k.object->kind;
            unsigned bits = 0;
            for (size_t l = 0; l < sc->world.lights.size() && l < 32; l++)
                if (orc::is_shadowed(c, sc->world, sc->world.lights[l], k.over_point)) bits |= 1u << l;
            out[15] = (double)bits;
            rc = 1;
        }
    } catch (std::exception& e) {
        g_err = e.what();
        rc = -1;
    }
    orc::set_thread_cache(nullptr);
    orc::free_thread_cache(cache);
    return rc;
}

}  // extern "C"
