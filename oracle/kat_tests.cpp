// TEST INFRASTRUCTURE ONLY.
// Known-answer tests of the reference's own unit tests, restated against the oracle, so that
// the oracle is pinned to the numbers the reference's test-suite holds.  Each block cites the
// reference test it restates (file:line, paths relative to the reference repository).
// Expected values and tolerances are the reference's: exact `==` where the Rust test compares
// f64 directly, |a-b| < 1e-5 where it goes through the epsilon PartialEq of Tuple/Color/Matrix.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <limits>
#include <string>

#include "rt_oracle.hpp"

using namespace orc;

static int g_checks = 0, g_fail = 0;
#define CHECK(c)                                                         \
    do {                                                                 \
        g_checks++;                                                      \
        if (!(c)) {                                                      \
            g_fail++;                                                    \
            std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #c);     \
        }                                                                \
    } while (0)
#define CHECK_T(a, b) CHECK(tuple_eq((a), (b)))
#define CHECK_C(a, b) CHECK(color_eq((a), (b)))
static bool near(double a, double b) { return std::fabs(a - b) < EPSILON; }
static const double PI = 3.14159265358979323846;
static const double INF = std::numeric_limits<double>::infinity();

static M4 mat(std::initializer_list<double> v) {
    M4 m;
    int k = 0;
    for (double x : v) {
        m.m[k / 4][k % 4] = x;
        k++;
    }
    return m;
}

static void test_tuple() {  // src/tuple.rs:441-538,622-635
    CHECK_T((Tuple{3, -2, 5, 1} + Tuple{-2, 3, 1, 0}), (Tuple{1, 1, 6, 1}));
    CHECK_T(pt(3, 2, 1) - pt(5, 6, 7), vec(-2, -4, -6));
    CHECK_T(pt(3, 2, 1) - vec(5, 6, 7), pt(-2, -4, -6));
    CHECK_T(-(Tuple{1, -2, 3, -4}), (Tuple{-1, 2, -3, 4}));
    CHECK_T((Tuple{1, -2, 3, -4} * 3.5), (Tuple{3.5, -7, 10.5, -14}));
    CHECK(magnitude(vec(1, 2, 3)) == std::sqrt(14.0));
    CHECK(magnitude(vec(-1, -2, -3)) == std::sqrt(14.0));
    CHECK_T(normalize(vec(4, 0, 0)), vec(1, 0, 0));
    CHECK_T(normalize(vec(1, 2, 3)), vec(0.26726, 0.53452, 0.80178));
    CHECK(dot(vec(1, 2, 3), vec(2, 3, 4)) == 20.0);
    CHECK_T(cross(vec(1, 2, 3), vec(2, 3, 4)), vec(-1, 2, -1));
    CHECK_T(cross(vec(2, 3, 4), vec(1, 2, 3)), vec(1, -2, 1));
    CHECK_T(reflect(vec(1, -1, 0), vec(0, 1, 0)), vec(1, 1, 0));
    CHECK_T(reflect(vec(0, -1, 0), vec(std::sqrt(2.0) / 2, std::sqrt(2.0) / 2, 0)), vec(1, 0, 0));
    // 8-bit quantisation through the PPM expectations, src/tuple.rs:585-609
    uint8_t q[3];
    color_to_u8({1.5, 0.0, 0.0}, q);
    CHECK(q[0] == 255 && q[1] == 0 && q[2] == 0);
    color_to_u8({0.0, 0.5, 0.0}, q);
    CHECK(q[0] == 0 && q[1] == 128 && q[2] == 0);
    color_to_u8({-0.5, 0.0, 1.0}, q);
    CHECK(q[0] == 0 && q[1] == 0 && q[2] == 255);
    color_to_u8({1.0, 0.8, 0.6}, q);
    CHECK(q[0] == 255 && q[1] == 204 && q[2] == 153);
}

static void test_matrix() {  // src/matrix.rs:385-790
    M4 a = mat({1, 2, 3, 4, 5, 6, 7, 8, 9, 8, 7, 6, 5, 4, 3, 2});
    M4 b = mat({-2, 1, 2, 3, 3, 2, 1, -1, 4, 3, 6, 5, 1, 2, 7, 8});
    CHECK(m4_eq(mul(a, b), mat({20, 22, 50, 48, 44, 54, 114, 108, 40, 58, 110, 102, 16, 26, 46, 42})));
    CHECK_T(mul(mat({1, 2, 3, 4, 2, 4, 4, 2, 8, 6, 4, 1, 0, 0, 0, 1}), Tuple{1, 2, 3, 1}), (Tuple{18, 24, 33, 1}));
    CHECK(m4_eq(transpose(mat({0, 9, 3, 0, 9, 8, 0, 8, 1, 8, 5, 3, 0, 0, 5, 8})),
                mat({0, 9, 1, 0, 9, 8, 8, 0, 3, 0, 5, 5, 0, 8, 3, 8})));
    CHECK(determinant(mat({-2, -8, 3, 5, -3, 1, 7, 3, 1, 2, -9, 6, -6, 7, 7, -9})) == -4071.0);
    CHECK(determinant(mat({6, 4, 4, 4, 5, 5, 7, 6, 4, -9, 3, -7, 9, 1, 7, -6})) == -2120.0);
    M4 inv;
    CHECK(!inverse(mat({-4, 2, -2, -3, 9, 6, 2, 6, 0, -5, 1, -5, 0, 0, 0, 0}), &inv));
    M4 c = mat({-5, 2, 6, -8, 1, -5, 1, 8, 7, 7, -6, -7, 1, -3, 7, 4});
    CHECK(inverse(c, &inv));
    CHECK(determinant(c) == 532.0);
    CHECK(inv.m[3][2] == -160.0 / 532.0);
    CHECK(inv.m[2][3] == 105.0 / 532.0);
    CHECK(m4_eq(inv, mat({0.21805, 0.45113, 0.24060, -0.04511, -0.80827, -1.45677, -0.44361, 0.52068, -0.07895,
                          -0.22368, -0.05263, 0.19737, -0.52256, -0.81391, -0.30075, 0.30639})));
    CHECK(inverse(mat({8, -5, 9, 2, 7, 5, 6, 1, -6, 0, 9, 6, -3, 0, -9, -4}), &inv));
    CHECK(m4_eq(inv, mat({-0.15385, -0.15385, -0.28205, -0.53846, -0.07692, 0.12308, 0.02564, 0.03077, 0.35897,
                          0.35897, 0.43590, 0.92308, -0.69231, -0.69231, -0.76923, -1.92308})));
    CHECK(inverse(mat({9, 3, 0, 9, -5, -2, -6, -3, -4, 9, 6, 4, -7, 6, 6, 2}), &inv));
    CHECK(m4_eq(inv, mat({-0.04074, -0.07778, 0.14444, -0.22222, -0.07778, 0.03333, 0.36667, -0.33333, -0.02901,
                          -0.14630, -0.10926, 0.12963, 0.17778, 0.06667, -0.26667, 0.33333})));
    M4 a2 = mat({3, -9, 7, -3, 3, -8, 2, -9, -4, 4, 4, 1, -6, 5, -1, 1});
    M4 b2 = mat({8, 2, 2, 2, 3, -1, 7, 0, 7, 0, 5, 4, 6, -2, 0, 5});
    CHECK(inverse(b2, &inv));
    CHECK(m4_eq(mul(mul(a2, b2), inv), a2));
    CHECK(inverse(identity(), &inv));
    CHECK(m4_eq(inv, identity()));
    // transforms
    CHECK_T(mul(translation(5, -3, 2), pt(-3, 4, 5)), pt(2, 1, 7));
    CHECK(inverse(translation(5, -3, 2), &inv));
    CHECK_T(mul(inv, pt(-3, 4, 5)), pt(-8, 7, 3));
    CHECK_T(mul(translation(5, -3, 2), vec(-3, 4, 5)), vec(-3, 4, 5));
    CHECK_T(mul(scaling(2, 3, 4), pt(-4, 6, 8)), pt(-8, 18, 32));
    CHECK(inverse(scaling(2, 3, 4), &inv));
    CHECK_T(mul(inv, vec(-4, 6, 8)), vec(-2, 2, 2));
    double h = std::sqrt(2.0) / 2;
    CHECK_T(mul(rotation_x(PI / 4), pt(0, 1, 0)), pt(0, h, h));
    CHECK_T(mul(rotation_x(PI / 2), pt(0, 1, 0)), pt(0, 0, 1));
    CHECK_T(mul(rotation_y(PI / 4), pt(0, 0, 1)), pt(h, 0, h));
    CHECK_T(mul(rotation_y(PI / 2), pt(0, 0, 1)), pt(1, 0, 0));
    CHECK_T(mul(rotation_z(PI / 4), pt(0, 1, 0)), pt(-h, h, 0));
    CHECK_T(mul(rotation_z(PI / 2), pt(0, 1, 0)), pt(-1, 0, 0));
    CHECK_T(mul(shearing(1, 0, 0, 0, 0, 0), pt(2, 3, 4)), pt(5, 3, 4));
    CHECK_T(mul(shearing(0, 0, 0, 0, 0, 1), pt(2, 3, 4)), pt(2, 3, 7));
    M4 t = mul(translation(10, 5, 7), mul(scaling(5, 5, 5), rotation_x(PI / 2)));
    CHECK_T(mul(t, pt(1, 0, 1)), pt(15, 0, 7));
    // view transform, src/scene.rs:369-405
    CHECK(m4_eq(view_transform(pt(0, 0, 0), pt(0, 0, -1), vec(0, 1, 0)), identity()));
    CHECK(m4_eq(view_transform(pt(0, 0, 0), pt(0, 0, 1), vec(0, 1, 0)), scaling(-1, 1, -1)));
    CHECK(m4_eq(view_transform(pt(0, 0, 8), pt(0, 0, 0), vec(0, 1, 0)), translation(0, 0, -8)));
    CHECK(m4_eq(view_transform(pt(1, 3, 2), pt(4, -2, 8), vec(1, 1, 0)),
                mat({-0.50709, 0.50709, 0.67612, -2.36643, 0.76772, 0.60609, 0.12122, -2.82843, -0.35857, 0.59761,
                     -0.71714, 0.0, 0, 0, 0, 1})));
}

static void test_ray() {  // src/ray.rs:65-89
    Ray r{pt(2, 3, 4), vec(1, 0, 0)};
    CHECK_T(ray_position(r, 2.5), pt(4.5, 3, 4));
    CHECK_T(ray_position(r, -1.0), pt(1, 3, 4));
    Ray q{pt(1, 2, 3), vec(0, 1, 0)};
    Ray r2 = ray_transform(q, translation(3, 4, 5));
    CHECK_T(r2.origin, pt(4, 6, 8));
    CHECK_T(r2.direction, vec(0, 1, 0));
    r2 = ray_transform(q, scaling(2, 3, 4));
    CHECK_T(r2.origin, pt(2, 6, 12));
    CHECK_T(r2.direction, vec(0, 3, 0));
}

static void test_sphere_plane() {  // src/shapes/sphere.rs:110-200, src/shapes/plane.rs:85-123
    Ctx c;
    auto s = make_shape(SPHERE);
    Xs xs = intersect(c, *s, {pt(0, 0, -5), vec(0, 0, 1)});
    CHECK(xs.size() == 2 && xs[0].t == 4.0 && xs[1].t == 6.0);
    xs = intersect(c, *s, {pt(0, 1, -5), vec(0, 0, 1)});
    CHECK(xs.size() == 2 && xs[0].t == 5.0 && xs[1].t == 5.0);
    xs = intersect(c, *s, {pt(0, 2, -5), vec(0, 0, 1)});
    CHECK(xs.empty());
    xs = intersect(c, *s, {pt(0, 0, 0), vec(0, 0, 1)});
    CHECK(xs.size() == 2 && xs[0].t == -1.0 && xs[1].t == 1.0);
    xs = intersect(c, *s, {pt(0, 0, 5), vec(0, 0, 1)});
    CHECK(xs.size() == 2 && xs[0].t == -6.0 && xs[1].t == -4.0);
    s->transform = scaling(2, 2, 2);
    xs = intersect(c, *s, {pt(0, 0, -5), vec(0, 0, 1)});
    CHECK(xs.size() == 2 && xs[0].t == 3.0 && xs[1].t == 7.0);
    s->transform = translation(5, 0, 0);
    CHECK(intersect(c, *s, {pt(0, 0, -5), vec(0, 0, 1)}).empty());
    s->transform = identity();
    Intersection i0 = isect(0, s.get());
    CHECK_T(normal_at(c, *s, pt(1, 0, 0), i0), vec(1, 0, 0));
    double q = std::sqrt(3.0) / 3;
    Tuple n = normal_at(c, *s, pt(q, q, q), i0);
    CHECK_T(n, vec(q, q, q));
    CHECK_T(n, normalize(n));
    // src/shape.rs:235-249 (normal on a translated / transformed shape; a sphere's local normal
    // equals the TestShape's v(x,y,z))
    s->transform = translation(0, 1, 0);
    CHECK_T(normal_at(c, *s, pt(0, 1.70711, -0.70711), i0), vec(0, 0.70711, -0.70711));
    s->transform = mul(scaling(1, 0.5, 1), rotation_z(PI / 5));
    CHECK_T(normal_at(c, *s, pt(0, std::sqrt(2.0) / 2, -std::sqrt(2.0) / 2), i0), vec(0, 0.97014, -0.24254));

    auto p = make_shape(PLANE);
    CHECK(local_intersect(c, *p, {pt(0, 10, 0), vec(0, 0, 1)}).empty());
    CHECK(local_intersect(c, *p, {pt(0, 0, 0), vec(0, 0, 1)}).empty());
    xs = local_intersect(c, *p, {pt(0, 1, 0), vec(0, -1, 0)});
    CHECK(xs.size() == 1 && xs[0].t == 1.0 && xs[0].object == p.get());
    xs = local_intersect(c, *p, {pt(0, -1, 0), vec(0, 1, 0)});
    CHECK(xs.size() == 1 && xs[0].t == 1.0);
    CHECK_T(normal_at(c, *p, pt(10, 0, -10), isect(0, p.get())), vec(0, 1, 0));
}

static void test_cube() {  // src/shapes/cube.rs:124-181
    Ctx c;
    auto cube = make_shape(CUBE);
    struct {
        Tuple o, d;
        double t1, t2;
    } hits[] = {{pt(5, 0.5, 0), vec(-1, 0, 0), 4, 6}, {pt(-5, 0.5, 0), vec(1, 0, 0), 4, 6},
                {pt(0.5, 5, 0), vec(0, -1, 0), 4, 6}, {pt(0.5, -5, 0), vec(0, 1, 0), 4, 6},
                {pt(0.5, 0, 5), vec(0, 0, -1), 4, 6}, {pt(0.5, 0, -5), vec(0, 0, 1), 4, 6},
                {pt(0, 0.5, 0), vec(0, 0, 1), -1, 1}};
    for (auto& h : hits) {
        Xs xs = local_intersect(c, *cube, {h.o, h.d});
        CHECK(xs.size() == 2 && xs[0].t == h.t1 && xs[1].t == h.t2);
    }
    Ray misses[] = {{pt(-2, 0, 0), vec(0.2673, 0.5345, 0.8018)}, {pt(0, -2, 0), vec(0.8018, 0.2673, 0.5345)},
                    {pt(0, 0, -2), vec(0.5345, 0.8018, 0.2673)}, {pt(2, 0, 2), vec(0, 0, -1)},
                    {pt(0, 2, 2), vec(0, -1, 0)},                {pt(2, 0, 2), vec(-1, 0, 0)}};
    for (auto& r : misses) CHECK(local_intersect(c, *cube, r).empty());
    Intersection i0 = isect(0, cube.get());
    CHECK_T(normal_at(c, *cube, pt(1, 0.5, -0.8), i0), vec(1, 0, 0));
    CHECK_T(normal_at(c, *cube, pt(-1, -0.2, 0.9), i0), vec(-1, 0, 0));
    CHECK_T(normal_at(c, *cube, pt(-0.4, 1, -0.1), i0), vec(0, 1, 0));
    CHECK_T(normal_at(c, *cube, pt(0.3, -1, -0.7), i0), vec(0, -1, 0));
    CHECK_T(normal_at(c, *cube, pt(0.6, 0.3, 1), i0), vec(0, 0, 1));
    CHECK_T(normal_at(c, *cube, pt(0.4, 0.4, -1), i0), vec(0, 0, -1));
    CHECK_T(normal_at(c, *cube, pt(1, 1, 1), i0), vec(1, 0, 0));
    CHECK_T(normal_at(c, *cube, pt(-1, -1, -1), i0), vec(-1, 0, 0));
}

static void test_cylinder_cone() {  // src/shapes/cylinder.rs:164-269, src/shapes/cone.rs:174-240
    Ctx c;
    auto cyl = make_cyl(CYLINDER, -INF, INF, false);
    CHECK(local_intersect(c, *cyl, {pt(1, 0, 0), normalize(vec(0, 1, 0))}).empty());
    CHECK(local_intersect(c, *cyl, {pt(0, 0, 0), normalize(vec(0, 1, 0))}).empty());
    CHECK(local_intersect(c, *cyl, {pt(0, 0, -5), normalize(vec(1, 1, 1))}).empty());
    Xs xs = local_intersect(c, *cyl, {pt(1, 0, -5), normalize(vec(0, 0, 1))});
    CHECK(xs.size() == 2 && near(xs[0].t, 5) && near(xs[1].t, 5));
    xs = local_intersect(c, *cyl, {pt(0, 0, -5), normalize(vec(0, 0, 1))});
    CHECK(xs.size() == 2 && near(xs[0].t, 4) && near(xs[1].t, 6));
    xs = local_intersect(c, *cyl, {pt(0.5, 0, -5), normalize(vec(0.1, 1, 1))});
    CHECK(xs.size() == 2 && near(xs[0].t, 6.80798) && near(xs[1].t, 7.08872));
    auto cc = make_cyl(CYLINDER, 1, 2, false);
    struct {
        Tuple o, d;
        size_t n;
    } trunc[] = {{pt(0, 1.5, 0), vec(0.1, 1, 0), 0}, {pt(0, 3, -5), vec(0, 0, 1), 0}, {pt(0, 0, -5), vec(0, 0, 1), 0},
                 {pt(0, 2, -5), vec(0, 0, 1), 0},    {pt(0, 1, -5), vec(0, 0, 1), 0}, {pt(0, 1.5, -2), vec(0, 0, 1), 2}};
    for (auto& k : trunc) CHECK(local_intersect(c, *cc, {k.o, normalize(k.d)}).size() == k.n);
    auto cl = make_cyl(CYLINDER, 1, 2, true);
    struct {
        Tuple o, d;
        size_t n;
    } caps[] = {{pt(0, 3, 0), vec(0, -1, 0), 2},
                {pt(0, 3, -2), vec(0, -1, 2), 2},
                {pt(0, 4, -2), vec(0, -1, 1), 2},
                {pt(0, 0, -2), vec(0, 1, 2), 2},
                {pt(0, -1, -2), vec(0, 1, 1), 2}};
    for (auto& k : caps) CHECK(intersect(c, *cl, {k.o, normalize(k.d)}).size() == k.n);
    Intersection i0 = isect(0, cyl.get());
    CHECK_T(local_normal_at(*cyl, pt(1, 0, 0), i0), vec(1, 0, 0));
    CHECK_T(local_normal_at(*cyl, pt(0, 5, -1), i0), vec(0, 0, -1));
    CHECK_T(local_normal_at(*cyl, pt(0, -2, 1), i0), vec(0, 0, 1));
    CHECK_T(local_normal_at(*cyl, pt(-1, 1, 0), i0), vec(-1, 0, 0));
    CHECK_T(local_normal_at(*cl, pt(0, 1, 0), i0), vec(0, -1, 0));
    CHECK_T(local_normal_at(*cl, pt(0.5, 1, 0), i0), vec(0, -1, 0));
    CHECK_T(local_normal_at(*cl, pt(0, 1, 0.5), i0), vec(0, -1, 0));
    CHECK_T(local_normal_at(*cl, pt(0, 2, 0), i0), vec(0, 1, 0));
    CHECK_T(local_normal_at(*cl, pt(0.5, 2, 0), i0), vec(0, 1, 0));
    CHECK_T(local_normal_at(*cl, pt(0, 2, 0.5), i0), vec(0, 1, 0));

    auto cone = make_cyl(CONE, -INF, INF, false);
    xs = local_intersect(c, *cone, {pt(0, 0, -5), normalize(vec(0, 0, 1))});
    CHECK(xs.size() == 2 && near(xs[0].t, 5) && near(xs[1].t, 5));
    xs = local_intersect(c, *cone, {pt(0, 0, -5), normalize(vec(1, 1, 1))});
    CHECK(xs.size() == 2 && near(xs[0].t, 8.66025) && near(xs[1].t, 8.66025));
    xs = local_intersect(c, *cone, {pt(1, 1, -5), normalize(vec(-0.5, -1, 1))});
    CHECK(xs.size() == 2 && near(xs[0].t, 4.55006) && near(xs[1].t, 49.44994));
    xs = local_intersect(c, *cone, {pt(0, 0, -1), normalize(vec(0, 1, 1))});
    CHECK(xs.size() == 1 && near(xs[0].t, 0.35355));
    auto ccone = make_cyl(CONE, -0.5, 0.5, true);
    CHECK(intersect(c, *ccone, {pt(0, 0, -5), normalize(vec(0, 1, 0))}).size() == 0);
    CHECK(intersect(c, *ccone, {pt(0, 0, -0.25), normalize(vec(0, 1, 1))}).size() == 2);
    CHECK(intersect(c, *ccone, {pt(0, 0, -0.25), normalize(vec(0, 1, 0))}).size() == 4);
    CHECK_T(local_normal_at(*cone, pt(0, 0, 0), i0), vec(0, 0, 0));
    CHECK_T(local_normal_at(*cone, pt(1, 1, 1), i0), vec(1, -std::sqrt(2.0), 1));
    CHECK_T(local_normal_at(*cone, pt(-1, -1, 0), i0), vec(-1, 1, 0));
}

static void test_triangle() {  // src/shapes/triangle.rs:135-244
    Ctx c;
    auto t = make_triangle(pt(0, 1, 0), pt(-1, 0, 0), pt(1, 0, 0));
    CHECK_T(t->e1, vec(-1, -1, 0));
    CHECK_T(t->e2, vec(1, -1, 0));
    CHECK_T(t->normal, vec(0, 0, -1));
    CHECK(local_intersect(c, *t, {pt(0, -1, -2), vec(0, 1, 0)}).empty());
    CHECK(local_intersect(c, *t, {pt(1, 1, -2), vec(0, 0, 1)}).empty());
    CHECK(local_intersect(c, *t, {pt(-1, 1, -2), vec(0, 0, 1)}).empty());
    CHECK(local_intersect(c, *t, {pt(0, -1, -2), vec(0, 0, 1)}).empty());
    Xs xs = local_intersect(c, *t, {pt(0, 0.5, -2), vec(0, 0, 1)});
    CHECK(xs.size() == 1 && xs[0].t == 2.0);
    auto tri = make_smooth_triangle(pt(0, 1, 0), pt(-1, 0, 0), pt(1, 0, 0), vec(0, 1, 0), vec(-1, 0, 0), vec(1, 0, 0));
    Ray r{pt(-0.2, 0.3, -2), vec(0, 0, 1)};
    xs = local_intersect(c, *tri, r);
    CHECK(xs.size() == 1 && xs[0].has_uv && near(xs[0].u, 0.45) && near(xs[0].v, 0.25));
    Intersection i = isect_uv(1.0, tri.get(), 0.45, 0.25);
    CHECK_T(normal_at(c, *tri, pt(0, 0, 0), i), vec(-0.5547, 0.83205, 0));
    Xs one{i};
    Comps k = prepare_computations(c, i, r, one);
    CHECK_T(k.normalv, vec(-0.5547, 0.83205, 0));
}

static Shape* child(Shape* g, int i) { return g->children[i].get(); }

static void test_group() {  // src/shapes/group.rs:139-233
    Ctx c;
    auto g = make_shape(GROUP);
    CHECK(local_intersect(c, *g, {pt(0, 0, 0), vec(0, 0, 1)}).empty());
    auto s1 = make_shape(SPHERE);
    auto s2 = make_shape(SPHERE);
    s2->transform = translation(0, 0, -3);
    auto s3 = make_shape(SPHERE);
    s3->transform = translation(5, 0, 0);
    g->children.push_back(std::move(s1));
    g->children.push_back(std::move(s2));
    g->children.push_back(std::move(s3));
    Xs xs = local_intersect(c, *g, {pt(0, 0, -5), vec(0, 0, 1)});
    CHECK(xs.size() == 4);
    CHECK(xs[0].object == child(g.get(), 1) && xs[1].object == child(g.get(), 1));
    CHECK(xs[2].object == child(g.get(), 0) && xs[3].object == child(g.get(), 0));
    auto g2 = make_shape(GROUP);
    g2->transform = scaling(2, 2, 2);
    auto s = make_shape(SPHERE);
    s->transform = translation(5, 0, 0);
    g2->children.push_back(std::move(s));
    CHECK(intersect(c, *g2, {pt(10, 0, -10), vec(0, 0, 1)}).size() == 2);
    // world_to_object / normal_to_world / normal on a child object
    auto outer = make_shape(GROUP);
    outer->transform = rotation_y(PI / 2);
    auto inner = make_shape(GROUP);
    inner->transform = scaling(2, 2, 2);
    auto sp = make_shape(SPHERE);
    sp->transform = translation(5, 0, 0);
    inner->children.push_back(std::move(sp));
    outer->children.push_back(std::move(inner));
    refresh_parents(*outer);
    Shape* leaf = child(child(outer.get(), 0), 0);
    CHECK(leaf->parent_transforms.size() == 2);
    CHECK_T(world_to_object(c, *leaf, pt(-2, 0, -10)), pt(0, 0, -1));
    child(outer.get(), 0)->transform = scaling(1, 2, 3);
    refresh_parents(*outer);
    double q = std::sqrt(3.0) / 3;
    CHECK_T(normal_to_world(c, *leaf, vec(q, q, q)), vec(0.28571, 0.42857, -0.85714));
    CHECK_T(normal_at(c, *leaf, pt(1.7321, 1.1547, -5.5774), isect(0, leaf)), vec(0.2857, 0.42854, -0.85716));
}

static void test_csg() {  // src/shapes/csg.rs:147-227
    Ctx c;
    struct {
        CsgOp op;
        bool left, inl, inr, res;
    } tt[] = {{OP_UNION, true, true, true, false},
              {OP_UNION, true, true, false, true},
              {OP_UNION, true, false, true, false},
              {OP_UNION, true, false, false, true},
              {OP_UNION, false, true, true, false},
              {OP_UNION, false, true, false, false},
              {OP_UNION, false, false, true, true},
              {OP_UNION, false, false, false, true},
              {OP_INTERSECTION, true, true, true, true},
              {OP_INTERSECTION, true, true, false, false},
              {OP_INTERSECTION, true, false, true, true},
              {OP_INTERSECTION, true, false, false, false},
              {OP_INTERSECTION, false, true, true, true},
              {OP_INTERSECTION, false, true, false, true},
              {OP_INTERSECTION, false, false, true, false},
              {OP_INTERSECTION, false, false, false, false},
              {OP_DIFFERENCE, true, true, true, false},
              {OP_DIFFERENCE, true, true, false, true},
              {OP_DIFFERENCE, true, false, true, false},
              {OP_DIFFERENCE, true, false, false, true},
              {OP_DIFFERENCE, false, true, true, true},
              {OP_DIFFERENCE, false, true, false, true},
              {OP_DIFFERENCE, false, false, true, false},
              {OP_DIFFERENCE, false, false, false, false}};
    for (auto& k : tt) CHECK(csg_intersection_allowed(k.op, k.left, k.inl, k.inr) == k.res);
    struct {
        CsgOp op;
        int x0, x1;
    } fl[] = {{OP_UNION, 0, 3}, {OP_INTERSECTION, 1, 2}, {OP_DIFFERENCE, 0, 1}};
    for (auto& k : fl) {
        auto csg = make_csg(k.op, make_shape(SPHERE), make_shape(CUBE));
        Xs xs{isect(1, csg->left.get()), isect(2, csg->right.get()), isect(3, csg->left.get()),
              isect(4, csg->right.get())};
        Xs res = csg_filter(*csg, xs);
        CHECK(res.size() == 2 && res[0].t == xs[k.x0].t && res[1].t == xs[k.x1].t);
    }
    auto u = make_csg(OP_UNION, make_shape(SPHERE), make_shape(CUBE));
    CHECK(local_intersect(c, *u, {pt(0, 2, -5), vec(0, 0, 1)}).empty());
    auto s2 = make_shape(SPHERE);
    s2->transform = mul(translation(0, 0, 0.5), identity());
    auto u2 = make_csg(OP_UNION, make_shape(SPHERE), std::move(s2));
    Xs xs = local_intersect(c, *u2, {pt(0, 0, -5), vec(0, 0, 1)});
    CHECK(xs.size() == 2 && xs[0].t == 4.0 && xs[0].object == u2->left.get() && xs[1].t == 6.5 &&
          xs[1].object == u2->right.get());
}

static void test_intersection() {  // src/intersection.rs:155-345
    Ctx c;
    auto s = make_shape(SPHERE);
    {
        Xs xs{isect(5, s.get()), isect(7, s.get()), isect(-3, s.get()), isect(2, s.get())};
        std::stable_sort(xs.begin(), xs.end(), [](const Intersection& a, const Intersection& b) { return a.t < b.t; });
        CHECK(hit(xs) && hit(xs)->t == 2.0);
        Xs neg{isect(-2, s.get()), isect(-1, s.get())};
        CHECK(hit(neg) == nullptr);
    }
    Ray r{pt(0, 0, -5), vec(0, 0, 1)};
    Intersection i = isect(4, s.get());
    Comps k = prepare_computations(c, i, r, Xs{i});
    CHECK(k.t == 4.0);
    CHECK_T(k.point, pt(0, 0, -1));
    CHECK_T(k.eyev, vec(0, 0, -1));
    CHECK_T(k.normalv, vec(0, 0, -1));
    CHECK(!k.inside);
    Ray r2{pt(0, 0, 0), vec(0, 0, 1)};
    i = isect(1, s.get());
    k = prepare_computations(c, i, r2, Xs{i});
    CHECK_T(k.point, pt(0, 0, 1));
    CHECK(k.inside);
    CHECK_T(k.normalv, vec(0, 0, -1));
    auto p = make_shape(PLANE);
    double h = std::sqrt(2.0) / 2;
    i = isect(std::sqrt(2.0), p.get());
    k = prepare_computations(c, i, {pt(0, 1, -1), vec(0, -h, h)}, Xs{i});
    CHECK_T(k.reflectv, vec(0, h, h));
    // n1/n2, :249-288
    auto a = glass_sphere();
    a->transform = scaling(2, 2, 2);
    auto b = glass_sphere();
    b->transform = translation(0, 0, -0.25);
    b->material.refractive_index = 2.0;
    auto cc = glass_sphere();
    cc->transform = translation(0, 0, 0.25);
    cc->material.refractive_index = 2.5;
    Ray r3{pt(0, 0, -4), vec(0, 0, 1)};
    Xs xs{isect(2, a.get()),    isect(2.75, b.get()), isect(3.25, cc.get()),
          isect(4.75, b.get()), isect(5.25, cc.get()), isect(6, a.get())};
    double exp[6][2] = {{1.0, 1.5}, {1.5, 2.0}, {2.0, 2.5}, {2.5, 2.5}, {2.5, 1.5}, {1.5, 1.0}};
    for (int q = 0; q < 6; q++) {
        k = prepare_computations(c, xs[q], r3, xs);
        CHECK(k.n1 == exp[q][0] && k.n2 == exp[q][1]);
    }
    // under point, :291-300
    auto g = glass_sphere();
    g->transform = translation(0, 0, 1);
    i = isect(5, g.get());
    k = prepare_computations(c, i, r, Xs{i});
    CHECK(k.under_point.z > EPSILON / 2 && k.point.z < k.under_point.z);
    // over point, src/scene.rs:580-591
    auto o = make_shape(SPHERE);
    o->transform = translation(0, 0, 1);
    i = isect(5, o.get());
    k = prepare_computations(c, i, r, Xs{i});
    CHECK(k.over_point.z < -EPSILON / 2 && k.point.z > k.over_point.z);
    // schlick, :303-345
    auto gs = glass_sphere();
    Xs x2{isect(-h, gs.get()), isect(h, gs.get())};
    k = prepare_computations(c, x2[1], {pt(0, 0, h), vec(0, 1, 0)}, x2);
    CHECK(schlick(k) == 1.0);
    Xs x3{isect(-1, gs.get()), isect(1, gs.get())};
    k = prepare_computations(c, x3[1], {pt(0, 0, 0), vec(0, 1, 0)}, x3);
    CHECK(near(schlick(k), 0.04));
    Xs x4{isect(1.8589, gs.get())};
    k = prepare_computations(c, x4[0], {pt(0, 0.99, -2), vec(0, 0, 1)}, x4);
    CHECK(near(schlick(k), 0.48873));
}

static void test_material_pattern() {  // src/material.rs:159-285, src/pattern.rs:112-232
    Ctx c;
    Material m;
    auto s = make_shape(SPHERE);
    Tuple pos = pt(0, 0, 0);
    double h = std::sqrt(2.0) / 2;
    PointLight l{pt(0, 0, -10), {1, 1, 1}};
    CHECK_C(lighting(c, m, *s, l, pos, vec(0, 0, -1), vec(0, 0, -1), false), (Color{1.9, 1.9, 1.9}));
    CHECK_C(lighting(c, m, *s, l, pos, vec(0, 0, -1), vec(0, 0, -1), true), (Color{0.1, 0.1, 0.1}));
    CHECK_C(lighting(c, m, *s, l, pos, vec(0, h, h), vec(0, 0, -1), false), (Color{1.0, 1.0, 1.0}));
    PointLight l2{pt(0, 10, -10), {1, 1, 1}};
    CHECK_C(lighting(c, m, *s, l2, pos, vec(0, 0, -1), vec(0, 0, -1), false), (Color{0.7364, 0.7364, 0.7364}));
    CHECK_C(lighting(c, m, *s, l2, pos, vec(0, -h, -h), vec(0, 0, -1), false), (Color{1.6364, 1.6364, 1.6364}));
    PointLight l3{pt(0, 0, 10), {1, 1, 1}};
    CHECK_C(lighting(c, m, *s, l3, pos, vec(0, 0, -1), vec(0, 0, -1), false), (Color{0.1, 0.1, 0.1}));
    Material pm;
    pm.pattern.kind = PAT_STRIPE;
    pm.pattern.a = {1, 1, 1};
    pm.pattern.b = {0, 0, 0};
    pm.ambient = 1, pm.diffuse = 0, pm.specular = 0;
    CHECK_C(lighting(c, pm, *s, l, pt(0.9, 0, 0), vec(0, 0, -1), vec(0, 0, -1), false), (Color{1, 1, 1}));
    CHECK_C(lighting(c, pm, *s, l, pt(1.1, 0, 0), vec(0, 0, -1), vec(0, 0, -1), false), (Color{0, 0, 0}));
    Color W{1, 1, 1}, B{0, 0, 0};
    Pattern st;
    st.kind = PAT_STRIPE, st.a = W, st.b = B;
    CHECK_C(pattern_at(st, pt(0, 2, 0)), W);
    CHECK_C(pattern_at(st, pt(0, 0, 2)), W);
    CHECK_C(pattern_at(st, pt(0.9, 0, 0)), W);
    CHECK_C(pattern_at(st, pt(1.0, 0, 0)), B);
    CHECK_C(pattern_at(st, pt(-0.1, 0, 0)), B);
    CHECK_C(pattern_at(st, pt(-1.0, 0, 0)), B);
    CHECK_C(pattern_at(st, pt(-1.1, 0, 0)), W);
    Pattern tp;
    tp.kind = PAT_TEST;
    auto sc = make_shape(SPHERE);
    sc->transform = scaling(2, 2, 2);
    CHECK_C(pattern_at_object(c, tp, *sc, pt(2, 3, 4)), (Color{1, 1.5, 2}));
    tp.transform = translation(0.5, 1, 1.5);
    CHECK_C(pattern_at_object(c, tp, *sc, pt(2.5, 3, 3.5)), (Color{0.75, 0.5, 0.25}));
    CHECK_C(pattern_at_object(c, st, *sc, pt(1.5, 0, 0)), W);
    Pattern st2 = st;
    st2.transform = scaling(2, 2, 2);
    CHECK_C(pattern_at_object(c, st2, *s, pt(1.5, 0, 0)), W);
    CHECK_C(pattern_at_object(c, st2, *sc, pt(2.5, 0, 0)), W);
    Pattern gr;
    gr.kind = PAT_GRADIENT, gr.a = W, gr.b = B;
    CHECK_C(pattern_at(gr, pt(0.25, 0, 0)), (Color{0.75, 0.75, 0.75}));
    CHECK_C(pattern_at(gr, pt(0.75, 0, 0)), (Color{0.25, 0.25, 0.25}));
    Pattern rg;
    rg.kind = PAT_RING, rg.a = W, rg.b = B;
    CHECK_C(pattern_at(rg, pt(0, 0, 0)), W);
    CHECK_C(pattern_at(rg, pt(1, 0, 0)), B);
    CHECK_C(pattern_at(rg, pt(0, 0, 1)), B);
    CHECK_C(pattern_at(rg, pt(0.708, 0, 0.708)), B);
    Pattern ch;
    ch.kind = PAT_CHECKERS, ch.a = W, ch.b = B;
    CHECK_C(pattern_at(ch, pt(0.99, 0, 0)), W);
    CHECK_C(pattern_at(ch, pt(1.01, 0, 0)), B);
    CHECK_C(pattern_at(ch, pt(0, 0.99, 0)), W);
    CHECK_C(pattern_at(ch, pt(0, 1.01, 0)), B);
    CHECK_C(pattern_at(ch, pt(0, 0, 0.99)), W);
    CHECK_C(pattern_at(ch, pt(0, 0, 1.01)), B);
}

static std::unique_ptr<Shape> plane_at(double y, const Material& m) {
    auto p = make_shape(PLANE);
    p->transform = translation(0, y, 0);
    p->material = m;
    return p;
}

static void test_world() {  // src/scene.rs:287-708
    Ctx c;
    double h = std::sqrt(2.0) / 2, s2 = std::sqrt(2.0);
    {
        World w = default_world();
        Xs xs = world_intersect(c, w, {pt(0, 0, -5), vec(0, 0, 1)});
        CHECK(xs.size() == 4 && xs[0].t == 4.0 && xs[1].t == 4.5 && xs[2].t == 5.5 && xs[3].t == 6.0);
        Ray r{pt(0, 0, -5), vec(0, 0, 1)};
        Intersection i = isect(4, w.objects[0].get());
        CHECK_C(shade_hit(c, w, prepare_computations(c, i, r, Xs{i}), 1), (Color{0.38066, 0.47583, 0.2855}));
        w.lights[0] = {pt(0, 0.25, 0), {1, 1, 1}};
        Ray r2{pt(0, 0, 0), vec(0, 0, 1)};
        i = isect(0.5, w.objects[1].get());
        CHECK_C(shade_hit(c, w, prepare_computations(c, i, r2, Xs{i}), 1), (Color{0.90498, 0.90498, 0.90498}));
    }
    {
        World w = default_world();
        CHECK_C(color_at(c, w, {pt(0, 0, -5), vec(0, 1, 0)}, 1), (Color{0, 0, 0}));
        CHECK_C(color_at(c, w, {pt(0, 0, -5), vec(0, 0, 1)}, 1), (Color{0.38066, 0.47583, 0.2855}));
        w.objects[0]->material.ambient = 1.0;
        w.objects[1]->material.ambient = 1.0;
        CHECK_C(color_at(c, w, {pt(0, 0, 0.75), vec(0, 0, -1)}, 1), w.objects[1]->material.color);
        // mutually reflective planes terminate
        World m;
        m.lights.push_back({pt(0, 0, 0), {1, 1, 1}});
        Material mm;
        mm.reflective = 1.0;
        m.objects.push_back(plane_at(-1, mm));
        m.objects.push_back(plane_at(1, mm));
        color_at(c, m, {pt(0, 0, 0), vec(0, 1, 0)}, 1);
        CHECK(true);
    }
    {  // camera, :414-468
        Camera cam = make_camera(200, 125, PI / 2);
        CHECK(near(cam.pixel_size, 0.01));
        cam = make_camera(125, 200, PI / 2);
        CHECK(near(cam.pixel_size, 0.01));
        cam = make_camera(201, 101, PI / 2);
        Ray r = ray_for_pixel(c, cam, 100, 50);
        CHECK_T(r.origin, pt(0, 0, 0));
        CHECK_T(r.direction, vec(0, 0, -1));
        r = ray_for_pixel(c, cam, 0, 0);
        CHECK_T(r.direction, vec(0.66519, 0.33259, -0.66851));
        cam.transform = mul(rotation_y(PI / 4), translation(0, -2, 5));
        r = ray_for_pixel(c, cam, 100, 50);
        CHECK_T(r.origin, pt(0, 2, -5));
        CHECK_T(r.direction, vec(h, 0, -h));
        World w = default_world();
        Camera c11 = make_camera(11, 11, PI / 2);
        c11.transform = view_transform(pt(0, 0, -5), pt(0, 0, 0), vec(0, 1, 0));
        world_refresh_parents(w);
        CHECK_C(color_at(c, w, ray_for_pixel(c, c11, 5, 5), 4), (Color{0.38066, 0.47583, 0.2855}));
    }
    {  // shadows, :471-508
        World w = default_world();
        CHECK(!is_shadowed(c, w, w.lights[0], pt(0, 10, 0)));
        CHECK(is_shadowed(c, w, w.lights[0], pt(10, -10, 10)));
        CHECK(!is_shadowed(c, w, w.lights[0], pt(-20, 20, -20)));
        CHECK(!is_shadowed(c, w, w.lights[0], pt(-2, 2, -2)));
        World w2;
        w2.lights.push_back({pt(0, 0, -10), {1, 1, 1}});
        w2.objects.push_back(make_shape(SPHERE));
        auto sb = make_shape(SPHERE);
        sb->transform = translation(0, 0, 10);
        w2.objects.push_back(std::move(sb));
        Intersection i = isect(4, w2.objects[1].get());
        CHECK_C(shade_hit(c, w2, prepare_computations(c, i, {pt(0, 0, 5), vec(0, 0, 1)}, Xs{i}), 1),
                (Color{0.1, 0.1, 0.1}));
    }
    Ray rr{pt(0, 0, -3), vec(0, -h, h)};
    {  // reflective, :511-526 and :594-626
        World w = default_world();
        Material m;
        m.reflective = 0.5;
        auto shape = plane_at(-1, m);
        Intersection i = isect(s2, shape.get());
        Comps k = prepare_computations(c, i, rr, Xs{i});
        CHECK_C(shade_hit(c, w, k, 5), (Color{0.87675, 0.92434, 0.82917}));
        CHECK_C(reflected_color(c, w, k, 1), (Color{0.19033, 0.23791, 0.142749}));
        CHECK_C(reflected_color(c, w, k, 0), (Color{0, 0, 0}));
        World w3 = default_world();
        w3.objects[1]->material.ambient = 1.0;
        i = isect(1, w3.objects[1].get());
        CHECK_C(reflected_color(c, w3, prepare_computations(c, i, {pt(0, 0, 0), vec(0, 0, 1)}, Xs{i}), 1),
                (Color{0, 0, 0}));
    }
    {  // transparent, :529-551
        World w = default_world();
        Material m;
        m.transparency = 0.5;
        m.refractive_index = 1.5;
        w.objects.push_back(plane_at(-1, m));
        auto ball = make_shape(SPHERE);
        ball->material.color = {1, 0, 0};
        ball->material.ambient = 0.5;
        ball->transform = translation(0, -3.5, -0.5);
        w.objects.push_back(std::move(ball));
        Xs xs{isect(s2, w.objects[2].get())};
        CHECK_C(shade_hit(c, w, prepare_computations(c, xs[0], rr, xs), 5), (Color{0.93642, 0.68642, 0.68642}));
        // reflective + transparent, :554-577
        w.objects[2]->material.reflective = 0.5;
        Xs ys{isect(s2, w.objects[2].get())};
        CHECK_C(shade_hit(c, w, prepare_computations(c, ys[0], rr, ys), 5), (Color{0.93391, 0.69643, 0.69243}));
    }
    {  // refracted colour, :629-708
        World w = default_world();
        Ray r{pt(0, 0, -5), vec(0, 0, 1)};
        Xs xs{isect(4, w.objects[0].get()), isect(6, w.objects[0].get())};
        CHECK_C(refracted_color(c, w, prepare_computations(c, xs[0], r, xs), 5), (Color{0, 0, 0}));
        w.objects[0]->material.transparency = 1.0;
        w.objects[0]->material.refractive_index = 1.5;
        Xs ys{isect(4, w.objects[0].get()), isect(6, w.objects[0].get())};
        CHECK_C(refracted_color(c, w, prepare_computations(c, ys[0], r, ys), 0), (Color{0, 0, 0}));
        Ray ri{pt(0, 0, h), vec(0, 1, 0)};
        Xs zs{isect(-h, w.objects[0].get()), isect(h, w.objects[0].get())};
        CHECK_C(refracted_color(c, w, prepare_computations(c, zs[1], ri, zs), 5), (Color{0, 0, 0}));
        World w2 = default_world();
        w2.objects[0]->material.ambient = 1.0;
        w2.objects[0]->material.pattern.kind = PAT_TEST;
        w2.objects[1]->material.transparency = 1.0;
        w2.objects[1]->material.refractive_index = 1.5;
        Ray r4{pt(0, 0, 0.1), vec(0, 1, 0)};
        Xs qs{isect(-0.9899, w2.objects[0].get()), isect(-0.4899, w2.objects[1].get()),
              isect(0.4899, w2.objects[1].get()), isect(0.9899, w2.objects[0].get())};
        CHECK_C(refracted_color(c, w2, prepare_computations(c, qs[2], r4, qs), 5), (Color{0, 0.99888, 0.04721}));
    }
}

// faithful and cached evaluation must agree to the bit
static void test_modes_agree() {
    World w = default_world();
    Material m;
    m.reflective = 0.5, m.transparency = 0.5, m.refractive_index = 1.5;
    w.objects.push_back(plane_at(-1, m));
    w.lights.push_back({pt(5, 10, -3), {0.3, 0.3, 0.3}});
    Camera cam = make_camera(16, 12, 1.0);
    cam.transform = view_transform(pt(0, 1.5, -5), pt(0, 0, 0), vec(0, 1, 0));
    Ctx a, b;
    a.faithful = true;
    for (int y = 0; y < 12; y++)
        for (int x = 0; x < 16; x++) {
            Color ca = color_at(a, w, ray_for_pixel(a, cam, x, y), 4);
            Color cb = color_at(b, w, ray_for_pixel(b, cam, x, y), 4);
            CHECK(ca.r == cb.r && ca.g == cb.g && ca.b == cb.b);
        }
    CHECK(a.stats.rays > b.stats.rays);  // two lights => duplicated sub-trees in faithful mode
}

int main() {
    test_tuple();
    test_matrix();
    test_ray();
    test_sphere_plane();
    test_cube();
    test_cylinder_cone();
    test_triangle();
    test_group();
    test_csg();
    test_intersection();
    test_material_pattern();
    test_world();
    test_modes_agree();
    std::printf("kat: %d checks, %d failures\n", g_checks, g_fail);
    return g_fail ? 1 : 0;
}
