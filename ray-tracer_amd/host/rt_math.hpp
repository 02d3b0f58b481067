// Host-side value types of the scene API: Tuple, Color, Matrix4x4.
// They mirror the reference's public types (src/tuple.rs, src/matrix.rs) because the host
// flattens scenes with them: every inverse uploaded to the GPU is computed here with the
// reference's cofactor algorithm (src/matrix.rs:185-257) in plain f64 (this file must be
// compiled with -ffp-contract=off), so a cached matrix has the bits the reference would
// recompute at every call.
#pragma once
#include <cmath>

namespace rt {

constexpr double EPSILON = 0.00001;  // src/lib.rs:26
inline bool equal(double a, double b) { return std::fabs(a - b) < EPSILON; }  // src/lib.rs:28-30

struct Tuple {
    double x = 0, y = 0, z = 0, w = 0;
    static Tuple point(double x, double y, double z) { return {x, y, z, 1.0}; }
    static Tuple vector(double x, double y, double z) { return {x, y, z, 0.0}; }
    bool is_point() const { return w == 1.0; }
    bool is_vector() const { return w == 0.0; }
    double dot(const Tuple& o) const { return x * o.x + y * o.y + z * o.z + w * o.w; }
    double magnitude() const { return std::sqrt(x * x + y * y + z * z + w * w); }
    Tuple normalize() const {
        double m = magnitude();
        return {x / m, y / m, z / m, w / m};
    }
    bool operator==(const Tuple& o) const { return equal(x, o.x) && equal(y, o.y) && equal(z, o.z) && equal(w, o.w); }
};
inline Tuple pt(double x, double y, double z) { return Tuple::point(x, y, z); }
inline Tuple v(double x, double y, double z) { return Tuple::vector(x, y, z); }
inline Tuple operator+(Tuple a, Tuple b) { return {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
inline Tuple operator-(Tuple a, Tuple b) { return {a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w}; }
inline Tuple operator-(Tuple a) { return {-a.x, -a.y, -a.z, -a.w}; }
inline Tuple operator*(Tuple a, double s) { return {a.x * s, a.y * s, a.z * s, a.w * s}; }
inline Tuple operator/(Tuple a, double s) { return {a.x / s, a.y / s, a.z / s, a.w / s}; }
// cross product; keeps the left operand's w (src/tuple.rs:167-180)
inline Tuple cross(Tuple a, Tuple b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x, a.w}; }

struct Color {
    double red = 0, green = 0, blue = 0;
    bool operator==(const Color& o) const { return equal(red, o.red) && equal(green, o.green) && equal(blue, o.blue); }
};
inline Color black() { return {0, 0, 0}; }
inline Color white() { return {1, 1, 1}; }

struct Matrix4x4 {
    double m[4][4];
    static Matrix4x4 identity() {
        Matrix4x4 r{};
        for (int i = 0; i < 4; i++) r.m[i][i] = 1.0;
        return r;
    }
    static Matrix4x4 translation(double x, double y, double z) {
        Matrix4x4 r = identity();
        r.m[0][3] = x, r.m[1][3] = y, r.m[2][3] = z;
        return r;
    }
    static Matrix4x4 scaling(double x, double y, double z) {
        Matrix4x4 r = identity();
        r.m[0][0] = x, r.m[1][1] = y, r.m[2][2] = z;
        return r;
    }
    static Matrix4x4 rotation_x(double a) {
        Matrix4x4 r = identity();
        r.m[1][1] = std::cos(a), r.m[1][2] = -std::sin(a), r.m[2][1] = std::sin(a), r.m[2][2] = std::cos(a);
        return r;
    }
    static Matrix4x4 rotation_y(double a) {
        Matrix4x4 r = identity();
        r.m[0][0] = std::cos(a), r.m[0][2] = std::sin(a), r.m[2][0] = -std::sin(a), r.m[2][2] = std::cos(a);
        return r;
    }
    static Matrix4x4 rotation_z(double a) {
        Matrix4x4 r = identity();
        r.m[0][0] = std::cos(a), r.m[0][1] = -std::sin(a), r.m[1][0] = std::sin(a), r.m[1][1] = std::cos(a);
        return r;
    }
    static Matrix4x4 shearing(double xy, double xz, double yx, double yz, double zx, double zy) {
        Matrix4x4 r = identity();
        r.m[0][1] = xy, r.m[0][2] = xz, r.m[1][0] = yx, r.m[1][2] = yz, r.m[2][0] = zx, r.m[2][1] = zy;
        return r;
    }
    // fluent builders pre-multiply (src/matrix.rs:153-171)
    Matrix4x4 rotate_x(double a) const { return rotation_x(a) * *this; }
    Matrix4x4 rotate_y(double a) const { return rotation_y(a) * *this; }
    Matrix4x4 rotate_z(double a) const { return rotation_z(a) * *this; }
    Matrix4x4 scale(double x, double y, double z) const { return scaling(x, y, z) * *this; }
    Matrix4x4 translate(double x, double y, double z) const { return translation(x, y, z) * *this; }

    Matrix4x4 operator*(const Matrix4x4& b) const {
        Matrix4x4 r;
        for (int i = 0; i < 4; i++)
            for (int j = 0; j < 4; j++) r.m[i][j] = m[i][0] * b.m[0][j] + m[i][1] * b.m[1][j] + m[i][2] * b.m[2][j] + m[i][3] * b.m[3][j];
        return r;
    }
    Tuple operator*(const Tuple& t) const {
        double b[4] = {t.x, t.y, t.z, t.w}, o[4];
        for (int r = 0; r < 4; r++) o[r] = m[r][0] * b[0] + m[r][1] * b[1] + m[r][2] * b[2] + m[r][3] * b[3];
        return {o[0], o[1], o[2], o[3]};
    }
    Matrix4x4 transpose() const {
        Matrix4x4 r;
        for (int i = 0; i < 4; i++)
            for (int j = 0; j < 4; j++) r.m[i][j] = m[j][i];
        return r;
    }
    bool operator==(const Matrix4x4& o) const {
        for (int i = 0; i < 4; i++)
            for (int j = 0; j < 4; j++)
                if (!equal(m[i][j], o.m[i][j])) return false;
        return true;
    }
    // Laplace expansion along the first row at every level, term order as in the reference.
    static double det3(const double a[3][3]) {
        double d = 0.0;
        for (int c = 0; c < 3; c++) {
            int c0 = c == 0 ? 1 : 0, c1 = c == 2 ? 1 : 2;
            double minor = a[1][c0] * a[2][c1] - a[1][c1] * a[2][c0];
            d += a[0][c] * (minor * (c % 2 == 0 ? 1.0 : -1.0));
        }
        return d;
    }
    double cofactor(int row, int col) const {
        double s[3][3];
        for (int r = 0, sr = 0; r < 4; r++) {
            if (r == row) continue;
            for (int c = 0, sc = 0; c < 4; c++) {
                if (c == col) continue;
                s[sr][sc++] = m[r][c];
            }
            sr++;
        }
        return det3(s) * ((row + col) % 2 == 0 ? 1.0 : -1.0);
    }
    double determinant() const {
        double d = 0.0;
        for (int c = 0; c < 4; c++) d += m[0][c] * cofactor(0, c);
        return d;
    }
    bool is_invertible() const { return determinant() != 0.0; }
    // false when singular (the reference returns None and its callers panic)
    bool inverse(Matrix4x4* out) const {
        double d = determinant();
        if (d == 0.0) return false;
        for (int r = 0; r < 4; r++)
            for (int c = 0; c < 4; c++) out->m[c][r] = cofactor(r, c) / d;
        return true;
    }
};
inline Matrix4x4 id() { return Matrix4x4::identity(); }

}  // namespace rt
