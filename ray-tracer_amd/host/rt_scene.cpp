// Scene model, flattening and Camera::render of the host-side API (see rt_scene.hpp).
#include "rt_scene.hpp"

#include <zlib.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <limits>
#include <mutex>

namespace rt {

// Triangles per MESH leaf.  The device tests a leaf's triangles only for rays whose line meets the
// bounds of that leaf, so short runs (consecutive faces of an OBJ file lie close together) let most rays
// skip most of a mesh.  Must not exceed RTAMD_MAX_MESH_TRIS.
static constexpr int kMeshLeafTris = 64;  // most triangles in one leaf (a wave has 64 lanes: one triangle each when few rays come near)

// ------------------------------------------------------------------ patterns / materials
static bool color_bits_eq(const Color& a, const Color& b) { return std::memcmp(&a, &b, sizeof(Color)) == 0; }

bool Pattern::operator==(const Pattern& o) const {  // derived PartialEq, src/pattern.rs:3-16
    if (design != o.design) return false;
    if (design != PatternDesign::Test && !(a == o.a && b == o.b)) return false;
    return transform == o.transform;
}
static Pattern mk_pattern(PatternDesign d, Color a, Color b) {
    Pattern p;
    p.design = d, p.a = a, p.b = b;
    return p;
}
Pattern stripe_pattern(Color a, Color b) { return mk_pattern(PatternDesign::Stripe, a, b); }
Pattern stripe_patternt(Color a, Color b, const Matrix4x4& t) {
    Pattern p = mk_pattern(PatternDesign::Stripe, a, b);
    p.transform = t;
    return p;
}
Pattern gradient_pattern(Color a, Color b) { return mk_pattern(PatternDesign::Gradient, a, b); }
Pattern ring_pattern(Color a, Color b) { return mk_pattern(PatternDesign::Ring, a, b); }
Pattern checkers_pattern(Color a, Color b) { return mk_pattern(PatternDesign::Checkers, a, b); }
Pattern test_pattern() { return mk_pattern(PatternDesign::Test, black(), black()); }

bool Material::operator==(const Material& o) const {  // derived PartialEq, src/material.rs:7-18
    if (!(color == o.color)) return false;
    if (has_pattern != o.has_pattern) return false;
    if (has_pattern && !(pattern == o.pattern)) return false;
    return ambient == o.ambient && diffuse == o.diffuse && specular == o.specular && shininess == o.shininess &&
           reflective == o.reflective && transparency == o.transparency && refractive_index == o.refractive_index;
}

// ------------------------------------------------------------------ shapes
bool Shape::operator==(const Shape& other) const {
    return props.transform == other.props.transform && props.material == other.props.material && shape_eq(other);
}
bool Shape::includes(const Shape& other) const { return *this == other; }

Sphere Sphere::glass() {
    Sphere s;
    s.props.material.transparency = 1.0;
    s.props.material.refractive_index = 1.5;
    return s;
}
Cylinder::Cylinder()
    : min(-std::numeric_limits<double>::infinity()), max(std::numeric_limits<double>::infinity()), closed(false) {}
Cone::Cone() : min(-std::numeric_limits<double>::infinity()), max(std::numeric_limits<double>::infinity()), closed(false) {}

Triangle::Triangle(Tuple a, Tuple b, Tuple c) : p1(a), p2(b), p3(c) {
    e1 = p2 - p1;
    e2 = p3 - p1;
    normal = cross(e2, e1).normalize();
}
Triangle Triangle::smooth(Tuple p1, Tuple p2, Tuple p3, Tuple n1, Tuple n2, Tuple n3) {
    Triangle t(p1, p2, p3);
    t.is_smooth = true;
    t.normal = Tuple{};
    t.n1 = n1, t.n2 = n2, t.n3 = n3;
    return t;
}

void Group::refresh_parents() {  // src/shapes/group.rs:76-89
    Material material = props.material;
    std::vector<Matrix4x4> child_transforms = props.parent_transforms;
    child_transforms.push_back(props.transform);
    Material dflt;
    for (auto& ch : children) {
        if (ch->material() == dflt) ch->set_material(material);
        ch->props.parent_transforms = child_transforms;
        if (ch->kind() == Kind::Group) ch->refresh_parents();
    }
}
bool Group::includes(const Shape& other) const {
    for (auto& ch : children)
        if (ch->includes(other)) return true;
    return false;
}
bool Group::shape_eq(const Shape& other) const {
    return other.kind() == Kind::Group && children.size() == static_cast<const Group&>(other).children.size();
}
bool Csg::includes(const Shape& other) const { return left->includes(other) || right->includes(other); }
bool Csg::shape_eq(const Shape& other) const {
    if (other.kind() != Kind::Csg) return false;
    auto& o = static_cast<const Csg&>(other);
    return *left == *o.left && *right == *o.right;
}

// ------------------------------------------------------------------ canvas
void Canvas::quantize(Color c, uint8_t out[3]) {
    double ch[3] = {c.red, c.green, c.blue};
    for (int k = 0; k < 3; k++) {
        double r = std::round(ch[k] * 255.0);  // f64::round: half away from zero
        long long q = 0;                       // `as isize`: NaN -> 0, saturating
        if (r >= 9.2e18)
            q = 255;
        else if (r > -9.2e18)
            q = (long long)r;
        out[k] = (uint8_t)std::min(255LL, std::max(0LL, q));
    }
}
std::vector<uint8_t> Canvas::to_rgb8() const {
    if (has_u8()) return std::vector<uint8_t>(rgb8_.begin(), rgb8_.end());
    std::vector<uint8_t> out(pixels_.size() * 3);
    for (size_t i = 0; i < pixels_.size(); i++) quantize(pixels_[i], &out[3 * i]);
    return out;
}
std::string Canvas::to_ppm() const {
    std::string ppm = "P3\n" + std::to_string(width_) + " " + std::to_string(height_) + "\n255\n";
    std::vector<uint8_t> px = to_rgb8();
    for (size_t row = 0; row < height_; row++) {
        size_t col = 0;  // characters already on the current line
        for (size_t k = 0; k < width_ * 3; k++) {
            std::string text = std::to_string((unsigned)px[row * width_ * 3 + k]);
            if (col == 0) {
            } else if (text.size() + col + 1 > 70) {
                ppm += "\n";
                col = 0;
            } else {
                ppm += " ";
                col += 1;
            }
            ppm += text;
            col += text.size();
        }
        ppm += "\n";
    }
    return ppm;
}
void Canvas::save_ppm(const std::string& path) const {
    std::ofstream f(path, std::ios::binary);
    if (!f) throw Error("cannot write " + path);
    f << to_ppm();
}
static void png_chunk(std::ofstream& f, const char* type, const std::vector<uint8_t>& data) {
    uint32_t len = (uint32_t)data.size();
    uint8_t hdr[8] = {(uint8_t)(len >> 24), (uint8_t)(len >> 16), (uint8_t)(len >> 8), (uint8_t)len,
                      (uint8_t)type[0],     (uint8_t)type[1],     (uint8_t)type[2],    (uint8_t)type[3]};
    f.write((const char*)hdr, 8);
    if (len) f.write((const char*)data.data(), len);
    uLong crc = crc32(0L, hdr + 4, 4);
    if (len) crc = crc32(crc, data.data(), len);
    uint8_t c[4] = {(uint8_t)(crc >> 24), (uint8_t)(crc >> 16), (uint8_t)(crc >> 8), (uint8_t)crc};
    f.write((const char*)c, 4);
}
void Canvas::save_png(const std::string& path) const {
    if (has_u8()) return write_png8(path, rgb8_.data(), width_, height_);
    const std::vector<uint8_t> px = to_rgb8();
    write_png8(path, px.data(), width_, height_);
}
void Canvas::write_png8(const std::string& path, const uint8_t* px, size_t width_, size_t height_) {
    std::vector<uint8_t> raw;
    raw.reserve((width_ * 3 + 1) * height_);
    for (size_t y = 0; y < height_; y++) {
        raw.push_back(0);  // filter: none
        raw.insert(raw.end(), px + y * width_ * 3, px + (y + 1) * width_ * 3);
    }
    uLongf clen = compressBound(raw.size());
    std::vector<uint8_t> z(clen);
    if (compress2(z.data(), &clen, raw.data(), raw.size(), 6) != Z_OK) throw Error("png: zlib failure");
    z.resize(clen);
    std::ofstream f(path, std::ios::binary);
    if (!f) throw Error("cannot write " + path);
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    f.write((const char*)sig, 8);
    uint32_t w = (uint32_t)width_, h = (uint32_t)height_;
    std::vector<uint8_t> ihdr = {(uint8_t)(w >> 24), (uint8_t)(w >> 16), (uint8_t)(w >> 8), (uint8_t)w,
                                 (uint8_t)(h >> 24), (uint8_t)(h >> 16), (uint8_t)(h >> 8), (uint8_t)h,
                                 8,                  2,                  0,                 0,
                                 0};  // 8-bit RGB
    png_chunk(f, "IHDR", ihdr);
    png_chunk(f, "IDAT", z);
    png_chunk(f, "IEND", {});
}

// ------------------------------------------------------------------ world / camera
World World::default_world() {
    World w;
    w.lights.push_back({pt(-10, 10, -10), {1, 1, 1}});
    Material mat;
    mat.color = {0.8, 1.0, 0.6};
    mat.diffuse = 0.7;
    mat.specular = 0.2;
    w.objects.push_back(Sphere().material(mat).shape());
    w.objects.push_back(Sphere().transform(Matrix4x4::scaling(0.5, 0.5, 0.5)).shape());
    return w;
}
void World::refresh_parents() {
    for (auto& o : objects) o->refresh_parents();
}
Matrix4x4 view_transform(Tuple from, Tuple to, Tuple up) {  // src/scene.rs:157-169
    Tuple forward = (to - from).normalize();
    Tuple upn = up.normalize();
    Tuple left = cross(forward, upn);
    Tuple true_up = cross(left, forward);
    Matrix4x4 o = Matrix4x4::identity();
    o.m[0][0] = left.x, o.m[0][1] = left.y, o.m[0][2] = left.z;
    o.m[1][0] = true_up.x, o.m[1][1] = true_up.y, o.m[1][2] = true_up.z;
    o.m[2][0] = -forward.x, o.m[2][1] = -forward.y, o.m[2][2] = -forward.z;
    return o * Matrix4x4::translation(-from.x, -from.y, -from.z);
}
Camera::Camera(size_t hs, size_t vs, double fov) : hsize(hs), vsize(vs), field_of_view(fov) {  // :182-209
    double h = (double)hs, vv = (double)vs;
    double half_view = std::tan(fov / 2.0);
    double aspect = h / vv;
    if (aspect >= 1.0) {
        half_width_ = half_view;
        half_height_ = half_view / aspect;
    } else {
        half_width_ = half_view * aspect;
        half_height_ = half_view;
    }
    pixel_size_ = (half_width_ * 2.0) / h;
}

// ------------------------------------------------------------------ flattening
namespace {

Matrix4x4 inv_or_throw(const Matrix4x4& m, const char* what) {
    Matrix4x4 r;
    if (!m.inverse(&r)) throw Error(std::string("singular transform on ") + what + " (the reference panics in inverse().unwrap())");
    return r;
}
void put34(double* dst, const Matrix4x4& m) {
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 4; c++) dst[r * 4 + c] = m.m[r][c];
}
bool bits_eq(const Matrix4x4& a, const Matrix4x4& b) { return std::memcmp(&a, &b, sizeof(Matrix4x4)) == 0; }
bool mat_bits_eq(const Material& a, const Material& b) {
    if (!color_bits_eq(a.color, b.color) || a.has_pattern != b.has_pattern) return false;
    if (a.has_pattern && !(a.pattern.design == b.pattern.design && color_bits_eq(a.pattern.a, b.pattern.a) &&
                           color_bits_eq(a.pattern.b, b.pattern.b) && bits_eq(a.pattern.transform, b.pattern.transform)))
        return false;
    const double fa[7] = {a.ambient, a.diffuse, a.specular, a.shininess, a.reflective, a.transparency, a.refractive_index};
    const double fb[7] = {b.ambient, b.diffuse, b.specular, b.shininess, b.reflective, b.transparency, b.refractive_index};
    return std::memcmp(fa, fb, sizeof fa) == 0;
}

struct Flattener {
    FlatScene& fs;
    std::vector<Material> mats;            // host copies for de-duplication
    std::vector<const Shape*> leaf_shape;  // representative shape per leaf
    std::vector<int> leaf_root;            // csg root index or -1
    std::vector<Matrix4x4> ray_levels;     // inverse of every group and CSG object above the shape being visited, root first
    // open mesh run
    const Shape* run_parent = nullptr;
    bool run_open = false;
    int mesh_leaf_tris = kMeshLeafTris;

    explicit Flattener(FlatScene& f) : fs(f) {}

    int material_index(const Material& mt) {
        for (size_t i = 0; i < mats.size(); i++)
            if (mat_bits_eq(mats[i], mt)) return (int)i;
        mats.push_back(mt);
        rtamd_material d{};
        d.color[0] = mt.color.red, d.color[1] = mt.color.green, d.color[2] = mt.color.blue;
        d.ambient = mt.ambient, d.diffuse = mt.diffuse, d.specular = mt.specular, d.shininess = mt.shininess;
        d.reflective = mt.reflective, d.transparency = mt.transparency, d.refractive_index = mt.refractive_index;
        d.pattern_kind = RTAMD_PAT_NONE;
        put34(d.pat_inv, Matrix4x4::identity());
        if (mt.has_pattern) {
            switch (mt.pattern.design) {
                case PatternDesign::Stripe: d.pattern_kind = RTAMD_PAT_STRIPE; break;
                case PatternDesign::Gradient: d.pattern_kind = RTAMD_PAT_GRADIENT; break;
                case PatternDesign::Ring: d.pattern_kind = RTAMD_PAT_RING; break;
                case PatternDesign::Checkers: d.pattern_kind = RTAMD_PAT_CHECKERS; break;
                case PatternDesign::Test: d.pattern_kind = RTAMD_PAT_TEST; break;
            }
            d.pat_a[0] = mt.pattern.a.red, d.pat_a[1] = mt.pattern.a.green, d.pat_a[2] = mt.pattern.a.blue;
            d.pat_b[0] = mt.pattern.b.red, d.pat_b[1] = mt.pattern.b.green, d.pat_b[2] = mt.pattern.b.blue;
            put34(d.pat_inv, inv_or_throw(mt.pattern.transform, "a pattern"));
        }
        fs.materials.push_back(d);
        return (int)mats.size() - 1;
    }

    int push_chain(const Shape& s) {
        int start = (int)(fs.chain.size() / 12);
        size_t levels = s.props.parent_transforms.size() + 1;
        if (levels > RTAMD_MAX_CHAIN) throw Error("group nesting deeper than the device limit (RTAMD_MAX_CHAIN)");
        for (auto& p : s.props.parent_transforms) {
            fs.chain.resize(fs.chain.size() + 12);
            put34(&fs.chain[fs.chain.size() - 12], inv_or_throw(p, "a group"));
        }
        fs.chain.resize(fs.chain.size() + 12);
        put34(&fs.chain[fs.chain.size() - 12], inv_or_throw(s.props.transform, "a shape"));
        return start;
    }

    void push_triangle_data(const Triangle& t) {
        const Tuple g[3] = {t.p1, t.e1, t.e2};
        for (auto& q : g) fs.tri_geom.insert(fs.tri_geom.end(), {q.x, q.y, q.z});
        if (t.is_smooth) {
            const Tuple n[3] = {t.n1, t.n2, t.n3};
            for (auto& q : n) fs.tri_normals.insert(fs.tri_normals.end(), {q.x, q.y, q.z});
        } else {
            fs.tri_normals.insert(fs.tri_normals.end(), {t.normal.x, t.normal.y, t.normal.z, 0, 0, 0, 0, 0, 0});
        }
    }

    void leaf(const Shape& s, const Matrix4x4& accum, const Shape* parent, int root) {
        int mat = material_index(s.material());
        if (s.kind() == Kind::Triangle) {
            auto& t = static_cast<const Triangle&>(s);
            if (run_open && run_parent == parent && !fs.leaves.empty()) {
                rtamd_leaf& L = fs.leaves.back();
                const Shape* first = leaf_shape.back();
                bool smooth = (L.flags & RTAMD_LEAF_SMOOTH) != 0;
                bool shadow = (L.flags & RTAMD_LEAF_SHADOW) != 0;
                if (L.kind == RTAMD_MESH && L.tri_count < mesh_leaf_tris && L.material == mat && smooth == t.is_smooth && shadow == s.shadow() &&
                    bits_eq(first->transform(), s.transform()) && leaf_root.back() == root &&
                    first->parent_transforms().size() == s.parent_transforms().size() &&
                    (s.parent_transforms().empty() ||
                     std::memcmp(first->parent_transforms().data(), s.parent_transforms().data(),
                                 sizeof(Matrix4x4) * s.parent_transforms().size()) == 0)) {
                    L.tri_count++;
                    push_triangle_data(t);
                    return;
                }
            }
        }
        rtamd_leaf L{};
        Matrix4x4 w2o = inv_or_throw(s.transform(), "a shape") * accum;
        put34(L.w2o, w2o);
        L.material = mat;
        L.flags = s.shadow() ? RTAMD_LEAF_SHADOW : 0;
        L.chain_start = push_chain(s);
        L.chain_len = (int)s.parent_transforms().size() + 1;
        // The levels a ray passes on its way to the leaf (src/shape.rs:39-42 at every group and CSG object above it).  Groups
        // alone above the leaf: the levels of the chain just written.  Below a CSG object the leaf's parent_transforms stop at
        // (or never saw) that object, so the levels get a run of `chain` of their own.
        bool same_levels = root < 0 && ray_levels.size() + 1 == (size_t)L.chain_len;
        for (size_t i = 0; same_levels && i < ray_levels.size(); i++) {
            double lv[12];
            put34(lv, ray_levels[i]);
            same_levels = std::memcmp(lv, &fs.chain[((size_t)L.chain_start + i) * 12], sizeof lv) == 0;
        }
        if (same_levels) {
            L.ray_start = L.chain_start, L.ray_len = L.chain_len;
        } else {
            if (ray_levels.size() + 1 > RTAMD_MAX_RAY_LEVELS) throw Error("groups and CSG objects nested deeper than the device limit (RTAMD_MAX_RAY_LEVELS)");
            L.ray_start = (int)(fs.chain.size() / 12), L.ray_len = (int)ray_levels.size() + 1;
            for (auto& lv : ray_levels) {
                fs.chain.resize(fs.chain.size() + 12);
                put34(&fs.chain[fs.chain.size() - 12], lv);
            }
            fs.chain.resize(fs.chain.size() + 12);
            put34(&fs.chain[fs.chain.size() - 12], inv_or_throw(s.props.transform, "a shape"));
        }
        run_open = false;
        switch (s.kind()) {
            case Kind::Sphere: L.kind = RTAMD_SPHERE; break;
            case Kind::Plane: L.kind = RTAMD_PLANE; break;
            case Kind::Cube: L.kind = RTAMD_CUBE; break;
            case Kind::Cylinder: {
                auto& c = static_cast<const Cylinder&>(s);
                L.kind = RTAMD_CYLINDER, L.p0 = c.min, L.p1 = c.max;
                if (c.closed) L.flags |= RTAMD_LEAF_CLOSED;
                break;
            }
            case Kind::Cone: {
                auto& c = static_cast<const Cone&>(s);
                L.kind = RTAMD_CONE, L.p0 = c.min, L.p1 = c.max;
                if (c.closed) L.flags |= RTAMD_LEAF_CLOSED;
                break;
            }
            case Kind::Triangle: {
                auto& t = static_cast<const Triangle&>(s);
                L.kind = RTAMD_MESH;
                if (t.is_smooth) L.flags |= RTAMD_LEAF_SMOOTH;
                L.tri_start = (int)(fs.tri_geom.size() / 9);
                L.tri_count = 1;
                push_triangle_data(t);
                run_open = true;
                run_parent = parent;
                break;
            }
            default: throw Error("internal: not a leaf");
        }
        fs.leaves.push_back(L);
        leaf_shape.push_back(&s);
        leaf_root.push_back(root);
    }

    // returns nothing; nodes of a CSG root are appended in post-order
    void visit(const Shape& s, const Matrix4x4& accum, const Shape* parent, int root) {
        if (s.kind() == Kind::Group) {
            auto& g = static_cast<const Group&>(s);
            Matrix4x4 own = inv_or_throw(g.transform(), "a group");
            Matrix4x4 a = own * accum;
            run_open = false;
            ray_levels.push_back(own);
            for (auto& ch : g.children) visit(*ch, a, &s, root);
            ray_levels.pop_back();
            run_open = false;
        } else if (s.kind() == Kind::Csg) {
            auto& c = static_cast<const Csg&>(s);
            Matrix4x4 own = inv_or_throw(c.transform(), "a csg");
            Matrix4x4 a = own * accum;
            run_open = false;
            ray_levels.push_back(own);
            bool top = root < 0;
            int my_root = root;
            if (top) {
                my_root = (int)fs.csg_roots.size();
                rtamd_csg_root r{};
                r.node_start = (int)fs.csg_nodes.size();
                r.leaf_lo = (int)fs.leaves.size();
                fs.csg_roots.push_back(r);
            }
            rtamd_csg_node n{};
            n.op = c.op == Op::Union ? RTAMD_CSG_UNION : c.op == Op::Intersection ? RTAMD_CSG_INTERSECTION : RTAMD_CSG_DIFFERENCE;
            n.leaf_lo = (int)fs.leaves.size();
            visit(*c.left, a, &s, my_root);
            run_open = false;
            n.leaf_mid = (int)fs.leaves.size();
            visit(*c.right, a, &s, my_root);
            run_open = false;
            n.leaf_hi = (int)fs.leaves.size();
            ray_levels.pop_back();
            fs.csg_nodes.push_back(n);
            if (top) {
                rtamd_csg_root& r = fs.csg_roots[my_root];
                r.node_count = (int)fs.csg_nodes.size() - r.node_start;
                r.leaf_hi = (int)fs.leaves.size();
                if (r.leaf_hi - r.leaf_lo > RTAMD_MAX_CSG_LEAVES) throw Error("CSG object with more leaves than the device limit (RTAMD_MAX_CSG_LEAVES)");
            }
        } else {
            leaf(s, accum, parent, root);
        }
    }

    void classes_and_visits() {
        size_t n = fs.leaves.size();
        int n_classes = 0;
        std::vector<int> first_of_class;
        for (size_t i = 0; i < n; i++) {
            int cls = -1;
            for (size_t k = 0; k < first_of_class.size(); k++)
                if (*leaf_shape[first_of_class[k]] == *leaf_shape[i]) {
                    cls = (int)k;
                    break;
                }
            if (cls < 0) {
                cls = n_classes++;
                first_of_class.push_back((int)i);
            }
            fs.leaves[i].eq_class = cls;
        }
        fs.view.n_classes = n_classes;
        for (auto& nd : fs.csg_nodes) {
            nd.left_mask = 0;
            for (int l = nd.leaf_lo; l < nd.leaf_hi; l++) {
                bool left = l < nd.leaf_mid;
                for (int q = nd.leaf_lo; q < nd.leaf_mid && !left; q++) left = fs.leaves[q].eq_class == fs.leaves[l].eq_class;
                if (left) nd.left_mask |= (uint64_t)1 << (l - nd.leaf_lo);
            }
        }
        std::vector<char> root_seen(fs.csg_roots.size(), 0);
        for (int cls = 0; cls < n_classes; cls++) {
            int last_root = -1;
            size_t first_visit = fs.visits.size();
            for (size_t i = 0; i < n; i++) {
                if (fs.leaves[i].eq_class != cls) continue;
                rtamd_visit v{};
                v.eq_class = cls;
                v.class_material = fs.leaves[first_of_class[cls]].material;
                if (leaf_root[i] < 0) {
                    v.unit = (int)i;
                    v.flags = RTAMD_VISIT_FIRST;
                } else {
                    if (leaf_root[i] == last_root) continue;  // this root is already visited for this class
                    last_root = leaf_root[i];
                    v.unit = leaf_root[i];
                    v.flags = RTAMD_VISIT_CSG;
                    if (!root_seen[v.unit]) {
                        v.flags |= RTAMD_VISIT_FIRST;
                        root_seen[v.unit] = 1;
                    }
                }
                fs.visits.push_back(v);
            }
            if (fs.visits.size() > first_visit) fs.visits.back().flags |= RTAMD_VISIT_CLASS_END;
        }
    }
};

}  // namespace

void FlatScene::finish() {
    view.n_lights = (int)lights.size(), view.lights = lights.data();
    view.n_leaves = (int)leaves.size(), view.leaves = leaves.data();
    view.n_chain = (int)(chain.size() / 12), view.chain = chain.data();
    view.n_tris = (int)(tri_geom.size() / 9), view.tri_geom = tri_geom.data(), view.tri_normals = tri_normals.data();
    view.n_materials = (int)materials.size(), view.materials = materials.data();
    view.n_csg_nodes = (int)csg_nodes.size(), view.csg_nodes = csg_nodes.data();
    view.n_csg_roots = (int)csg_roots.size(), view.csg_roots = csg_roots.data();
    view.n_visits = (int)visits.size(), view.visits = visits.data();
}

std::unique_ptr<FlatScene> flatten(World& world, const Camera& camera, int max_remaining) {
    world.refresh_parents();  // Camera::render does this first, src/scene.rs:237
    auto fs = std::make_unique<FlatScene>();
    if (world.lights.size() > RTAMD_MAX_LIGHTS) throw Error("more lights than the device limit (RTAMD_MAX_LIGHTS)");
    if (max_remaining < 0 || max_remaining > RTAMD_MAX_REMAINING) throw Error("recursion budget beyond the device limit");
    fs->view.camera.hsize = (int)camera.hsize, fs->view.camera.vsize = (int)camera.vsize;
    fs->view.camera.pixel_size = camera.pixel_size();
    fs->view.camera.half_width = camera.half_width(), fs->view.camera.half_height = camera.half_height();
    put34(fs->view.camera.inv, inv_or_throw(camera.transform, "the camera"));
    fs->view.max_remaining = max_remaining;
    for (auto& l : world.lights) {
        rtamd_light d{};
        d.position[0] = l.position.x, d.position[1] = l.position.y, d.position[2] = l.position.z;
        d.intensity[0] = l.intensity.red, d.intensity[1] = l.intensity.green, d.intensity[2] = l.intensity.blue;
        fs->lights.push_back(d);
    }
    Flattener f(*fs);
    // (how many triangles the device tests as one run is the device library's choice: rtamd_scene_upload cuts the pools of
    // triangles again; here consecutive triangles simply fill leaves of kMeshLeafTris)
    if (const char* e = std::getenv("RT_MESH_LEAF")) f.mesh_leaf_tris = std::max(1, std::min(kMeshLeafTris, std::atoi(e)));  // (diagnostic)
    for (auto& o : world.objects) {
        f.visit(*o, Matrix4x4::identity(), nullptr, -1);
    }
    f.classes_and_visits();
    fs->finish();
    return fs;
}

// ------------------------------------------------------------------ render
namespace {
std::mutex g_ctx_mu;
rtamd_ctx* g_ctx = nullptr;
rtamd_ctx* default_ctx() {
    std::lock_guard<std::mutex> lk(g_ctx_mu);
    if (!g_ctx) {
        int dev = 0;
        if (const char* e = std::getenv("RTAMD_DEVICE")) dev = std::atoi(e);
        rtamd_ctx* c = nullptr;
        int rc = rtamd_ctx_create(dev, &c);
        if (rc != RTAMD_OK) throw Error(std::string("rtamd_ctx_create failed: ") + rtamd_last_error(nullptr));
        g_ctx = c;
    }
    return g_ctx;
}
double ms_since(std::chrono::steady_clock::time_point t0) {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}
}  // namespace

Canvas Camera::render(World& world, int mode, RenderStats* stats, int want) const {
    if (!(want & (RTAMD_WANT_F64 | RTAMD_WANT_U8))) throw Error("render: neither the f64 pixels nor the 8-bit channels were asked for");
    Canvas image(hsize, vsize, want, false);  // (every pixel is written by the render)
    render_into(image, world, mode, stats);
    return image;
}

void Camera::render_into(Canvas& image, World& world, int mode, RenderStats* stats) const {
    if (image.width() != hsize || image.height() != vsize) throw Error("render: the canvas is not of the camera's size");
    auto t0 = std::chrono::steady_clock::now();
    auto fs = flatten(world, *this);
    double flatten_ms = ms_since(t0);
    rtamd_ctx* ctx = default_ctx();
    t0 = std::chrono::steady_clock::now();
    if (rtamd_scene_upload(ctx, &fs->view) != RTAMD_OK) throw Error(std::string("scene upload: ") + rtamd_last_error(ctx));
    double upload_ms = ms_since(t0);
    static_assert(sizeof(Color) == 3 * sizeof(double), "Color must be three packed doubles");
    if (stats) rtamd_reset_stats(ctx);
    t0 = std::chrono::steady_clock::now();
    int rc = rtamd_render(ctx, mode, image.pixels().empty() ? nullptr : reinterpret_cast<double*>(image.pixels().data()),
                          image.rgb8().empty() ? nullptr : image.rgb8().data());
    if (rc != RTAMD_OK) throw Error(std::string("render: ") + rtamd_last_error(ctx));
    if (stats) {
        stats->render_ms = ms_since(t0);
        stats->flatten_ms = flatten_ms, stats->upload_ms = upload_ms;
        rtamd_get_stats(ctx, &stats->gpu);
    }
}

}  // namespace rt
