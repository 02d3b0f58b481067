// Host-side scene API: the C++ mirror of the reference's public types for the render path
// (World, Camera, shapes, Material, Pattern, Canvas, ObjParser, the YAML scene loader).
// The host builds a scene with these, `Camera::render` flattens it once into the plain
// `rtamd_scene` of include/rtamd.h and hands it to the GPU library; there is no CPU renderer
// behind this API -- without the HIP library and a GPU, render() throws.
#pragma once
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <new>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/rtamd.h"
#include "rt_math.hpp"

namespace rt {

struct Error : std::runtime_error {
    using std::runtime_error::runtime_error;
};

// ---- src/pattern.rs:3-16, 64-104
enum class PatternDesign { Stripe, Gradient, Ring, Checkers, Test };
struct Pattern {
    PatternDesign design = PatternDesign::Test;
    Color a, b;
    Matrix4x4 transform = Matrix4x4::identity();
    bool operator==(const Pattern& o) const;
};
Pattern stripe_pattern(Color a, Color b);
Pattern stripe_patternt(Color a, Color b, const Matrix4x4& transform);
Pattern gradient_pattern(Color a, Color b);
Pattern ring_pattern(Color a, Color b);
Pattern checkers_pattern(Color a, Color b);
Pattern test_pattern();

// ---- src/material.rs:7-79
struct Material {
    Color color{1, 1, 1};
    bool has_pattern = false;
    Pattern pattern;
    double ambient = 0.1, diffuse = 0.9, specular = 0.9, shininess = 200.0;
    double reflective = 0.0, transparency = 0.0, refractive_index = 1.0;
    bool operator==(const Material& o) const;
    Material& rgb(double r, double g, double b) { color = {r, g, b}; return *this; }
    Material& set_pattern(const Pattern& p) { has_pattern = true; pattern = p; return *this; }
};
inline Material m() { return Material(); }

struct PointLight {  // src/ray.rs:26-48
    Tuple position = pt(0, 0, 0);
    Color intensity{1, 1, 1};
};

// ---- src/shape.rs:5-21
struct Props {
    Matrix4x4 transform = Matrix4x4::identity();
    Material material;
    std::vector<Matrix4x4> parent_transforms;
    bool shadow = true;
};

enum class Kind { Sphere, Plane, Cube, Cylinder, Cone, Triangle, Group, Csg };
enum class Op { Union, Intersection, Difference };  // src/shapes/csg.rs:5-9

class Shape {
   public:
    Shape() = default;
    Shape(const Shape&) = default;
    Shape(Shape&&) noexcept = default;
    Shape& operator=(const Shape&) = default;
    Shape& operator=(Shape&&) noexcept = default;
    virtual ~Shape() = default;
    virtual Kind kind() const = 0;
    Props props;
    const Matrix4x4& transform() const { return props.transform; }
    void set_transform(const Matrix4x4& t) { props.transform = t; }
    const Material& material() const { return props.material; }
    Material& material_mut() { return props.material; }
    void set_material(const Material& mat) { props.material = mat; }
    bool shadow() const { return props.shadow; }
    void set_shadow(bool s) { props.shadow = s; }
    const std::vector<Matrix4x4>& parent_transforms() const { return props.parent_transforms; }
    virtual void refresh_parents() {}                 // src/shape.rs:98
    virtual bool includes(const Shape& other) const;  // leaf: value equality
    // `PartialEq for dyn Shape` (src/shape.rs:23-27,119-123)
    bool operator==(const Shape& other) const;

   protected:
    virtual bool shape_eq(const Shape& other) const { return kind() == other.kind(); }
};
typedef std::unique_ptr<Shape> ShapePtr;

template <class D>
class ShapeT : public Shape {
   public:
    D& transform(const Matrix4x4& t) { props.transform = t; return static_cast<D&>(*this); }
    using Shape::transform;
    D& material(const Material& mat) { props.material = mat; return static_cast<D&>(*this); }
    using Shape::material;
    D& shadow(bool s) { props.shadow = s; return static_cast<D&>(*this); }
    using Shape::shadow;
    ShapePtr shape() { return std::make_unique<D>(std::move(static_cast<D&>(*this))); }
};

class Sphere : public ShapeT<Sphere> {
   public:
    Kind kind() const override { return Kind::Sphere; }
    static Sphere glass();  // src/shapes/sphere.rs:16-21
};
class Plane : public ShapeT<Plane> {
   public:
    Kind kind() const override { return Kind::Plane; }
};
class Cube : public ShapeT<Cube> {
   public:
    Kind kind() const override { return Kind::Cube; }
};
class Cylinder : public ShapeT<Cylinder> {  // src/shapes/cylinder.rs:6-42
   public:
    double min, max;
    bool closed;
    Cylinder();  // unbounded, open
    Cylinder(double min, double max, bool closed) : min(min), max(max), closed(closed) {}
    Kind kind() const override { return Kind::Cylinder; }
};
class Cone : public ShapeT<Cone> {  // src/shapes/cone.rs:6-42
   public:
    double min, max;
    bool closed;
    Cone();
    Cone(double min, double max, bool closed) : min(min), max(max), closed(closed) {}
    Kind kind() const override { return Kind::Cone; }
};
class Triangle : public ShapeT<Triangle> {  // src/shapes/triangle.rs:5-49
   public:
    Tuple p1, p2, p3, e1, e2;
    bool is_smooth = false;
    Tuple normal;      // Normal::Default
    Tuple n1, n2, n3;  // Normal::Smooth
    Triangle(Tuple p1, Tuple p2, Tuple p3);
    static Triangle smooth(Tuple p1, Tuple p2, Tuple p3, Tuple n1, Tuple n2, Tuple n3);
    Kind kind() const override { return Kind::Triangle; }
};
class Group : public ShapeT<Group> {  // src/shapes/group.rs
   public:
    std::vector<ShapePtr> children;
    Kind kind() const override { return Kind::Group; }
    void add_child(ShapePtr c) { children.push_back(std::move(c)); }
    void refresh_parents() override;
    bool includes(const Shape& other) const override;

   protected:
    bool shape_eq(const Shape& other) const override;
};
class Csg : public ShapeT<Csg> {  // src/shapes/csg.rs
   public:
    Op op;
    ShapePtr left, right;
    Csg(Op op, ShapePtr l, ShapePtr r) : op(op), left(std::move(l)), right(std::move(r)) {}
    static Csg union_(ShapePtr l, ShapePtr r) { return Csg(Op::Union, std::move(l), std::move(r)); }
    static Csg difference(ShapePtr l, ShapePtr r) { return Csg(Op::Difference, std::move(l), std::move(r)); }
    static Csg intersection(ShapePtr l, ShapePtr r) { return Csg(Op::Intersection, std::move(l), std::move(r)); }
    Kind kind() const override { return Kind::Csg; }
    bool includes(const Shape& other) const override;

   protected:
    bool shape_eq(const Shape& other) const override;
};

// Host memory for pixels: page-locked (rtamd_host_alloc) where the GPU library can give such memory, so that a frame
// arrives at the rate of the PCIe link; ordinary memory on a machine without a GPU (file conversions, tests).
template <class T>
class HostBuf {
   public:
    HostBuf() = default;
    HostBuf(size_t n, bool zeroed) : n_(n) {
        if (!n) return;
        p_ = static_cast<T*>(rtamd_host_alloc(n * sizeof(T)));
        pinned_ = p_ != nullptr;
        if (!p_) p_ = static_cast<T*>(std::malloc(n * sizeof(T)));
        if (!p_) throw std::bad_alloc();
        if (zeroed) std::memset(static_cast<void*>(p_), 0, n * sizeof(T));
    }
    HostBuf(const HostBuf& o) : HostBuf(o.n_, false) {
        if (n_) std::memcpy(static_cast<void*>(p_), static_cast<const void*>(o.p_), n_ * sizeof(T));
    }
    HostBuf(HostBuf&& o) noexcept : p_(o.p_), n_(o.n_), pinned_(o.pinned_) { o.p_ = nullptr, o.n_ = 0; }
    HostBuf& operator=(HostBuf o) noexcept {
        std::swap(p_, o.p_), std::swap(n_, o.n_), std::swap(pinned_, o.pinned_);
        return *this;
    }
    ~HostBuf() {
        if (p_ && pinned_) rtamd_host_free(p_);
        if (p_ && !pinned_) std::free(p_);
    }
    T* data() { return p_; }
    const T* data() const { return p_; }
    size_t size() const { return n_; }
    bool empty() const { return n_ == 0; }
    bool page_locked() const { return pinned_; }
    T& operator[](size_t i) { return p_[i]; }
    const T& operator[](size_t i) const { return p_[i]; }
    const T& at(size_t i) const {
        if (i >= n_) throw std::out_of_range("pixel index out of range");
        return p_[i];
    }
    T* begin() { return p_; }
    T* end() { return p_ + n_; }
    const T* begin() const { return p_; }
    const T* end() const { return p_ + n_; }

   private:
    T* p_ = nullptr;
    size_t n_ = 0;
    bool pinned_ = false;
};

// ---- src/tuple.rs:309-391
// A canvas holds the f64 pixels (Canvas.pixels), the 8-bit channels the device quantised from them
// (`From<Color> for (usize, usize, usize)`, src/tuple.rs:226-237 -- all a PNG or PPM file needs: 3 bytes per pixel
// instead of 24 over the PCIe link), or both: RTAMD_WANT_F64 / RTAMD_WANT_U8.
class Canvas {
   public:
    Canvas(size_t width, size_t height, int want = RTAMD_WANT_F64, bool zeroed = true)
        : width_(width), height_(height), pixels_((want & RTAMD_WANT_F64) ? width * height : 0, zeroed),
          rgb8_((want & RTAMD_WANT_U8) ? width * height * 3 : 0, zeroed) {}
    size_t width() const { return width_; }
    size_t height() const { return height_; }
    bool has_f64() const { return !pixels_.empty() || width_ * height_ == 0; }
    bool has_u8() const { return !rgb8_.empty(); }
    void write_pixel(size_t x, size_t y, Color c) {
        if (x >= width_ || y >= height_ || pixels_.empty()) return;
        pixels_[x + y * width_] = c;
    }
    Color pixel_at(size_t x, size_t y) const { return pixels_.at(x + y * width_); }
    HostBuf<Color>& pixels() { return pixels_; }
    const HostBuf<Color>& pixels() const { return pixels_; }
    HostBuf<uint8_t>& rgb8() { return rgb8_; }
    const HostBuf<uint8_t>& rgb8() const { return rgb8_; }
    // `From<Color> for (usize, usize, usize)`, src/tuple.rs:226-237
    static void quantize(Color c, uint8_t out[3]);
    std::vector<uint8_t> to_rgb8() const;          // the device's 8-bit channels when the canvas holds them, else quantised here
    std::string to_ppm() const;                    // P3, lines wrapped at 70 columns, :343-371
    void save_png(const std::string& path) const;  // what to_image().save() writes, :373-390
    void save_ppm(const std::string& path) const;
    static void write_png8(const std::string& path, const uint8_t* rgb8, size_t width, size_t height);

   private:
    size_t width_, height_;
    HostBuf<Color> pixels_;
    HostBuf<uint8_t> rgb8_;
};

struct RenderStats {
    rtamd_stats gpu{};
    double flatten_ms = 0, upload_ms = 0, render_ms = 0;
};

// ---- src/scene.rs
struct World {
    std::vector<ShapePtr> objects;
    std::vector<PointLight> lights;
    static World empty() { return World(); }
    static World default_world();  // `impl Default for World`, :139-155
    void refresh_parents();
};
Matrix4x4 view_transform(Tuple from, Tuple to, Tuple up);

class Camera {
   public:
    size_t hsize, vsize;
    double field_of_view;
    Matrix4x4 transform = Matrix4x4::identity();
    Camera(size_t hsize, size_t vsize, double field_of_view);
    double pixel_size() const { return pixel_size_; }
    double half_width() const { return half_width_; }
    double half_height() const { return half_height_; }
    // Camera::render (src/scene.rs:236-248) on the GPU. mode: RTAMD_MODE_*; want: RTAMD_WANT_F64 | RTAMD_WANT_U8, what
    // the canvas is to hold (a caller that only writes an image file asks for the 8-bit channels alone).
    Canvas render(World& world, int mode = RTAMD_MODE_FAST, RenderStats* stats = nullptr, int want = RTAMD_WANT_F64) const;
    // The same into a canvas that exists (of this camera's size; it keeps what it holds: f64, 8-bit or both): a host
    // that renders frame after frame pins its memory once.
    void render_into(Canvas& image, World& world, int mode = RTAMD_MODE_FAST, RenderStats* stats = nullptr) const;

   private:
    double pixel_size_, half_width_, half_height_;
};

// Flattened scene: owns the arrays an `rtamd_scene` points into.
struct FlatScene {
    rtamd_scene view{};
    std::vector<rtamd_light> lights;
    std::vector<rtamd_leaf> leaves;
    std::vector<double> chain, tri_geom, tri_normals;
    std::vector<rtamd_material> materials;
    std::vector<rtamd_csg_node> csg_nodes;
    std::vector<rtamd_csg_root> csg_roots;
    std::vector<rtamd_visit> visits;
    void finish();  // point `view` at the vectors
};
// World (after refresh_parents) + Camera -> FlatScene. Throws rt::Error on a singular
// transform (the reference panics there) and on scenes beyond the device limits.
std::unique_ptr<FlatScene> flatten(World& world, const Camera& camera, int max_remaining = 4);

// ---- src/obj.rs
class ObjParser {
   public:
    static ObjParser from_str(const std::string& contents);
    static ObjParser from_file(const std::string& path);
    size_t lines_ignored = 0;
    std::vector<Tuple> vertices, normals;
    Group default_group;
    std::vector<std::pair<std::string, Group>> groups;  // insertion order
    Group into_group();

   private:
    std::vector<Triangle> fan_triangulation(const std::vector<std::pair<size_t, size_t>>& idxs) const;
    Group* current_ = nullptr;
    std::string current_name_;
    bool has_current_ = false;
};

// ---- src/bin/yaml_renderer.rs
struct Scene {
    std::unique_ptr<Camera> camera;
    World world;
};
std::unique_ptr<Scene> load_yaml_scene(const std::string& path);
std::unique_ptr<Scene> load_yaml_scene_str(const std::string& text);

// Scene-graph text (".sg"): a loss-free dump of World + Camera (hierarchy, own transforms,
// materials) used to hand the same scene to the GPU path and to an external checker.
void save_sg(const Scene& scene, const std::string& path);
std::unique_ptr<Scene> load_sg(const std::string& path);

}  // namespace rt
