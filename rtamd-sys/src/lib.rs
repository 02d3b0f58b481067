//! `extern "C"` mirror of `include/rtamd.h` and `include/rtamd_host.h`, one to one: every struct is `#[repr(C)]` with
//! the header's field names, types and order (`tests/test_host.py` compares these lists with the ctypes mirror the
//! test-suite renders through), every function the headers declare is declared here.
//!
//! What the calls replace in the reference: `Camera::render` (`src/scene.rs:236-248`).  The host keeps its `World`
//! (`src/scene.rs:6-9`), hands it over node by node through the `rth_*` scene builder, and gets `Canvas.pixels` back.
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_double, c_int, c_void};

pub const RTAMD_OK: c_int = 0;
pub const RTAMD_E_INVALID: c_int = -1;
pub const RTAMD_E_HIP: c_int = -2;
pub const RTAMD_E_NO_SCENE: c_int = -3;
pub const RTAMD_E_LIMIT: c_int = -4;
pub const RTAMD_E_OVERFLOW: c_int = -5;
pub const RTAMD_MODE_STRICT: c_int = 0;
pub const RTAMD_MODE_FAST: c_int = 1;
pub const RTAMD_WANT_F64: c_int = 1;
pub const RTAMD_WANT_U8: c_int = 2;
pub const RTAMD_CSG_UNION: c_int = 0;
pub const RTAMD_CSG_INTERSECTION: c_int = 1;
pub const RTAMD_CSG_DIFFERENCE: c_int = 2;
pub const RTH_PAT_NONE: i32 = 0;
pub const RTH_PAT_STRIPE: i32 = 1;
pub const RTH_PAT_GRADIENT: i32 = 2;
pub const RTH_PAT_RING: i32 = 3;
pub const RTH_PAT_CHECKERS: i32 = 4;
pub const RTH_PAT_TEST: i32 = 5;

// ---- include/rtamd.h
#[repr(C)]
pub struct rtamd_leaf {
    pub w2o: [f64; 12],
    pub p0: f64,
    pub p1: f64,
    pub kind: i32,
    pub flags: u32,
    pub material: i32,
    pub eq_class: i32,
    pub chain_start: i32,
    pub chain_len: i32,
    pub tri_start: i32,
    pub tri_count: i32,
    pub ray_start: i32,
    pub ray_len: i32,
}
#[repr(C)]
pub struct rtamd_material {
    pub color: [f64; 3],
    pub ambient: f64,
    pub diffuse: f64,
    pub specular: f64,
    pub shininess: f64,
    pub reflective: f64,
    pub transparency: f64,
    pub refractive_index: f64,
    pub pat_a: [f64; 3],
    pub pat_b: [f64; 3],
    pub pat_inv: [f64; 12],
    pub pattern_kind: i32,
    pub pad_: i32,
}
#[repr(C)]
pub struct rtamd_csg_node {
    pub left_mask: u64,
    pub op: i32,
    pub leaf_lo: i32,
    pub leaf_mid: i32,
    pub leaf_hi: i32,
    pub pad_: i32,
}
#[repr(C)]
pub struct rtamd_csg_root {
    pub node_start: i32,
    pub node_count: i32,
    pub leaf_lo: i32,
    pub leaf_hi: i32,
}
#[repr(C)]
pub struct rtamd_visit {
    pub unit: i32,
    pub eq_class: i32,
    pub flags: u32,
    pub class_material: i32,
}
#[repr(C)]
pub struct rtamd_camera {
    pub hsize: i32,
    pub vsize: i32,
    pub pixel_size: f64,
    pub half_width: f64,
    pub half_height: f64,
    pub inv: [f64; 12],
}
#[repr(C)]
pub struct rtamd_light {
    pub position: [f64; 3],
    pub intensity: [f64; 3],
}
#[repr(C)]
pub struct rtamd_scene {
    pub camera: rtamd_camera,
    pub max_remaining: i32,
    pub n_lights: i32,
    pub lights: *const rtamd_light,
    pub n_leaves: i32,
    pub leaves: *const rtamd_leaf,
    pub n_chain: i32,
    pub chain: *const f64,
    pub n_tris: i32,
    pub tri_geom: *const f64,
    pub tri_normals: *const f64,
    pub n_materials: i32,
    pub materials: *const rtamd_material,
    pub n_csg_nodes: i32,
    pub csg_nodes: *const rtamd_csg_node,
    pub n_csg_roots: i32,
    pub csg_roots: *const rtamd_csg_root,
    pub n_visits: i32,
    pub visits: *const rtamd_visit,
    pub n_classes: i32,
}
#[repr(C)]
pub struct rtamd_rows {
    pub tile_rows: i32,
    pub part: i32,
    pub n_parts: i32,
    pub span: i32,
}
#[repr(C)]
#[derive(Default)]
pub struct rtamd_stats {
    pub rays: u64,
    pub primary_rays: u64,
    pub shadow_rays: u64,
    pub reflect_rays: u64,
    pub refract_rays: u64,
    pub container_passes: u64,
    pub shade_points: u64,
    pub light_samples: u64,
    pub csg_overflows: u64,
    pub kernel_ms: f64,
    pub tests_sphere: u64,
    pub tests_plane: u64,
    pub tests_cube: u64,
    pub tests_slab2: u64,
    pub tests_slab3: u64,
    pub tests_cylinder: u64,
    pub tests_cone: u64,
    pub tests_caps: u64,
    pub tests_tri_first: u64,
    pub tests_tri_second: u64,
    pub tests_box: u64,
    pub tests_run_transform: u64,
    pub team_shade_points: u64,
}
#[repr(C)]
#[derive(Default)]
pub struct rtamd_layout {
    pub first_row: i64,
    pub chunk_rows: i64,
    pub stride_rows: i64,
    pub n_chunks: i64,
    pub tail_rows: i64,
}
pub enum rtamd_ctx {}
pub enum rtamd_multi {}

extern "C" {
    pub fn rtamd_ctx_create(device: c_int, out: *mut *mut rtamd_ctx) -> c_int;
    pub fn rtamd_ctx_destroy(ctx: *mut rtamd_ctx);
    pub fn rtamd_last_error(ctx: *const rtamd_ctx) -> *const c_char;
    pub fn rtamd_scene_upload(ctx: *mut rtamd_ctx, scene: *const rtamd_scene) -> c_int;
    pub fn rtamd_part_rows(ctx: *const rtamd_ctx, rows: *const rtamd_rows) -> i64;
    pub fn rtamd_prepare_rows(ctx: *mut rtamd_ctx, rows: *const rtamd_rows, mode: c_int) -> c_int;
    pub fn rtamd_render_rows(ctx: *mut rtamd_ctx, rows: *const rtamd_rows, mode: c_int, d_rgb: *mut f64, d_rgb8: *mut u8, stream: *mut c_void) -> c_int;
    /// allocates nothing (callers under HIP graph capture, serving loops)
    pub fn rtamd_render_rows_static(ctx: *mut rtamd_ctx, rows: *const rtamd_rows, mode: c_int, d_rgb: *mut f64, d_rgb8: *mut u8, stream: *mut c_void) -> c_int;
    pub fn rtamd_render(ctx: *mut rtamd_ctx, mode: c_int, h_rgb: *mut f64, h_rgb8: *mut u8) -> c_int;
    pub fn rtamd_host_alloc(bytes: usize) -> *mut c_void;
    pub fn rtamd_host_free(p: *mut c_void);
    pub fn rtamd_color_at(ctx: *mut rtamd_ctx, mode: c_int, h_rays: *const f64, n: i64, remaining: c_int, h_rgb: *mut f64) -> c_int;
    pub fn rtamd_count_work(ctx: *mut rtamd_ctx, on: c_int) -> c_int;
    pub fn rtamd_get_stats(ctx: *mut rtamd_ctx, out: *mut rtamd_stats) -> c_int;
    pub fn rtamd_reset_stats(ctx: *mut rtamd_ctx) -> c_int;
    // several GPUs, one frame
    pub fn rtamd_rows_layout(vsize: c_int, rows: *const rtamd_rows, out: *mut rtamd_layout) -> c_int;
    pub fn rtamd_multi_create(devices: *const c_int, n_devices: c_int, out: *mut *mut rtamd_multi) -> c_int;
    pub fn rtamd_multi_destroy(m: *mut rtamd_multi);
    pub fn rtamd_multi_last_error(m: *const rtamd_multi) -> *const c_char;
    pub fn rtamd_multi_upload(m: *mut rtamd_multi, scene: *const rtamd_scene) -> c_int;
    pub fn rtamd_multi_render(m: *mut rtamd_multi, mode: c_int, tile_rows: c_int, first_parts: c_int, other_parts: c_int, h_rgb: *mut f64, h_rgb8: *mut u8, want: c_int) -> c_int;
    pub fn rtamd_multi_frame(m: *mut rtamd_multi, d_rgb: *mut *mut f64, d_rgb8: *mut *mut u8) -> c_int;
    pub fn rtamd_multi_get_stats(m: *mut rtamd_multi, total: *mut rtamd_stats, render_ms: *mut f64, place_ms: *mut f64) -> c_int;
    pub fn rtamd_multi_reset_stats(m: *mut rtamd_multi) -> c_int;
}

// ---- include/rtamd_host.h
#[repr(C)]
pub struct rth_material {
    pub color: [f64; 3],
    pub ambient: f64,
    pub diffuse: f64,
    pub specular: f64,
    pub shininess: f64,
    pub reflective: f64,
    pub transparency: f64,
    pub refractive_index: f64,
    pub pattern: i32,
    pub pad_: i32,
    pub pat_a: [f64; 3],
    pub pat_b: [f64; 3],
    pub pat_transform: [f64; 16],
}
pub enum rth_world {}

extern "C" {
    pub fn rth_last_error() -> *const c_char;
    pub fn rth_material_default(out: *mut rth_material);
    pub fn rth_world_new() -> *mut rth_world;
    pub fn rth_world_free(w: *mut rth_world);
    // transform: 16 doubles row-major = &Matrix4x4 (src/matrix.rs:68), may be null for the identity;
    // material may be null for Material::new()
    pub fn rth_add_sphere(w: *mut rth_world, transform: *const f64, material: *const rth_material, shadow: c_int) -> c_int;
    pub fn rth_add_plane(w: *mut rth_world, transform: *const f64, material: *const rth_material, shadow: c_int) -> c_int;
    pub fn rth_add_cube(w: *mut rth_world, transform: *const f64, material: *const rth_material, shadow: c_int) -> c_int;
    pub fn rth_add_cylinder(w: *mut rth_world, transform: *const f64, material: *const rth_material, shadow: c_int, min: c_double, max: c_double, closed: c_int) -> c_int;
    pub fn rth_add_cone(w: *mut rth_world, transform: *const f64, material: *const rth_material, shadow: c_int, min: c_double, max: c_double, closed: c_int) -> c_int;
    /// points: &[p1, p2, p3] of Tuples (12 doubles)
    pub fn rth_add_triangle(w: *mut rth_world, transform: *const f64, material: *const rth_material, shadow: c_int, points: *const f64) -> c_int;
    pub fn rth_add_smooth_triangle(w: *mut rth_world, transform: *const f64, material: *const rth_material, shadow: c_int, points: *const f64, normals: *const f64) -> c_int;
    pub fn rth_group_begin(w: *mut rth_world, transform: *const f64, material: *const rth_material, shadow: c_int) -> c_int;
    pub fn rth_group_end(w: *mut rth_world) -> c_int;
    pub fn rth_csg_begin(w: *mut rth_world, op: c_int, transform: *const f64, material: *const rth_material, shadow: c_int) -> c_int;
    pub fn rth_csg_end(w: *mut rth_world) -> c_int;
    pub fn rth_add_light(w: *mut rth_world, position: *const f64, intensity: *const f64) -> c_int;
    pub fn rth_set_camera(w: *mut rth_world, hsize: c_int, vsize: c_int, field_of_view: c_double, transform: *const f64) -> c_int;
    pub fn rth_world_scene(w: *mut rth_world) -> *mut c_void;
    // scenes from files
    pub fn rth_load_yaml(path: *const c_char) -> *mut c_void;
    pub fn rth_load_yaml_str(text: *const c_char) -> *mut c_void;
    pub fn rth_load_sg(path: *const c_char) -> *mut c_void;
    pub fn rth_builtin_scene(name: *const c_char, arg: *const c_char) -> *mut c_void;
    pub fn rth_save_sg(scene: *mut c_void, path: *const c_char) -> c_int;
    pub fn rth_free_scene(scene: *mut c_void);
    pub fn rth_scene_info(scene: *mut c_void, hsize: *mut c_int, vsize: *mut c_int, n_objects: *mut c_int, n_lights: *mut c_int) -> c_int;
    pub fn rth_set_resolution(scene: *mut c_void, hsize: c_int, vsize: c_int) -> c_int;
    // World + Camera -> rtamd_scene
    pub fn rth_flatten(scene: *mut c_void, max_remaining: c_int) -> *mut c_void;
    pub fn rth_flat_view(flat: *mut c_void) -> *const rtamd_scene;
    pub fn rth_free_flat(flat: *mut c_void);
    pub fn rth_render(scene: *mut c_void, mode: c_int, rgb: *mut f64, rgb8: *mut u8, stats: *mut rtamd_stats) -> c_int;
    // Canvas output
    pub fn rth_canvas_save(rgb: *const f64, w: c_int, h: c_int, path: *const c_char, as_ppm: c_int) -> c_int;
    pub fn rth_canvas_quantize(rgb: *const f64, n_pixels: i64, out: *mut u8) -> c_int;
    pub fn rth_save_png8(rgb8: *const u8, w: c_int, h: c_int, path: *const c_char) -> c_int;
    pub fn rth_canvas_ppm(rgb: *const f64, w: c_int, h: c_int) -> *mut c_char;
    pub fn rth_free_str(s: *mut c_char);
    pub fn rth_obj_info(text: *const c_char, n_vertices: *mut c_int, n_normals: *mut c_int, n_default_tris: *mut c_int, n_groups: *mut c_int,
                        n_group_tris: *mut c_int, lines_ignored: *mut c_int, n_smooth: *mut c_int) -> c_int;
    pub fn rth_obj_triangle(text: *const c_char, k: c_int, out21: *mut f64) -> c_int;
}
