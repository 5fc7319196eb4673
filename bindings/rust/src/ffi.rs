//! Raw FFI of `librtiow_hip.so` (C ABI: `include/rtiow_hip.h`, ABI version 5).
//!
//! UNCOMPILED: there is no rustc/cargo in the build image of this repository, so this file has never
//! been through a Rust compiler.  What IS checked (tests/test_rust_binding.py, CPU): every `#[repr(C)]`
//! struct below has the same fields, in the same order and of the same C type, as its counterpart in
//! include/rtiow_hip.h; the `const` assertions state the sizes and offsets the C side is tested for
//! (tests/test_cabi.py); and the `extern "C"` block declares exactly the functions the header declares.
//!
//! What the entry points stand in for in the reference (Druthyn/rtiow): the iterator expression at
//! src/main.rs:122-139 (`rt_render`), `Color::to_rgba` src/vec3.rs:403-421 + the flip src/main.rs:141-145
//! (`rt_resolve_rgba8`), both in one call with the sums kept on the device (`rt_render_rgba8`: the bytes
//! `ImageBuffer::from_vec` takes at src/main.rs:147), the capture of `&world` at src/main.rs:135 (`rt_upload_scene`).
#![allow(non_camel_case_types, dead_code)]

use core::mem::{offset_of, size_of};
use std::os::raw::{c_char, c_void};

pub const RTIOW_HIP_ABI_VERSION: i32 = 5;

pub const RT_OK: i32 = 0;
pub const RT_ERR_INVALID_ARGUMENT: i32 = -1;
pub const RT_ERR_NO_DEVICE: i32 = -2;
pub const RT_ERR_HIP: i32 = -3;
pub const RT_ERR_NO_SCENE: i32 = -4;
pub const RT_ERR_OUT_OF_MEMORY: i32 = -5;

/// Material kinds: the three `impl Scatter` of src/materials.rs.
pub const RT_LAMBERTIAN: i32 = 0;
pub const RT_METAL: i32 = 1;
pub const RT_DIALECTRIC: i32 = 2;

pub const RT_FLAG_ACCUMULATE: u32 = 0x1;
pub const RT_FLAG_NO_FILTER: u32 = 0x2;
pub const RT_FLAG_DIAG_STATS: u32 = 0x4;
pub const RT_FLAG_UNIFORM53: u32 = 0x8;
pub const RT_FLAG_OVERLAPPED: u32 = 0x10;
pub const RT_FLAG_KNOWN: u32 = 0x1f;

/// The two limits of the boundary where the reference's own types are unbounded (include/rtiow_hip.h): the length of
/// `HittableList` (src/shapes/mod.rs:52) and the pixel sum (src/main.rs:127,135: here exact u64 sums of samples clamped at 2^16).
pub const RT_MAX_SPHERES: i32 = 1 << 24;
pub const RT_SAMPLE_CLAMP: f64 = 65536.0;

/// Opaque `rt_context`.
#[repr(C)]
pub struct rt_context {
    _private: [u8; 0],
}

/// One sphere, flattened (src/shapes/sphere.rs:9-13 + src/materials.rs:9-11,34-37,64-66).  LIST ORDER IS
/// PART OF THE INPUT: on equal t the later sphere wins (src/shapes/mod.rs:61-67).
#[repr(C)]
#[derive(Clone, Copy, Debug, PartialEq)]
pub struct rt_sphere {
    pub center: [f64; 3],
    pub radius: f64,
    pub albedo: [f64; 3],
    pub param: f64,
    pub kind: i32,
    pub reserved: i32,
}

/// src/camera.rs:4-13 minus `w` (never read by `get_ray`).
#[repr(C)]
#[derive(Clone, Copy, Debug, PartialEq)]
pub struct rt_camera {
    pub origin: [f64; 3],
    pub lower_left_corner: [f64; 3],
    pub horizontal: [f64; 3],
    pub vertical: [f64; 3],
    pub u: [f64; 3],
    pub v: [f64; 3],
    pub lens_radius: f64,
}

/// What src/main.rs:24-28,44 fixes at compile time, plus sharding.
#[repr(C)]
#[derive(Clone, Copy, Debug, PartialEq)]
pub struct rt_params {
    pub width: i32,
    pub height: i32,
    pub spp: i32,
    pub sample_begin: i32,
    pub max_depth: i32,
    pub t_min: f64,
    pub seed: u64,
    pub tile_rows: i32,
    pub shard_index: i32,
    pub shard_count: i32,
    pub flags: u32,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct rt_stats {
    pub samples: u64,
    pub rays_traced: u64,
    pub sphere_tests: u64,
    pub candidates: u64,
    pub exact_roots: u64,
    pub kernel_ms: f32,
    pub n_spheres: i32,
    pub grid_blocks: i32,
    pub block_threads: i32,
    pub scan_mode: i32,
    pub kernel_variant: i32,
    pub live_per_bounce: [u64; 64],
    pub direct_samples: u64,
}

// Layout assertions: the numbers tests/test_cabi.py asserts on the C side (ctypes mirrors of the header).
const _: () = assert!(size_of::<rt_sphere>() == 72);
const _: () = assert!(offset_of!(rt_sphere, kind) == 64);
const _: () = assert!(size_of::<rt_camera>() == 152);
const _: () = assert!(size_of::<rt_params>() == 56);
const _: () = assert!(offset_of!(rt_params, t_min) == 24);
const _: () = assert!(offset_of!(rt_params, seed) == 32);
const _: () = assert!(size_of::<rt_stats>() == 584);
const _: () = assert!(offset_of!(rt_stats, live_per_bounce) == 64);

#[link(name = "rtiow_hip")]
extern "C" {
    pub fn rt_create(device_id: i32, out: *mut *mut rt_context) -> i32;
    pub fn rt_destroy(ctx: *mut rt_context) -> i32;
    pub fn rt_upload_scene(ctx: *mut rt_context, spheres: *const rt_sphere, n: i32) -> i32;
    pub fn rt_shard_rows(p: *const rt_params, out_rows: *mut i32) -> i32;
    pub fn rt_shard_row_index(p: *const rt_params, compact_row: i32, out_j: *mut i32) -> i32;
    pub fn rt_render(ctx: *mut rt_context, cam: *const rt_camera, p: *const rt_params,
                     out_sum: *mut f32, out_fix: *mut u64, stats: *mut rt_stats) -> i32;
    pub fn rt_render_device(ctx: *mut rt_context, cam: *const rt_camera, p: *const rt_params,
                            d_fix: *mut c_void, stream: *mut c_void) -> i32;
    pub fn rt_fix_to_f32_device(ctx: *mut rt_context, d_fix: *const c_void, count: i64,
                                d_out_f32: *mut c_void, stream: *mut c_void) -> i32;
    pub fn rt_last_stats(ctx: *mut rt_context, stats: *mut rt_stats) -> i32;
    pub fn rt_resolve_rgba8_device(ctx: *mut rt_context, d_fix: *const c_void, width: i32, rows: i32,
                                   spp: i64, flip: i32, d_rgba: *mut c_void, stream: *mut c_void) -> i32;
    pub fn rt_resolve_rgba8(ctx: *mut rt_context, fix: *const u64, width: i32, rows: i32,
                            spp: i64, flip: i32, out_rgba: *mut u8) -> i32;
    pub fn rt_render_rgba8(ctx: *mut rt_context, cam: *const rt_camera, p: *const rt_params, flip: i32,
                           out_rgba: *mut u8, stats: *mut rt_stats) -> i32;
    pub fn rt_last_error() -> *const c_char;
    pub fn rt_backend_name() -> *const c_char;
    pub fn rt_abi_version() -> i32;
    pub fn rt_build_source_sha() -> *const c_char;
    pub fn rt_f64_div_sqrt_device(ctx: *mut rt_context, a: *const f64, b: *const f64, n: i32,
                                  out_div: *mut f64, out_sqrt: *mut f64) -> i32;
    pub fn rt_quantize_device(ctx: *mut rt_context, x: *const f64, n: i32, out: *mut u64) -> i32;
    pub fn rt_unit_accept_device(ctx: *mut rt_context, words: *const u32, n: i32, out_accept: *mut u32,
                                 out_uniforms: *mut f64) -> i32;
    pub fn rt_filter_tube_device(ctx: *mut rt_context, o: *const f64, d: *const f64, spheres32: *const rt_sphere,
                                 out_h: *mut f32, out_rows: *mut f32, out_bound: *mut f32, out_rho: *mut f32) -> i32;
    pub fn rt_tube_tile_host(spheres32: *const rt_sphere, out_words: *mut u32, out_bound: *mut f32,
                             out_rho: *mut f32) -> i32;
    pub fn rt_tile_layout_host(spheres: *const rt_sphere, n: i32, out_dims: *mut i32, out_grid: *mut f32,
                               out_slot_of: *mut i32, cap: i32) -> i32;
    pub fn rt_philox_device(ctx: *mut rt_context, ctr: *const u32, key: *const u32, out: *mut u32) -> i32;
}
