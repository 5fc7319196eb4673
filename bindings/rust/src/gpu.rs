//! Safe wrapper over `ffi.rs` and the flattening shim the reference needs.
//!
//! UNCOMPILED (no rustc/cargo in the build image).  See bindings/rust/PATCH.md for where each piece goes in
//! the reference tree; the struct layouts and the function list are checked against include/rtiow_hip.h by
//! tests/test_rust_binding.py.
//!
//! Why a shim: `Sphere { center, radius, mat }` (src/shapes/sphere.rs:9-13), `Lambertian { albedo }`,
//! `Metal { albedo, fuzz }`, `Dialectric { ir }` (src/materials.rs:9-11,34-37,64-66) keep their fields
//! private, and the world is a `Vec<Box<dyn Hit>>` (src/shapes/mod.rs:52) holding `Arc<dyn Scatter>`s: a GPU
//! backend cannot read the scene back through `Hit`/`Scatter`.  Rust privacy is per module, so the two
//! traits each gain ONE method, implemented next to the private fields:
//!
//!   trait Scatter { ...; fn describe(&self) -> MaterialDesc; }            // src/materials.rs:5-7
//!   trait Hit     { ...; fn describe(&self) -> Option<rt_sphere> { None } }   // src/shapes/mod.rs:48-50
//!
//! and `flatten(&world)` walks the list in push order (order is part of the input: on equal t the later
//! sphere wins, src/shapes/mod.rs:61-67).

use std::ffi::CStr;

use crate::camera::Camera;
use crate::ffi::*;
use crate::shapes::{Hit, HittableList};

/// What a material tells the backend about itself (the arguments of its `new`).
#[derive(Clone, Copy, Debug, PartialEq)]
pub struct MaterialDesc {
    pub kind: i32,          // RT_LAMBERTIAN / RT_METAL / RT_DIALECTRIC
    pub albedo: [f64; 3],   // Lambertian, Metal; (1,1,1) for Dialectric (its attenuation, materials.rs:103)
    pub param: f64,         // Metal: fuzz AS STORED (Metal::new, materials.rs:39-46, does not clamp it); Dialectric: ir
}

// ---- the impls that go INTO the reference's modules (they read private fields) -----------------------------
//
// src/materials.rs:
//   impl Scatter for Lambertian { fn describe(&self) -> MaterialDesc {
//       MaterialDesc { kind: RT_LAMBERTIAN, albedo: [self.albedo.x(), self.albedo.y(), self.albedo.z()], param: 0.0 } } }
//   impl Scatter for Metal { fn describe(&self) -> MaterialDesc {
//       MaterialDesc { kind: RT_METAL, albedo: [self.albedo.x(), self.albedo.y(), self.albedo.z()], param: self.fuzz } } }
//   impl Scatter for Dialectric { fn describe(&self) -> MaterialDesc {
//       MaterialDesc { kind: RT_DIALECTRIC, albedo: [1.0, 1.0, 1.0], param: self.ir } } }
//
// src/shapes/sphere.rs:
//   impl Hit for Sphere { fn describe(&self) -> Option<rt_sphere> {
//       let m = self.mat.describe();
//       Some(rt_sphere { center: [self.center.x(), self.center.y(), self.center.z()], radius: self.radius,
//                        albedo: m.albedo, param: m.param, kind: m.kind, reserved: 0 }) } }
//
// src/camera.rs (Camera's fields are private too, camera.rs:4-13):
//   impl Camera { pub fn to_rt(&self) -> rt_camera { rt_camera {
//       origin: v3(self.origin), lower_left_corner: v3(self.lower_left_corner), horizontal: v3(self.horizontal),
//       vertical: v3(self.vertical), u: v3(self.u), v: v3(self.v), lens_radius: self.lens_radius } } }
//   fn v3(a: Vec3) -> [f64; 3] { [a.x(), a.y(), a.z()] }

/// The flat scene in push order.  Fails on an object that does not describe itself (not a sphere).
pub fn flatten(world: &HittableList) -> Result<Vec<rt_sphere>, String> {
    world
        .iter()
        .enumerate()
        .map(|(i, obj)| obj.describe().ok_or_else(|| format!("object {i} of the world cannot be flattened")))
        .collect()
}

fn last_error() -> String {
    unsafe { CStr::from_ptr(rt_last_error()).to_string_lossy().into_owned() }
}

fn check(rc: i32, what: &str) -> Result<(), String> {
    if rc == RT_OK { Ok(()) } else { Err(format!("{what} failed ({rc}): {}", last_error())) }
}

/// One `rt_context` = one GPU.  Not `Sync`: a context is used from one thread at a time
/// (include/rtiow_hip.h, rules of the boundary); distinct contexts may be used concurrently.
pub struct GpuRenderer {
    ctx: *mut rt_context,
}

unsafe impl Send for GpuRenderer {}

impl GpuRenderer {
    pub fn new(device_id: i32) -> Result<Self, String> {
        let mut ctx = std::ptr::null_mut();
        check(unsafe { rt_create(device_id, &mut ctx) }, "rt_create")?;
        Ok(GpuRenderer { ctx })
    }

    /// Stands in for the capture of `&world` at src/main.rs:135.
    pub fn upload_world(&mut self, world: &HittableList) -> Result<(), String> {
        let flat = flatten(world)?;
        check(unsafe { rt_upload_scene(self.ctx, flat.as_ptr(), flat.len() as i32) }, "rt_upload_scene")
    }

    /// src/main.rs:122-136 (everything up to, not including, `to_rgba`): exact radiance sums, u64 with
    /// quantum 2^-32, `[rows][width][3]`, rows ascending from the BOTTOM of the image like `j` at main.rs:122.
    pub fn render(&mut self, cam: &Camera, p: &rt_params) -> Result<(Vec<u64>, rt_stats), String> {
        let mut rows = 0i32;
        check(unsafe { rt_shard_rows(p, &mut rows) }, "rt_shard_rows")?;
        let mut fix = vec![0u64; rows as usize * p.width as usize * 3];
        let mut stats: rt_stats = unsafe { std::mem::zeroed() };
        let rc = unsafe { rt_render(self.ctx, &cam.to_rt(), p, std::ptr::null_mut(), fix.as_mut_ptr(), &mut stats) };
        check(rc, "rt_render")?;
        Ok((fix, stats))
    }

    /// `pixel_color.to_rgba(255, SAMPLES_PER_PIXEL)` src/main.rs:137 for every pixel + the flip of
    /// src/main.rs:141-145 (flip = true gives the top row first, the order `ImageBuffer::from_vec` wants).
    pub fn resolve_rgba8(&mut self, fix: &[u64], width: i32, rows: i32, spp: i64, flip: bool) -> Result<Vec<u8>, String> {
        let mut out = vec![0u8; width as usize * rows as usize * 4];
        let rc = unsafe { rt_resolve_rgba8(self.ctx, fix.as_ptr(), width, rows, spp, flip as i32, out.as_mut_ptr()) };
        check(rc, "rt_resolve_rgba8")?;
        Ok(out)
    }

    /// src/main.rs:122-145 in ONE call: render, `to_rgba` and the flip all on the device; only the RGBA8 bytes
    /// (4 per pixel) cross PCIe.  Same bytes as `render` + `resolve_rgba8`, which move the 24-byte sums out and in again.
    pub fn render_rgba8(&mut self, cam: &Camera, p: &rt_params, flip: bool) -> Result<(Vec<u8>, rt_stats), String> {
        let mut rows = 0i32;
        check(unsafe { rt_shard_rows(p, &mut rows) }, "rt_shard_rows")?;
        let mut out = vec![0u8; rows as usize * p.width as usize * 4];
        let mut stats: rt_stats = unsafe { std::mem::zeroed() };
        let rc = unsafe { rt_render_rgba8(self.ctx, &cam.to_rt(), p, flip as i32, out.as_mut_ptr(), &mut stats) };
        check(rc, "rt_render_rgba8")?;
        Ok((out, stats))
    }
}

impl Drop for GpuRenderer {
    fn drop(&mut self) {
        unsafe { rt_destroy(self.ctx) };
    }
}

/// The replacement of src/main.rs:122-145: returns the bytes `image_buffer` is built from at main.rs:147.
pub fn render_image(world: &HittableList, cam: &Camera, width: u32, height: u32, spp: u64, max_depth: i32,
                    seed: u64) -> Result<Vec<u8>, String> {
    let mut gpu = GpuRenderer::new(0)?;
    gpu.upload_world(world)?;
    let p = rt_params { width: width as i32, height: height as i32, spp: spp as i32, sample_begin: 0, max_depth,
                        t_min: 0.0001,                  // main.rs:44
                        seed, tile_rows: 8, shard_index: 0, shard_count: 1, flags: 0 };
    let (pixels, _stats) = gpu.render_rgba8(cam, &p, true)?;      // (flip: top row first, main.rs:141-145)
    Ok(pixels)
}
