//! Binding of `libptrt.so` (C ABI in `include/ptrt.h`) and a `Renderer` with the method names of the reference's
//! `src/renderer.rs` (`new`, `render`, `progressive_rendering.reset_ready_samples`), so that `App` keeps calling
//! what it calls today while the compute pass runs as HIP kernels on gfx950.
//!
//! The reference's `#[repr(C)]` / `bytemuck::Pod` structs (`CameraBuffer` 64 B, `Settings` 16 B,
//! `SphereDataBuffer` 8208 B, `TriangleBuffer` 112 B, `Node` 32 B) are passed as bytes, unchanged.

use std::ffi::CStr;
use std::os::raw::{c_char, c_void};
use std::time::Instant;

/// Opaque renderer context (`pt_ctx`).
#[repr(C)]
pub struct PtCtx {
    _private: [u8; 0],
}

/// Opaque group of renderer contexts, one per GPU (`pt_group`).
#[repr(C)]
pub struct PtGroup {
    _private: [u8; 0],
}

/// `pt_stats_t`
#[repr(C)]
#[derive(Default, Debug, Clone, Copy, PartialEq, Eq)]
pub struct PtStats {
    pub rays: u64,
    pub paths: u64,
    pub interior: u64,
    pub tri_tests: u64,
    pub tri_accept: u64,
    pub sphere_tests: u64,
    pub sky_lookups: u64,
    pub bounces: u64,
    pub sphere_accept: u64,
    pub frames: u64,
    pub bvh_depth: u64,
    pub walks: u64,
    pub sky_loop_beside: u64,
    pub guarded: u64,
    pub rewalks: u64,
    pub guard_nodes: u64,
}

pub const PT_FLAG_REFERENCE_DISPATCH: u32 = 1;
pub const PT_FLAG_AOV: u32 = 2;
pub const PT_FLAG_DETAILED_STATS: u32 = 4;
pub const PT_FLAG_TILE_KERNEL: u32 = 8;
pub const PT_FLAG_POOL_KERNEL: u32 = 16;
pub const PT_FLAG_TRIANGLE_MATERIALS: u32 = 32;
pub const PT_FLAG_REGROUP_SHORT: u32 = 64;
pub const PT_FLAG_REBUILD_BVH: u32 = 128;
pub const PT_FLAG_REFERENCE_WALK: u32 = 256;
pub const PT_GROUP_RCCL: u32 = 1;
pub const PT_PRIM_NONE: u32 = 0xFFFF_FFFF;
pub const PT_PRIM_SPHERE: u32 = 0x8000_0000;

// One line per function of include/ptrt.h, same order, same arity (tests/test_bindings.py checks both).
extern "C" {
    pub fn pt_create(device: i32, width: u32, height: u32, out: *mut *mut PtCtx) -> i32;
    pub fn pt_destroy(ctx: *mut PtCtx);
    pub fn pt_last_error(ctx: *const PtCtx) -> *const c_char;
    pub fn pt_set_stream(ctx: *mut PtCtx, hip_stream: *mut c_void) -> i32;
    pub fn pt_set_flags(ctx: *mut PtCtx, flags: u32) -> i32;
    pub fn pt_set_mesh(ctx: *mut PtCtx, tris: *const c_void, n_tris: u32, nodes: *const c_void, n_nodes: u32, tri_idx: *const u32, n_idx: u32) -> i32;
    pub fn pt_set_sky_cube(ctx: *mut PtCtx, rgba32f: *const f32, size: u32) -> i32;
    pub fn pt_set_sky_equirect(ctx: *mut PtCtx, rgba32f: *const f32, w: u32, h: u32, cube_size: u32) -> i32;
    pub fn pt_read_sky_cube(ctx: *mut PtCtx, out: *mut f32, capacity_floats: usize, size: *mut u32) -> i32;
    pub fn pt_set_camera(ctx: *mut PtCtx, camera64: *const c_void) -> i32;
    pub fn pt_set_spheres(ctx: *mut PtCtx, sphere_data8208: *const c_void) -> i32;
    pub fn pt_set_settings(ctx: *mut PtCtx, settings16: *const c_void) -> i32;
    pub fn pt_render(ctx: *mut PtCtx, time: u32) -> i32;
    pub fn pt_render_frames(ctx: *mut PtCtx, time0: u32, stride: u32, n_frames: u32) -> i32;
    pub fn pt_render_frames_static(ctx: *mut PtCtx, time0: u32, stride: u32, n_frames: u32) -> i32;
    pub fn pt_reset(ctx: *mut PtCtx) -> i32;
    pub fn pt_sync(ctx: *mut PtCtx) -> i32;
    pub fn pt_read_sum_rgba32f(ctx: *mut PtCtx, out: *mut f32) -> i32;
    pub fn pt_read_mean_rgba32f(ctx: *mut PtCtx, out: *mut f32) -> i32;
    pub fn pt_read_rgba8(ctx: *mut PtCtx, out: *mut u8) -> i32;
    pub fn pt_set_window(ctx: *mut PtCtx, sample_size: u32) -> i32;
    pub fn pt_read_window(ctx: *mut PtCtx, rgba32f_out: *mut f32, rgba8_out: *mut u8, frames_shown: *mut u32) -> i32;
    pub fn pt_read_aov(ctx: *mut PtCtx, prim: *mut u32, t: *mut f32) -> i32;
    pub fn pt_read_sky_tiles(ctx: *mut PtCtx, out: *mut u32, capacity: usize, n_tiles: *mut u32) -> i32;
    pub fn pt_stats(ctx: *mut PtCtx, out: *mut PtStats) -> i32;
    pub fn pt_accum_device_ptr(ctx: *mut PtCtx) -> *mut c_void;
    pub fn pt_bind_accum(ctx: *mut PtCtx, device_ptr: *mut c_void) -> i32;
    pub fn pt_trace_rays(ctx: *mut PtCtx, origins: *const f32, dirs: *const f32, n: u32, prim: *mut u32, t: *mut f32, p: *mut f32, normal: *mut f32, front: *mut u32) -> i32;
    pub fn pt_trace_rays_walked(ctx: *mut PtCtx, origins: *const f32, dirs: *const f32, n: u32, prim: *mut u32, t: *mut f32, p: *mut f32, normal: *mut f32, front: *mut u32, walked: *mut u32) -> i32;
    pub fn pt_group_create(devices: *const i32, n: i32, width: u32, height: u32, flags: u32, out: *mut *mut PtGroup) -> i32;
    pub fn pt_group_destroy(g: *mut PtGroup);
    pub fn pt_group_last_error(g: *const PtGroup) -> *const c_char;
    pub fn pt_group_size(g: *const PtGroup) -> i32;
    pub fn pt_group_member(g: *mut PtGroup, i: i32) -> *mut PtCtx;
    pub fn pt_group_set_flags(g: *mut PtGroup, flags: u32) -> i32;
    pub fn pt_group_set_mesh(g: *mut PtGroup, tris: *const c_void, n_tris: u32, nodes: *const c_void, n_nodes: u32, tri_idx: *const u32, n_idx: u32) -> i32;
    pub fn pt_group_set_sky_cube(g: *mut PtGroup, rgba32f: *const f32, size: u32) -> i32;
    pub fn pt_group_set_sky_equirect(g: *mut PtGroup, rgba32f: *const f32, w: u32, h: u32, cube_size: u32) -> i32;
    pub fn pt_group_set_camera(g: *mut PtGroup, camera64: *const c_void) -> i32;
    pub fn pt_group_set_spheres(g: *mut PtGroup, sphere_data8208: *const c_void) -> i32;
    pub fn pt_group_set_settings(g: *mut PtGroup, settings16: *const c_void) -> i32;
    pub fn pt_group_render_frames(g: *mut PtGroup, time0: u32, n_frames: u32) -> i32;
    pub fn pt_group_reduce(g: *mut PtGroup, root: i32) -> i32;
    pub fn pt_group_reset(g: *mut PtGroup) -> i32;
    pub fn pt_group_sync(g: *mut PtGroup) -> i32;
    pub fn pt_group_stats(g: *mut PtGroup, out: *mut PtStats) -> i32;
    pub fn pt_bvh_build(tris: *const c_void, n_tris: u32, nodes_out: *mut c_void, idx_out: *mut u32) -> i64;
    pub fn pt_bvh_build_sah(tris: *const c_void, n_tris: u32, nodes_out: *mut c_void, idx_out: *mut u32) -> i64;
    pub fn pt_bvh_build_sah_padded(tris: *const c_void, n_tris: u32, pad: *const f32, nodes_out: *mut c_void, idx_out: *mut u32) -> i64;
    pub fn pt_plane_triangles(q: *const f32, u: *const f32, v: *const f32, albedo: *const f32, material: u32, tris_out: *mut c_void) -> i32;
    pub fn pt_pack_spheres(spheres8: *const f32, n: u32, sphere_data8208_out: *mut c_void) -> i32;
    pub fn pt_obj_load(path: *const c_char, tris_out: *mut c_void, capacity: u32, n_tris: *mut u32) -> i32;
    pub fn pt_hdr_decode(data: *const c_void, n_bytes: usize, rgba_out: *mut f32, capacity_pixels: usize, width: *mut u32, height: *mut u32) -> i32;
    pub fn pt_host_error() -> *const c_char;
}

/// A failed call: the library's status code and message.
#[derive(Debug, Clone)]
pub struct PtError {
    pub code: i32,
    pub message: String,
}

impl std::fmt::Display for PtError {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result {
        write!(f, "ptrt error {}: {}", self.code, self.message)
    }
}
impl std::error::Error for PtError {}

fn message(p: *const c_char) -> String {
    if p.is_null() {
        String::new()
    } else {
        unsafe { CStr::from_ptr(p) }.to_string_lossy().into_owned()
    }
}

/// Turns a status of a context call into a `Result` (message from `pt_last_error`).
fn ok(code: i32, ctx: *const PtCtx) -> Result<(), PtError> {
    if code == 0 {
        Ok(())
    } else {
        Err(PtError { code, message: message(unsafe { pt_last_error(ctx) }) })
    }
}

/// Same for the host-side helpers that take no context (message from `pt_host_error`).
fn ok_host(code: i64) -> Result<i64, PtError> {
    if code >= 0 {
        Ok(code)
    } else {
        Err(PtError { code: code as i32, message: message(unsafe { pt_host_error() }) })
    }
}

/// `Settings` of `src/renderer.rs:604-611`, defaults of `:411-416`.
#[repr(C)]
#[derive(Debug, Clone, Copy, bytemuck::Pod, bytemuck::Zeroable)]
pub struct Settings {
    pub samples_per_pixel: u32,
    pub depth: u32,
    pub t_min: f32,
    pub t_max: f32,
}

impl Default for Settings {
    fn default() -> Self {
        Settings { samples_per_pixel: 1, depth: 32, t_min: 0.0001, t_max: 1000.0 }
    }
}

/// The part of `ProgressiveRendering` (`src/renderer.rs:613-646`) the host still reads; the frames themselves are
/// accumulated on the GPU.
pub struct ProgressiveRendering {
    ctx: *mut PtCtx,
}

impl ProgressiveRendering {
    /// `reset_ready_samples` (`renderer.rs:635-637`): called by the UI and the scene on every change.
    pub fn reset_ready_samples(&mut self) {
        // pt_reset only queues two memsets; it fails only if the device is lost, which render() reports
        let _ = unsafe { pt_reset(self.ctx) };
    }
}

/// Decoded Radiance panorama, as `CubeTexture::from_equirectangular_hdri` (`src/texture.rs:179-188`) gets it from
/// the `image` crate: width x height RGBA32F, alpha 1.
pub fn decode_hdr(file: &[u8]) -> Result<(Vec<f32>, u32, u32), PtError> {
    let (mut w, mut h) = (0u32, 0u32);
    ok_host(unsafe { pt_hdr_decode(file.as_ptr().cast(), file.len(), std::ptr::null_mut(), 0, &mut w, &mut h) } as i64)?;
    let mut pano = vec![0f32; w as usize * h as usize * 4];
    ok_host(unsafe { pt_hdr_decode(file.as_ptr().cast(), file.len(), pano.as_mut_ptr(), pano.len() / 4, &mut w, &mut h) } as i64)?;
    Ok((pano, w, h))
}

/// `Bvh::from_triangles` (`src/scene/bvh.rs:34-66`) on up to 16 host threads: same nodes, same index permutation.
/// `tris` are 112-byte `TriangleBuffer` records; returns (32-byte `Node` records as bytes, triangle indices).
pub fn bvh_build(tris: &[u8]) -> Result<(Vec<u8>, Vec<u32>), PtError> {
    assert!(tris.len() % 112 == 0);
    let n = (tris.len() / 112) as u32;
    if n == 0 {
        return Ok((Vec::new(), Vec::new()));
    }
    let mut nodes = vec![0u8; (2 * n as usize - 1) * 32];
    let mut idx = vec![0u32; n as usize];
    let used = ok_host(unsafe { pt_bvh_build(tris.as_ptr().cast(), n, nodes.as_mut_ptr().cast(), idx.as_mut_ptr()) })?;
    nodes.truncate(used as usize * 32);
    Ok((nodes, idx))
}

/// Stand-in for `Renderer` of `src/renderer.rs:13-29`.  The wgpu device, queue, pipelines, bind groups and the 256
/// output textures are gone; what stays is the life cycle: construct with the static scene, `render` once per frame
/// with the per-frame uniforms, `progressive_rendering.reset_ready_samples()` on any change.
pub struct Renderer {
    ctx: *mut PtCtx,
    width: u32,
    height: u32,
    pub settings: Settings,
    start_time: Instant,
    pub progressive_rendering: ProgressiveRendering,
}

impl Renderer {
    /// `Renderer::new(device, queue, surface_config, scene)` (`renderer.rs:32-37`).  `triangles` / `nodes` /
    /// `triangle_indices` are the bytes of the `create_buffer_init` uploads (`renderer.rs:220-241`):
    /// `bytemuck::cast_slice(&Vec<TriangleBuffer>)`, `bytemuck::cast_slice(&scene.bvh.nodes)`,
    /// `&scene.bvh.triangle_indices`; `hdr_file` is what `include_bytes!` handed to
    /// `CubeTexture::from_equirectangular_hdri(.., 4096)` (`renderer.rs:215-218`).
    pub fn new(width: u32, height: u32, triangles: &[u8], nodes: &[u8], triangle_indices: &[u32], hdr_file: &[u8]) -> Result<Self, PtError> {
        let mut ctx: *mut PtCtx = std::ptr::null_mut();
        ok(unsafe { pt_create(0, width, height, &mut ctx) }, std::ptr::null())?;
        let r = Renderer {
            ctx,
            width,
            height,
            settings: Settings::default(),
            start_time: Instant::now(),
            progressive_rendering: ProgressiveRendering { ctx },
        };
        ok(
            unsafe {
                pt_set_mesh(ctx, triangles.as_ptr().cast(), (triangles.len() / 112) as u32, nodes.as_ptr().cast(), (nodes.len() / 32) as u32,
                            triangle_indices.as_ptr(), triangle_indices.len() as u32)
            },
            ctx,
        )?;
        let (pano, w, h) = decode_hdr(hdr_file)?;
        ok(unsafe { pt_set_sky_equirect(ctx, pano.as_ptr(), w, h, 4096) }, ctx)?;
        Ok(r)
    }

    /// The reference's on-screen picture instead of the fp32 running mean: a window of `sample_size` (1..=256)
    /// rgba8unorm frames averaged as `copy.wgsl:27-37` does (`renderer.rs:11, 418-425`); 0 turns it off.
    pub fn set_window(&mut self, sample_size: u32) -> Result<(), PtError> {
        ok(unsafe { pt_set_window(self.ctx, sample_size) }, self.ctx)
    }

    /// `Renderer::render(output, encoder, scene, queue)` (`renderer.rs:551-601`): the uniform uploads of
    /// `update_buffers` (`:518-548`, same bytes, same order), one dispatch of the compute pass (`:562-570`), and the
    /// picture the copy pass would draw (`:576-598`) as tightly packed RGBA8 into `out_rgba8`.
    /// `camera64` / `sphere_data8208` are `bytemuck::bytes_of(&CameraBuffer::from(&scene.camera))` and
    /// `bytemuck::bytes_of(&SphereDataBuffer::from(&scene.spheres))`; `moved_recently` is `scene.camera.moved_recently()`.
    pub fn render(&mut self, camera64: &[u8], sphere_data8208: &[u8], moved_recently: bool, out_rgba8: &mut [u8]) -> Result<(), PtError> {
        assert!(camera64.len() == 64 && sphere_data8208.len() == 8208);
        assert!(out_rgba8.len() >= self.width as usize * self.height as usize * 4);
        if moved_recently {
            self.progressive_rendering.reset_ready_samples(); // Renderer::update, renderer.rs:485-489
        }
        let time = (self.start_time.elapsed().as_millis() / 4) as u32; // renderer.rs:518-522: the shader reads the low 32 bits
        ok(unsafe { pt_set_camera(self.ctx, camera64.as_ptr().cast()) }, self.ctx)?;
        ok(unsafe { pt_set_spheres(self.ctx, sphere_data8208.as_ptr().cast()) }, self.ctx)?;
        ok(unsafe { pt_set_settings(self.ctx, bytemuck::bytes_of(&self.settings).as_ptr().cast()) }, self.ctx)?;
        ok(unsafe { pt_render(self.ctx, time) }, self.ctx)?;
        ok(unsafe { pt_read_rgba8(self.ctx, out_rgba8.as_mut_ptr()) }, self.ctx)
    }

    /// Offline rendering: `n_frames` dispatches with `time = time0 + i * stride` in one launch.
    pub fn render_frames(&mut self, time0: u32, stride: u32, n_frames: u32) -> Result<(), PtError> {
        ok(unsafe { pt_render_frames(self.ctx, time0, stride, n_frames) }, self.ctx)
    }

    /// The same without any device allocation (scheduling memory from the library's static 192 MiB array); fails with
    /// `PT_ERR_LIMIT` (-5) for launches that do not fit.
    pub fn render_frames_static(&mut self, time0: u32, stride: u32, n_frames: u32) -> Result<(), PtError> {
        ok(unsafe { pt_render_frames_static(self.ctx, time0, stride, n_frames) }, self.ctx)
    }

    /// fp32 mean of everything accumulated since the last reset (width * height * 4 floats).
    pub fn read_mean(&mut self) -> Result<Vec<f32>, PtError> {
        let mut out = vec![0f32; self.width as usize * self.height as usize * 4];
        ok(unsafe { pt_read_mean_rgba32f(self.ctx, out.as_mut_ptr()) }, self.ctx)?;
        Ok(out)
    }

    /// With a window set: the mean the copy pass shows after the last frame, as RGBA8, and how many frames it averaged.
    pub fn read_window(&mut self, out_rgba8: &mut [u8]) -> Result<u32, PtError> {
        assert!(out_rgba8.len() >= self.width as usize * self.height as usize * 4);
        let mut shown = 0u32;
        ok(unsafe { pt_read_window(self.ctx, std::ptr::null_mut(), out_rgba8.as_mut_ptr(), &mut shown) }, self.ctx)?;
        Ok(shown)
    }

    pub fn set_flags(&mut self, flags: u32) -> Result<(), PtError> {
        ok(unsafe { pt_set_flags(self.ctx, flags) }, self.ctx)
    }

    pub fn stats(&mut self) -> Result<PtStats, PtError> {
        let mut s = PtStats::default();
        ok(unsafe { pt_stats(self.ctx, &mut s) }, self.ctx)?;
        Ok(s)
    }

    /// Device address of the fp32 accumulation buffer (for an RCCL reduce across per-GPU renderers).
    pub fn accum_device_ptr(&mut self) -> *mut c_void {
        unsafe { pt_accum_device_ptr(self.ctx) }
    }
}

impl Drop for Renderer {
    fn drop(&mut self) {
        unsafe { pt_destroy(self.ctx) }
    }
}

/// Turns a status of a group call into a `Result` (message from `pt_group_last_error`).
fn ok_group(code: i32, g: *const PtGroup) -> Result<(), PtError> {
    if code == 0 {
        Ok(())
    } else {
        Err(PtError { code, message: message(unsafe { pt_group_last_error(g) }) })
    }
}

/// `Renderer` over all GPUs of a node, driven from this one thread: samples split across the GPUs by frame (frames are
/// independent given their `time`, `compute.wgsl:85`), the per-GPU sums added onto GPU `devices[0]`'s by `reduce`.
/// Same life cycle as `Renderer`: construct with the static scene, per-frame uniforms before rendering, reset on change.
pub struct RendererGroup {
    group: *mut PtGroup,
    width: u32,
    height: u32,
    pub settings: Settings,
}

impl RendererGroup {
    /// `rccl`: sum with RCCL's `ncclReduce` instead of peer copies added in a fixed order (`PT_GROUP_RCCL`).
    pub fn new(devices: &[i32], width: u32, height: u32, triangles: &[u8], nodes: &[u8], triangle_indices: &[u32], hdr_file: &[u8], rccl: bool) -> Result<Self, PtError> {
        let mut group: *mut PtGroup = std::ptr::null_mut();
        let flags = if rccl { PT_GROUP_RCCL } else { 0 };
        ok_group(unsafe { pt_group_create(devices.as_ptr(), devices.len() as i32, width, height, flags, &mut group) }, std::ptr::null())?;
        let r = RendererGroup { group, width, height, settings: Settings::default() };
        ok_group(
            unsafe {
                pt_group_set_mesh(group, triangles.as_ptr().cast(), (triangles.len() / 112) as u32, nodes.as_ptr().cast(), (nodes.len() / 32) as u32,
                                  triangle_indices.as_ptr(), triangle_indices.len() as u32)
            },
            group,
        )?;
        let (pano, w, h) = decode_hdr(hdr_file)?;
        ok_group(unsafe { pt_group_set_sky_equirect(group, pano.as_ptr(), w, h, 4096) }, group)?;
        Ok(r)
    }

    /// The uniform uploads of `update_buffers` (`renderer.rs:518-548`) on every GPU, then frames
    /// `time0 .. time0 + n_frames - 1` dealt over the GPUs (GPU g renders `time0 + g`, `time0 + g + G`, ...).
    pub fn render_frames(&mut self, camera64: &[u8], sphere_data8208: &[u8], time0: u32, n_frames: u32) -> Result<(), PtError> {
        assert!(camera64.len() == 64 && sphere_data8208.len() == 8208);
        ok_group(unsafe { pt_group_set_camera(self.group, camera64.as_ptr().cast()) }, self.group)?;
        ok_group(unsafe { pt_group_set_spheres(self.group, sphere_data8208.as_ptr().cast()) }, self.group)?;
        ok_group(unsafe { pt_group_set_settings(self.group, bytemuck::bytes_of(&self.settings).as_ptr().cast()) }, self.group)?;
        ok_group(unsafe { pt_group_render_frames(self.group, time0, n_frames) }, self.group)
    }

    /// Adds every GPU's sums onto the first GPU's (and clears the others'), then reads the fp32 mean over all frames
    /// rendered since the last reset.
    pub fn reduce_and_read_mean(&mut self) -> Result<Vec<f32>, PtError> {
        ok_group(unsafe { pt_group_reduce(self.group, 0) }, self.group)?;
        let root = unsafe { pt_group_member(self.group, 0) };
        let mut out = vec![0f32; self.width as usize * self.height as usize * 4];
        ok(unsafe { pt_read_mean_rgba32f(root, out.as_mut_ptr()) }, root)?;
        Ok(out)
    }

    /// `reset_ready_samples` (`renderer.rs:635-637`) on every GPU.
    pub fn reset_ready_samples(&mut self) {
        let _ = unsafe { pt_group_reset(self.group) };
    }

    pub fn stats(&mut self) -> Result<PtStats, PtError> {
        let mut s = PtStats::default();
        ok_group(unsafe { pt_group_stats(self.group, &mut s) }, self.group)?;
        Ok(s)
    }
}

impl Drop for RendererGroup {
    fn drop(&mut self) {
        unsafe { pt_group_destroy(self.group) }
    }
}
