//! Source-only Bevy 0.13 shim: replaces `FluidComputePlugin` / `FluidComputeWorkerPlugin` /
//! `FluidWorker` of src/fluid_compute.rs (:239-412) with calls into libwsfluid.so while keeping
//! `FluidPlugin`, `FluidStaticProps`, `FluidParticlesInitial`, `setup`, `update`, `despawn_liquid`
//! and both system-set chains (src/schedule.rs:24-36) as they are.
//!
//! NOT compiled in this repository (no rustc/cargo in the build image; Bevy is not vendored).
//! Build where Bevy 0.13 is available: add `links = "wsfluid"` / `cargo:rustc-link-lib=dylib=wsfluid`
//! in build.rs, drop the `bevy_app_compute` dependency and its `ShaderType` derives on
//! `Gravity` (src/gravity.rs:9), `FluidContainerExt` (src/fluid_container.rs:17) and
//! `FluidStaticProps` (src/fluid_compute.rs:41).
use bevy::prelude::*;
use std::os::raw::{c_char, c_int};

use crate::fluid_container::FluidContainer;
use crate::gravity::Gravity;
use crate::helpers::cube_fluid;
use crate::schedule::{InGameSet, ShaderPhysicsSet};
use crate::state::GameState;

// ---- include/wsfluid.h ------------------------------------------------------------------------
#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct WsParams {
    pub delta_time: f32,
    pub collision_damping: f32,
    pub smoothing_radius: f32,
    pub target_density: f32,
    pub pressure_scalar: f32,
    pub near_pressure_scalar: f32,
    pub viscosity_strength: f32,
    pub reserved0: f32,
    pub gravity: [f32; 4],
    pub ext_min: [f32; 4],
    pub ext_max: [f32; 4],
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct WsDeviceCfg {
    pub device: i32,
    pub flags: u32,
    pub rank: u32,
    pub world_size: u32,
    pub capacity: u32,
    pub ghost_capacity: u32,
    pub reserved: [u32; 2],
    pub stream: *mut std::ffi::c_void,
}

#[repr(C)]
pub struct WsHandle {
    _private: [u8; 0],
}

extern "C" {
    fn ws_create(params: *const WsParams, pos_xyz: *const f32, n: u32, cfg: *const WsDeviceCfg,
                 out: *mut *mut WsHandle) -> c_int;
    fn ws_destroy(h: *mut WsHandle) -> c_int;
    fn ws_step(h: *mut WsHandle) -> c_int;
    fn ws_ready(h: *mut WsHandle, ready: *mut c_int) -> c_int;
    fn ws_set_params(h: *mut WsHandle, params: *const WsParams) -> c_int;
    fn ws_read_positions(h: *mut WsHandle, out_xyz: *mut f32) -> c_int;
    fn ws_pin_host_buffer(h: *mut WsHandle, ptr: *mut std::ffi::c_void, bytes: u64) -> c_int;
    fn ws_unpin_host_buffer(h: *mut WsHandle, ptr: *mut std::ffi::c_void) -> c_int;
    fn ws_reset(h: *mut WsHandle, pos_xyz: *const f32) -> c_int;
    fn ws_last_error(h: *mut WsHandle) -> *const c_char;
}

const NI_SIZE: usize = 64; // src/fluid_compute.rs:15-17 (any size: the counting sort has no power-of-two limit)
const NJ_SIZE: usize = 32;
const NK_SIZE: usize = 32;
const PARTICLE_RADIUS: f32 = 0.1;

// FluidStaticProps, FluidParticlesInitial: unchanged from src/fluid_compute.rs:41-85 (minus ShaderType).
pub use crate::fluid_props::{FluidParticlesInitial, FluidStaticProps};

/// The resource that takes the place of `AppComputeWorker<FluidWorker>`.
#[derive(Resource)]
pub struct HipFluidWorker {
    handle: *mut WsHandle,
    positions: Vec<f32>, // n * 3, original-id order
}
// ws_handle is not thread-affine (every entry point sets its device); Bevy serialises access via ResMut.
unsafe impl Send for HipFluidWorker {}
unsafe impl Sync for HipFluidWorker {}

impl Drop for HipFluidWorker {
    fn drop(&mut self) {
        unsafe {
            ws_unpin_host_buffer(self.handle, self.positions.as_mut_ptr() as *mut _);
            ws_destroy(self.handle)
        };
    }
}

fn make_params(props: &FluidStaticProps, gravity: &Gravity, container: &FluidContainer) -> WsParams {
    let ext = container.get_ext(PARTICLE_RADIUS); // src/fluid_compute.rs:302
    WsParams {
        delta_time: props.delta_time,
        collision_damping: props.collision_damping,
        smoothing_radius: props.smoothing_radius,
        target_density: props.target_density,
        pressure_scalar: props.pressure_scalar,
        near_pressure_scalar: props.near_pressure_scalar,
        viscosity_strength: props.viscosity_strength,
        reserved0: 0.0,
        gravity: gravity.value.to_array(),
        ext_min: ext.ext_min.to_array(),
        ext_max: ext.ext_max.to_array(),
    }
}

pub struct FluidComputePlugin;

impl Plugin for FluidComputePlugin {
    fn build(&self, app: &mut App) {
        app.init_resource::<FluidStaticProps>()
            .init_resource::<FluidParticlesInitial>()
            .insert_resource(Time::<Fixed>::from_seconds((1.0f32 / 60.0).into())); // :385
    }

    // FluidWorker::build, src/fluid_compute.rs:277-366
    fn finish(&self, app: &mut App) {
        let world = &mut app.world;
        let params = make_params(world.resource::<FluidStaticProps>(), world.resource::<Gravity>(),
                                 world.resource::<FluidContainer>());
        let points = cube_fluid(NI_SIZE, NJ_SIZE, NK_SIZE, PARTICLE_RADIUS);
        world.resource_mut::<FluidParticlesInitial>().positions = points.clone();
        let flat: Vec<f32> = points.iter().flat_map(|p| p.to_array()).collect();
        let mut handle: *mut WsHandle = std::ptr::null_mut();
        let st = unsafe { ws_create(&params, flat.as_ptr(), points.len() as u32, std::ptr::null(), &mut handle) };
        assert_eq!(st, 0, "ws_create failed: {:?}", unsafe { std::ffi::CStr::from_ptr(ws_last_error(std::ptr::null_mut())) });
        // the readback buffer lives as long as the worker: page-lock it once (PCIe-rate ws_read_positions)
        let mut flat = flat;
        unsafe { ws_pin_host_buffer(handle, flat.as_mut_ptr() as *mut _, (flat.len() * 4) as u64) };
        app.insert_resource(HipFluidWorker { handle, positions: flat })
            // the reference's unmap_all (Prepare) has nothing to do here; run (Pass) enqueues one step
            .add_systems(PostUpdate, run_step.in_set(ShaderPhysicsSet::Pass));
    }
}

/// AppComputeWorker::run, src/fluid_compute.rs:396 — returns at once, the step runs on the GPU stream.
fn run_step(worker: ResMut<HipFluidWorker>) {
    unsafe { ws_step(worker.handle) };
}

/// update(), src/fluid_compute.rs:468-486
pub fn update(
    mut query: Query<(&mut Transform, &super::FluidParticleLabel)>,
    mut worker: ResMut<HipFluidWorker>,
    fluid_props: Res<FluidStaticProps>,
    gravity: Res<Gravity>,
    container: Res<FluidContainer>,
) {
    let mut ready: c_int = 0;
    unsafe { ws_ready(worker.handle, &mut ready) };
    if ready == 0 {
        return; // :474-476
    }
    let h = worker.handle;
    unsafe { ws_read_positions(h, worker.positions.as_mut_ptr()) }; // 12 B/particle instead of 80
    let params = make_params(&fluid_props, &gravity, &container);
    unsafe { ws_set_params(h, &params) }; // the three worker.write calls, :479-481
    let positions = &worker.positions;
    query.par_iter_mut().for_each(|(mut transform, particle)| {
        let i = particle.0 * 3;
        transform.translation = Vec3::new(positions[i], positions[i + 1], positions[i + 2]);
    });
}

/// despawn_liquid(), src/fluid_compute.rs:505-525
pub fn despawn_liquid(
    worker: ResMut<HipFluidWorker>,
    mut next_state: ResMut<NextState<GameState>>,
    fluid_initials: Res<FluidParticlesInitial>,
    keyboard_input: Res<ButtonInput<KeyCode>>,
) {
    let mut ready: c_int = 0;
    unsafe { ws_ready(worker.handle, &mut ready) };
    if !keyboard_input.just_pressed(KeyCode::Space) || ready == 0 {
        return;
    }
    next_state.set(GameState::GameOver);
    let flat: Vec<f32> = fluid_initials.positions.iter().flat_map(|p| p.to_array()).collect();
    unsafe { ws_reset(worker.handle, flat.as_ptr()) }; // the four write_slice calls, :521-524
}
