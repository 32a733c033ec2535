//! Replacement for `src/fluid_compute.rs` of qts8n/water-sandbox (Bevy 0.13): the same public surface
//! (`FluidPlugin`, `FluidComputePlugin`, `FluidStaticProps`, `FluidParticlesInitial`, the systems `setup`,
//! `update`, `despawn_liquid`, both system-set chains of src/schedule.rs:24-36) with the
//! `bevy_app_compute` worker and its WGSL passes (src/fluid_compute.rs:239-412, assets/*.wgsl) replaced by
//! calls into libwsfluid.so (include/wsfluid.h).
//!
//! STATUS: source only.  This build image has no rustc / cargo and Bevy is not vendored, so this file has
//! NOT been compiled or type-checked; it is written against Bevy 0.13's public API as the reference uses it.
//! To build: drop it in as src/fluid_compute.rs, remove the `bevy_app_compute` dependency and the
//! `ShaderType` derives on `Gravity` (src/gravity.rs:9) and `FluidContainerExt` (src/fluid_container.rs:17),
//! and link with `cargo:rustc-link-lib=dylib=wsfluid`.
use std::ffi::{c_void, CStr};
use std::os::raw::{c_char, c_int};

use bevy::prelude::*;

use crate::fluid_container::FluidContainer;
use crate::gravity::Gravity;
use crate::helpers::cube_fluid;
use crate::schedule::{InGameSet, ShaderPhysicsSet};
use crate::state::GameState;

// lattice block and particle radius of the reference, src/fluid_compute.rs:15-17,:20.  The library's counting
// sort has no power-of-two restriction (the FIXME at :15 no longer applies).
const NI_SIZE: usize = 64;
const NJ_SIZE: usize = 32;
const NK_SIZE: usize = 32;
const PARTICLE_RADIUS: f32 = 0.1;
const FIXED_STEP: f32 = 1. / 60.; // PARTICLE_LOOKAHEAD_SCALAR, src/fluid_compute.rs:27

// ---------------------------------------------------------------------------------------------------
// include/wsfluid.h
// ---------------------------------------------------------------------------------------------------
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
pub struct WsHandle {
    _opaque: [u8; 0],
}

extern "C" {
    fn ws_create(params: *const WsParams, pos_xyz: *const f32, n: u32, cfg: *const c_void, out: *mut *mut WsHandle) -> c_int;
    fn ws_destroy(h: *mut WsHandle) -> c_int;
    fn ws_step(h: *mut WsHandle) -> c_int;
    fn ws_ready(h: *mut WsHandle, ready: *mut c_int) -> c_int;
    fn ws_set_params(h: *mut WsHandle, params: *const WsParams) -> c_int;
    fn ws_read_positions_begin(h: *mut WsHandle, out_xyz: *mut f32) -> c_int;
    fn ws_read_positions_end(h: *mut WsHandle) -> c_int;
    fn ws_pin_host_buffer(h: *mut WsHandle, ptr: *mut c_void, bytes: u64) -> c_int;
    fn ws_unpin_host_buffer(h: *mut WsHandle, ptr: *mut c_void) -> c_int;
    fn ws_reset(h: *mut WsHandle, pos_xyz: *const f32) -> c_int;
    fn ws_last_error(h: *mut WsHandle) -> *const c_char;
    fn ws_abi_version() -> u32;
}

// include/wsfluid.h WS_ABI_VERSION as this file was written against it (2: round 5 -- ws_transport.struct_size, flags
// 64 / 128 / 256).  A library of another version lays its structs out differently: refuse it at start-up.
const WS_ABI_VERSION: u32 = 2;

// ws_status values the shim tells apart (include/wsfluid.h)
const WS_ERR_INVALID_ARG: c_int = 1;
const WS_ERR_OUT_OF_MEMORY: c_int = 3;

fn check(h: *mut WsHandle, status: c_int, what: &str) {
    if status != 0 {
        let text = unsafe { CStr::from_ptr(ws_last_error(h)) }.to_string_lossy().into_owned();
        panic!("{what}: wsfluid status {status}: {text}");
    }
}

// ---------------------------------------------------------------------------------------------------
// resources that keep their reference definitions (src/fluid_compute.rs:41-85), minus the wgpu derives
// ---------------------------------------------------------------------------------------------------
#[derive(Resource, Clone, Copy)]
pub struct FluidStaticProps {
    pub delta_time: f32,
    pub collision_damping: f32,
    pub smoothing_radius: f32,
    pub target_density: f32,
    pub pressure_scalar: f32,
    pub near_pressure_scalar: f32,
    pub viscosity_strength: f32,
}

impl Default for FluidStaticProps {
    fn default() -> Self {
        // src/fluid_compute.rs:20-27,:67-79
        Self {
            delta_time: FIXED_STEP,
            collision_damping: 0.95,
            smoothing_radius: 0.25,
            target_density: 10.,
            pressure_scalar: 22.,
            near_pressure_scalar: 2.,
            viscosity_strength: 0.1,
        }
    }
}

#[derive(Resource, Clone, Default)]
pub struct FluidParticlesInitial {
    pub positions: Vec<Vec3>,
}

#[derive(Component, Debug)]
struct FluidParticleLabel(usize);

#[derive(Component, Default, Debug)]
struct Velocity(Vec3);

// ---------------------------------------------------------------------------------------------------
// the worker resource: takes the place of AppComputeWorker<FluidWorker>
// ---------------------------------------------------------------------------------------------------
#[derive(Resource)]
pub struct HipFluidWorker {
    handle: *mut WsHandle,
    params: WsParams,      // container extents are fixed at build time, as in the reference (:302)
    positions: Vec<f32>,   // n * 3, original-id order; page-locked for the lifetime of the worker
    readback_in_flight: bool,
    have_positions: bool,
    rejected_radius: Option<f32>, // last smoothing radius ws_set_params refused (logged once)
}
// A ws_handle is not thread-affine (every entry point selects its device); ResMut serialises the calls.
unsafe impl Send for HipFluidWorker {}
unsafe impl Sync for HipFluidWorker {}

impl Drop for HipFluidWorker {
    fn drop(&mut self) {
        unsafe {
            if self.readback_in_flight {
                ws_read_positions_end(self.handle);
            }
            ws_unpin_host_buffer(self.handle, self.positions.as_mut_ptr() as *mut c_void);
            ws_destroy(self.handle);
        }
    }
}

fn flatten(points: &[Vec3]) -> Vec<f32> {
    let mut flat = Vec::with_capacity(points.len() * 3);
    for p in points {
        flat.extend_from_slice(&[p.x, p.y, p.z]);
    }
    flat
}

fn fill_uniforms(params: &mut WsParams, props: &FluidStaticProps, gravity: &Gravity) {
    params.delta_time = props.delta_time;
    params.collision_damping = props.collision_damping;
    params.smoothing_radius = props.smoothing_radius;
    params.target_density = props.target_density;
    params.pressure_scalar = props.pressure_scalar;
    params.near_pressure_scalar = props.near_pressure_scalar;
    params.viscosity_strength = props.viscosity_strength;
    params.gravity = gravity.value.to_array();
}

pub struct FluidComputePlugin;

impl Plugin for FluidComputePlugin {
    fn build(&self, app: &mut App) {
        app.init_resource::<FluidStaticProps>()
            .init_resource::<FluidParticlesInitial>()
            .insert_resource(Time::<Fixed>::from_seconds(FIXED_STEP.into())); // :385
    }

    // what FluidComputeWorkerPlugin::finish + FluidWorker::build do in the reference, :277-366,:389-398
    fn finish(&self, app: &mut App) {
        let world = &mut app.world;
        let props = *world.resource::<FluidStaticProps>();
        let gravity = *world.resource::<Gravity>();
        let ext = world.resource::<FluidContainer>().get_ext(PARTICLE_RADIUS); // uploaded ONCE, as at :302
        let points = cube_fluid(NI_SIZE, NJ_SIZE, NK_SIZE, PARTICLE_RADIUS);
        world.resource_mut::<FluidParticlesInitial>().positions = points.clone();

        let mut params = WsParams::default();
        fill_uniforms(&mut params, &props, &gravity);
        params.ext_min = ext.ext_min.to_array();
        params.ext_max = ext.ext_max.to_array();
        let mut positions = flatten(&points);
        let mut handle: *mut WsHandle = std::ptr::null_mut();
        let abi = unsafe { ws_abi_version() };
        assert_eq!(abi, WS_ABI_VERSION, "libwsfluid.so speaks ABI version {abi}, this shim was written against {WS_ABI_VERSION}");
        let status = unsafe { ws_create(&params, positions.as_ptr(), points.len() as u32, std::ptr::null(), &mut handle) };
        check(std::ptr::null_mut(), status, "ws_create");
        // the readback target lives as long as the worker: page-lock it once (PCIe-rate copies)
        let bytes = (positions.len() * std::mem::size_of::<f32>()) as u64;
        check(handle, unsafe { ws_pin_host_buffer(handle, positions.as_mut_ptr() as *mut c_void, bytes) }, "ws_pin_host_buffer");

        app.insert_resource(HipFluidWorker { handle, params, positions, readback_in_flight: false, have_positions: false, rejected_radius: None })
            .add_systems(PostUpdate, (
                prepare_step.in_set(ShaderPhysicsSet::Prepare), // where the reference unmaps its staging buffers (:395)
                run_step.in_set(ShaderPhysicsSet::Pass),        // AppComputeWorker::run (:396)
            ));
    }
}

/// `ShaderPhysicsSet::Prepare` (where the reference unmaps its staging buffers, :395): if the step submitted last
/// frame has finished but `update` did not consume its positions this frame, complete the copy now so that the
/// buffer is free for the next one.  Never waits for a step that is still running.
fn prepare_step(mut worker: ResMut<HipFluidWorker>) {
    if !worker.readback_in_flight {
        return;
    }
    let h = worker.handle;
    let mut ready: c_int = 0;
    check(h, unsafe { ws_ready(h, &mut ready) }, "ws_ready");
    if ready != 0 {
        check(h, unsafe { ws_read_positions_end(h) }, "ws_read_positions_end");
        worker.readback_in_flight = false;
        worker.have_positions = true;
    }
}

/// `ShaderPhysicsSet::Pass`: enqueue one step (returns at once) and start the asynchronous, id-ordered copy of the
/// positions that step produces; it overlaps with the rest of the frame and is consumed by next frame's `update`.
/// While the previous submission is still in flight the frame is skipped, as `AppComputeWorker::run` does.
fn run_step(mut worker: ResMut<HipFluidWorker>) {
    if worker.readback_in_flight {
        return;
    }
    let h = worker.handle;
    check(h, unsafe { ws_step(h) }, "ws_step");
    let out = worker.positions.as_mut_ptr();
    check(h, unsafe { ws_read_positions_begin(h, out) }, "ws_read_positions_begin");
    worker.readback_in_flight = true;
}

pub struct FluidPlugin;

impl Plugin for FluidPlugin {
    fn build(&self, app: &mut App) {
        app.add_plugins(FluidComputePlugin)
            .add_systems(OnExit(GameState::Menu), setup)
            .add_systems(Update, update.in_set(InGameSet::EntityUpdates))
            .add_systems(Update, despawn_liquid.in_set(InGameSet::DespawnEntities));
    }
}

/// One entity per particle, labelled with its original particle id (src/fluid_compute.rs:438-465).
fn setup(
    mut commands: Commands,
    mut meshes: ResMut<Assets<Mesh>>,
    mut materials: ResMut<Assets<StandardMaterial>>,
    fluid_initials: Res<FluidParticlesInitial>,
) {
    let mesh = meshes.add(Sphere::new(PARTICLE_RADIUS).mesh().ico(0).unwrap());
    let material = materials.add(StandardMaterial { base_color: Color::CYAN, ..default() });
    let bundles: Vec<_> = fluid_initials
        .positions
        .iter()
        .enumerate()
        .map(|(id, &point)| {
            (
                PbrBundle { mesh: mesh.clone(), material: material.clone(), transform: Transform::from_translation(point), ..default() },
                Velocity::default(),
                FluidParticleLabel(id),
            )
        })
        .collect();
    commands.spawn_batch(bundles);
}

/// update(), src/fluid_compute.rs:468-486: gate on ready, take the positions of the step submitted last frame,
/// refresh fluid_props / smoothing_kernel / gravity (NOT the container: the reference uploads it once).
fn update(
    mut query: Query<(&mut Transform, &FluidParticleLabel)>,
    mut worker: ResMut<HipFluidWorker>,
    fluid_props: Res<FluidStaticProps>,
    gravity: Res<Gravity>,
) {
    let h = worker.handle;
    let mut ready: c_int = 0;
    check(h, unsafe { ws_ready(h, &mut ready) }, "ws_ready");
    if ready == 0 {
        return; // :474-476
    }
    if worker.readback_in_flight {
        check(h, unsafe { ws_read_positions_end(h) }, "ws_read_positions_end"); // 12 B / particle instead of 80
        worker.readback_in_flight = false;
        worker.have_positions = true;
    }
    let mut params = worker.params;
    fill_uniforms(&mut params, &fluid_props, &gravity);
    // the three worker.write calls, :479-481.  A parameter set the library cannot take (a HUD key press can make the
    // smoothing radius zero or negative, src/hud.rs:135-138; WS_ERR_INVALID_ARG / WS_ERR_OUT_OF_MEMORY) must not take
    // the app down: the library keeps its previous parameters, and so does the worker; say so once per rejected value.
    // (Both codes come from the checks ws_set_params makes BEFORE it touches anything.  An allocation that fails after
    // the old cell tables were given up leaves the handle dead instead: the next ws_step reports it -- `check` below.)
    let status = unsafe { ws_set_params(h, &params) };
    if status == 0 {
        worker.params = params;
        worker.rejected_radius = None;
    } else if status == WS_ERR_INVALID_ARG || status == WS_ERR_OUT_OF_MEMORY {
        if worker.rejected_radius != Some(params.smoothing_radius) {
            let text = unsafe { CStr::from_ptr(ws_last_error(h)) }.to_string_lossy().into_owned();
            warn!("ws_set_params rejected (status {status}: {text}); keeping the previous parameters");
            worker.rejected_radius = Some(params.smoothing_radius);
        }
    } else {
        check(h, status, "ws_set_params");
    }
    if !worker.have_positions {
        return; // nothing has been stepped yet
    }
    let positions = &worker.positions;
    query.par_iter_mut().for_each(|(mut transform, particle)| {
        let at = particle.0 * 3;
        transform.translation = Vec3::new(positions[at], positions[at + 1], positions[at + 2]);
    });
}

/// despawn_liquid(), src/fluid_compute.rs:505-525: Space ends the game and puts the fluid back to t = 0.
fn despawn_liquid(
    mut worker: ResMut<HipFluidWorker>,
    mut next_state: ResMut<NextState<GameState>>,
    fluid_initials: Res<FluidParticlesInitial>,
    keyboard_input: Res<ButtonInput<KeyCode>>,
) {
    let h = worker.handle;
    let mut ready: c_int = 0;
    check(h, unsafe { ws_ready(h, &mut ready) }, "ws_ready");
    if !keyboard_input.just_pressed(KeyCode::Space) || ready == 0 {
        return;
    }
    next_state.set(GameState::GameOver);
    if worker.readback_in_flight {
        check(h, unsafe { ws_read_positions_end(h) }, "ws_read_positions_end");
        worker.readback_in_flight = false;
    }
    worker.have_positions = false;
    let flat = flatten(&fluid_initials.positions);
    // the four write_slice calls (:521-524): particles <- initial records, the three index buffers <- identity
    check(h, unsafe { ws_reset(h, flat.as_ptr()) }, "ws_reset");
}
