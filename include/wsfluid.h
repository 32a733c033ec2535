/*
 * wsfluid.h -- C ABI of the MI355X-native SPH fluid step (libwsfluid.so).
 *
 * This is the drop-in boundary for the one hot path of qts8n/water-sandbox: the
 * per-frame SPH step that the reference builds in src/fluid_compute.rs and runs as
 * assets/simulation.wgsl + assets/bitonic_sort.wgsl through bevy_app_compute's
 * AppComputeWorker.  Each entry point names the reference interface it replaces
 * (paths relative to the reference tree).  The reference-side binding (the Rust
 * `extern "C"` block and the Bevy systems that call it) is in INTEGRATION.md.
 *
 * Conventions: plain pointers and sizes only; the library owns all device memory;
 * the caller owns every pointer it passes and the library never keeps one after the
 * call returns; every function returns a ws_status (0 = ok) and never throws or
 * aborts; a handle is not re-entrant (calls on one handle must be serialised by the
 * caller -- Bevy's ResMut does this) but may be used from any host thread.
 * There is NO CPU fallback: without a gfx950 device ws_create fails with
 * WS_ERR_NO_DEVICE.
 */
#ifndef WSFLUID_H
#define WSFLUID_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2 (round 5): ws_transport starts with struct_size and has four callbacks; flags 64 / 128 / 256 have meanings; ws_read_stats out[5..14] are assigned.  A host built against
 * version 1 must not pass its structs to this library: compare ws_abi_version() with WS_ABI_VERSION at start-up. */
#define WS_ABI_VERSION 2

typedef enum ws_status {
    WS_OK = 0,
    WS_ERR_INVALID_ARG = 1,   /* null pointer, n == 0, non-finite / non-positive h, empty box */
    WS_ERR_NO_DEVICE = 2,     /* no HIP device / not gfx950 / bad device index */
    WS_ERR_OUT_OF_MEMORY = 3, /* hipMalloc failed or the cell grid would not fit */
    WS_ERR_HIP = 4,           /* any other HIP runtime error (text in ws_last_error) */
    WS_ERR_COMM = 5,          /* RCCL error in the multi-GPU halo exchange */
    WS_ERR_UNSUPPORTED = 6,   /* e.g. WS_FLAG_REFERENCE_ORDER on the product library, ws_create with world_size > 1 */
    WS_ERR_NOT_READY = 7      /* ws_try_* variants only */
} ws_status;

/*
 * Everything the reference uploads as uniforms, in one POD.
 *   fields 0..6   FluidStaticProps        src/fluid_compute.rs:41-51 (defaults :20-27,:67-79)
 *   gravity       Gravity.value           src/gravity.rs:9-13 (default (0,-9.8,0,0) :29-33)
 *   ext_min/max   FluidContainerExt       src/fluid_container.rs:17-22, from get_ext(0.1) :42-50
 * The five SmoothingKernel constants (src/fluid_compute.rs:30-38) are NOT passed in:
 * the library derives them exactly as get_smoothing_kernel does (:55-63) whenever
 * smoothing_radius is set, which is what update() does every frame (:480).
 */
typedef struct ws_params {
    float delta_time;
    float collision_damping;
    float smoothing_radius;
    float target_density;
    float pressure_scalar;
    float near_pressure_scalar;
    float viscosity_strength;
    float reserved0; /* must be 0 */
    float gravity[4];
    float ext_min[4];
    float ext_max[4];
} ws_params;

/* SmoothingKernel, src/fluid_compute.rs:30-38 */
typedef struct ws_smoothing_kernel {
    float pow2;
    float pow2_der;
    float pow3;
    float pow3_der;
    float spikey_pow3;
} ws_smoothing_kernel;

/* FluidParticle, src/fluid_compute.rs:106-115 == assets/simulation.wgsl:69-76.
 * 80 bytes, same field order and offsets, so a Rust Vec<FluidParticle> can be
 * passed straight in and out. */
typedef struct ws_particle80 {
    float position[4];
    float density[2];
    float pressure[2];
    float velocity[4];
    float acceleration[4];
    float predicted_position[4];
} ws_particle80;

/* Placement of this handle's share of the domain.  Zero-initialise for one GPU. */
typedef struct ws_device_cfg {
    int32_t device;          /* HIP device ordinal */
    uint32_t flags;          /* WS_FLAG_* */
    uint32_t rank;           /* slab index along x, 0-based (0 for one GPU) */
    uint32_t world_size;     /* number of slabs (0 or 1 = single GPU) */
    uint32_t capacity;       /* slabs: max particles this handle may own (0 = 2 n_local + 2^20);
                                head-room for particles migrating in */
    uint32_t ghost_capacity; /* slabs: max particles of ONE boundary layer = ghosts per face = the capacity of a halo
                                message (0 = 16 n_global / nx + 2^15, nx = cell layers along x: sixteen times an evenly
                                spread layer; what travels per step is sized by the layer's actual population) */
    uint32_t reserved[2];
    void *stream;            /* hipStream_t to enqueue on, or NULL: the library creates its own.
                                A host that moves halos with its own communication library passes
                                that library's stream here so both are ordered on it. */
} ws_device_cfg;

#define WS_FLAG_NONE 0u
#define WS_FLAG_PROFILE 1u /* record HIP events around every kernel of ws_step */
/* Validation mode: execute the reference's six passes literally on the GPU -- N-bucket hashed table,
 * the bitonic network stage by stage on the persisted permutation (src/fluid_compute.rs:256-271,
 * assets/bitonic_sort.wgsl:22-46), atomicMin cell offsets, bucket walks in OFFSET_TABLE order -- with
 * the same IEEE arithmetic.  It reproduces the reference's summation order, so it is comparable with
 * a bit-faithful CPU restatement on every float and on particle_indicies itself.  Slow (S dispatches
 * per step); single GPU only; never the benchmarked path. */
#define WS_FLAG_REFERENCE_ORDER 2u
/* Pair terms of K4/K5 with correctly rounded sqrt and division (the CPU oracle's arithmetic) instead of the
 * default hardware v_sqrt_f32 / v_rcp_f32 forms (1 ULP each; x / y evaluated as x * rcp(y)), which stay inside the
 * accuracy WGSL grants the reference's own GPU execution (x / y: 2.5 ULP).  ~2x slower density/force kernels. */
#define WS_FLAG_IEEE_DIVISION 4u
/* Replay the step from a captured hipGraph instead of launching its kernels one by one: what the reference does with
 * its pass graph (built once in FluidWorker::build, replayed by AppComputeWorker::run every frame,
 * src/fluid_compute.rs:309-363,:396).  The step is captured on the first steady-state ws_step (and again when a slab's
 * launch bound moves by more than 65 536 particles); results are identical to direct launches.  On slab handles the
 * flag is honoured with the library's own RCCL transport (and for a world of one); a host-supplied transport may
 * synchronise the stream inside its callbacks, so such handles keep launching directly.  Ignores WS_FLAG_PROFILE.  Off by default: direct launches already pipeline on the
 * stream and measure as fast on one MI355X. */
#define WS_FLAG_GRAPH 8u
/* Slab handles, how the messages of a step are sized.
 * DEFAULT (neither flag, no WS_FLAG_GRAPH): every message carries exactly what its sender has for it.  ws_step gathers
 * four words per rank and waits for them twice per step -- before the migration (the GPU idles for one small all-gather
 * and a copy) and before the halos (behind the kernel that needs no ghosts).  Nothing can overrun below the buffers'
 * capacities, and the messages are as small as they can be.
 * WS_FLAG_LAGGED_MESSAGES (implied by WS_FLAG_GRAPH: a captured step has its sizes baked in): ws_step never waits for
 * the device; every rank derives the sizes from the demand all ranks reported a few steps earlier (x 4 headroom on the
 * largest of the last eight reports).  A demand that outgrows that within four steps FAILS the run -- on every rank at the
 * same step, cleanly, but it fails: a pressure front that crosses a slab face broadside multiplies the particles changing
 * owner tenfold in one step.  For flows known to be smooth across the slab faces -- the benchmark
 * trajectories are.
 * WS_FLAG_FIXED_MESSAGES: every message travels at its full CAPACITY.  ws_step never waits, nothing below the
 * capacities can overrun, the sizes never change (so a captured step is never re-captured) -- and every link moves the
 * capacity every step (DESIGN.md 6 has the bytes).  This is what WS_FLAG_GRAPH uses on a handle with peers unless
 * WS_FLAG_LAGGED_MESSAGES is given with it: a captured multi-rank step is safe by default, the bet is the opt-in.
 * WS_FLAG_EXACT_MESSAGES asks for the default explicitly (and wins over WS_FLAG_GRAPH, which is then ignored).
 * Every rank must choose alike: ws_slab_create compares the ranks' choices and fails on ALL of them
 * (WS_ERR_INVALID_ARG) when they differ. */
#define WS_FLAG_EXACT_MESSAGES 16u
#define WS_FLAG_LAGGED_MESSAGES 32u
#define WS_FLAG_FIXED_MESSAGES 64u
/* Slab handles: keep the halo exchanges on the step's own stream (no early / late split of K4 / K5, no second
 * communicator).  Same results; for transports that cannot drive two streams, and for A/B runs.  Every rank alike. */
#define WS_FLAG_NO_OVERLAP 128u
/* Slab handles with peers: let WS_FLAG_GRAPH capture the step although no run on two or more GPUs has executed a
 * captured transport call yet (the path runs through the tests' stand-in for librccl only: DESIGN.md 6).  Needs the
 * library's own RCCL transport with two communicators.  Without it a multi-rank handle launches directly.  Every rank
 * alike. */
#define WS_FLAG_GRAPH_MULTIRANK 256u

typedef struct ws_handle ws_handle;

/* ---- host-side functions of the path (pure CPU, no device needed) ----------- */

/* FluidStaticProps::default + Gravity::default + FluidContainer::default().get_ext(0.1)
 * (src/fluid_compute.rs:67-79, src/gravity.rs:29-33, src/fluid_container.rs:8-9,34-40). */
ws_status ws_default_params(ws_params *out);
/* FluidStaticProps::get_smoothing_kernel, src/fluid_compute.rs:55-63. */
ws_status ws_get_smoothing_kernel(const ws_params *p, ws_smoothing_kernel *out);
/* helpers::cube_fluid, src/helpers.rs:3-20. out_xyz holds ni*nj*nk*3 floats. */
ws_status ws_cube_fluid(uint32_t ni, uint32_t nj, uint32_t nk, float particle_rad, float *out_xyz);
/* FluidContainer::get_ext, src/fluid_container.rs:42-50. */
ws_status ws_get_ext(const float position[3], const float size[3], float padding,
                     float ext_min[4], float ext_max[4]);
/* FluidWorker::get_bit_sorter_stages count, src/fluid_compute.rs:251-273.  The HIP
 * path does not run the network; exposed because the reference prints it (:321). */
uint32_t ws_bit_sorter_stage_count(uint32_t data_length);
const char *ws_status_string(ws_status s);
uint32_t ws_abi_version(void);

/* ---- lifetime: replaces FluidWorker::build + AppComputeWorkerBuilder --------- */

/* src/fluid_compute.rs:277-366.  pos_xyz: n*3 floats, particle id = index (the id
 * FluidParticleLabel carries, :416-417,:459).  Uploads the particles with
 * position = predicted_position = point and everything else 0 (:118-130), identity
 * permutation (:293).  cfg may be NULL (device 0, single GPU). */
ws_status ws_create(const ws_params *params, const float *pos_xyz, uint32_t n,
                    const ws_device_cfg *cfg, ws_handle **out);
ws_status ws_destroy(ws_handle *h);

/* ---- per frame ---------------------------------------------------------------- */

/* AppComputeWorker::run in ShaderPhysicsSet::Pass (src/fluid_compute.rs:396): enqueue
 * one full step hash -> sort -> cell starts -> density -> force -> integrate
 * (pass order :309-363) on the handle's stream and return without waiting. */
ws_status ws_step(ws_handle *h);
/* AppComputeWorker::ready (src/fluid_compute.rs:474,:511): *ready = 1 when every
 * enqueued step has finished, else 0.  Never blocks. */
ws_status ws_ready(ws_handle *h, int *ready);
/* Block until every enqueued step has finished (the reference has no equivalent; its
 * host simply skips the frame while !ready()). */
ws_status ws_sync(ws_handle *h);
/* The three worker.write calls of update() (src/fluid_compute.rs:479-481):
 * fluid_props, smoothing_kernel (re-derived here) and gravity; the container is
 * taken too (the reference uploads it once, :302).  Takes effect at the next ws_step.
 * Errors: WS_ERR_INVALID_ARG / WS_ERR_OUT_OF_MEMORY from the checks made BEFORE anything is touched (a radius <= 0, a
 * grid beyond the cell budget, on slab handles a slab whose share of the fluid on the new grid exceeds its capacity --
 * decided from the gathered state for every rank alike): the handle keeps its previous parameters and goes on.  A new
 * smoothing radius or container rebuilds the cell tables; an allocation that fails AFTER the old ones were given up
 * leaves the handle dead -- every later ws_step returns WS_ERR_HIP with the reason -- never half-built tables under
 * a handle that still steps. */
ws_status ws_set_params(ws_handle *h, const ws_params *params);
/* worker.read_vec::<FluidParticle>("particles") followed by `.position.xyz()` per
 * label (src/fluid_compute.rs:478,:483-485): n*3 floats in ORIGINAL-ID order.
 * Waits for enqueued steps first. */
ws_status ws_read_positions(ws_handle *h, float *out_xyz);
/* The same readback split in two for a frame loop that overlaps it with the next step (SURVEY 8(b) "pipelining
 * contract"): _begin captures the positions as of the steps enqueued so far and starts the copy into out_xyz
 * (page-lock it: ws_pin_host_buffer) on a separate stream and returns; ws_step calls made after it run while
 * the copy is in flight; _end waits for the copy.  One readback in flight per handle; out_xyz must stay valid
 * until _end. */
ws_status ws_read_positions_begin(ws_handle *h, float *out_xyz);
ws_status ws_read_positions_end(ws_handle *h);
/* out_xyz == NULL in ws_read_positions_begin: the copy goes into one of TWO page-locked buffers the library owns
 * (SURVEY 8(b) "Ownership": the library owns all pinned staging), filled alternately; after _end,
 * ws_read_positions_view hands out the buffer the last finished readback filled.  It stays untouched until the
 * second-next _begin(h, NULL), so update() can still scatter frame k's positions into its Transforms
 * (src/fluid_compute.rs:483-485) while frame k+1's copy is in flight.  The copy runs on an SDMA engine, on a stream
 * of a priority of its own (it never shares a hardware queue with the step's stream). */
ws_status ws_read_positions_view(ws_handle *h, const float **out_xyz);
/* `velocities.length()` per particle in ORIGINAL-ID order: n floats -- what the reference's (commented-out)
 * speed colouring system reads back 80 B per particle for (src/fluid_compute.rs:489-502).  Waits for
 * enqueued steps first. */
ws_status ws_read_speeds(ws_handle *h, float *out_speed);
/* Optional: page-lock a host buffer the caller owns and keeps alive -- typically the position buffer
 * update() fills every frame -- so that ws_read_positions / ws_read_particles into it run at PCIe rate
 * (no pageable staging).  The caller must ws_unpin_host_buffer it before freeing it. */
ws_status ws_pin_host_buffer(ws_handle *h, void *ptr, uint64_t bytes);
ws_status ws_unpin_host_buffer(ws_handle *h, void *ptr);
/* The full read_vec view (src/fluid_compute.rs:478): n records of 80 bytes in
 * original-id order.  density/pressure/acceleration are the values the last step
 * computed (0 before the first step).  The step itself keeps positions and velocities only: the acceleration
 * field (read by nothing in the reference but this view) is produced here by one more pass of the force kernel over
 * the state the last step left behind -- the same bits the step used; a frame loop that reads positions pays
 * nothing for it. */
ws_status ws_read_particles(ws_handle *h, ws_particle80 *out);
/* despawn_liquid's four write_slice calls (src/fluid_compute.rs:517-524): particles
 * <- initial state from pos_xyz, index buffers <- identity. */
ws_status ws_reset(ws_handle *h, const float *pos_xyz);
/* worker.write_slice("particles", ..) with an arbitrary state (position, velocity,
 * predicted_position are taken; the other fields are recomputed by the next step
 * before they are read, as in the reference).  Checkpoint/restore and the tests'
 * teacher forcing use this. */
ws_status ws_write_particles(ws_handle *h, const ws_particle80 *in);

/* ---- the per-frame calls above on a SLAB handle (multi-GPU, below) ---------------------------------------------
 * ws_read_positions / _begin / _end, ws_read_speeds, ws_read_particles, ws_read_sort_view, ws_reset, ws_write_particles
 * and a ws_set_params that changes the smoothing radius or the container work on slab handles too, over GLOBAL,
 * id-ordered arrays (n_global entries), as COLLECTIVE calls: every rank makes the same call at the same point of its
 * frame, as the ranks of a multi-GPU host do anyway.
 *   reads:   every rank receives the whole array (the owned records of all slabs are all-gathered and scattered by id
 *            on the device); a rank that does not need it passes NULL and only contributes.
 *   loads:   every rank passes the SAME global array (what FluidParticlesInitial holds, src/fluid_compute.rs:82-85) and
 *            keeps the particles its cuts own; no data moves between ranks.  Sticky errors are cleared.
 *   re-grid: the particles are redistributed by the new cuts (any particle may change owner).
 * ws_num_particles stays the number of particles THIS slab owns. */

/* ---- multi-GPU: one handle = one x-slab of the domain, one process per GPU ------------------
 *
 * The domain is cut into world_size slabs along x on cell boundaries (x is the slowest axis of the
 * cell grid, so a slab is a contiguous range of the global cell order and its boundary layers are
 * contiguous particle ranges).  Interactions reach one cell, so a slab needs one ghost layer from
 * each x-neighbour.  ws_step on a slab handle enqueues, per step:
 *   hand particles whose predicted position left the slab to their new owner (migration: one send/recv with each
 *   neighbour, plus one all-to-all for the particles that cross several slabs in a step and for the status words) -> sort own
 *   particles -> send the two boundary layers' records to the neighbours (halo A) -> K4 -> send their densities
 *   (halo B) -> K5+K6; with the halos on a second stream while the particles that need no ghosts compute.
 * ws_step never waits for the END of the step it enqueues on a slab handle either: every message has a fixed capacity known
 * to both ends (ghost_capacity and sizes derived from it) and carries its record count in a header; the owned count, the
 * layer ranges and the ghost counts stay on the device.  By default it waits, twice per step, for four words per rank --
 * the record counts its messages are sized from (WS_FLAG_EXACT_MESSAGES, above); the rest of the step -- the late
 * kernels, the second halo, the force kernel -- is still running when it returns.  With WS_FLAG_LAGGED_MESSAGES it
 * waits for nothing of the step at all (only, a bounded run-ahead, for the status table of the step enqueued two
 * calls earlier) and launches its kernels over host-side upper bounds.  A capacity
 * overrun clamps, sets a sticky error bit that reaches every rank with the next step's all-to-all, and makes ws_step
 * return WS_ERR_OUT_OF_MEMORY on ALL ranks at the same step (two steps later), before any collective of that step --
 * no rank is left waiting in one.  ws_sync / ws_slab_read_particles / ws_slab_counters report the bits too, as soon
 * as this rank knows them -- which may be one or two steps before the other ranks do: such a report is information,
 * not the signal to stop; keep calling ws_step until IT fails (it does on every rank at the same step, and until
 * then it keeps issuing the step's collectives so that no peer waits alone).  ws_num_particles of a slab is exact
 * after ws_sync.
 * All data movement goes through the three transport callbacks below (bench.py uses the library's own RCCL
 * transport, ws_rccl_transport_create; tests also drive them with torch.distributed and with an in-process
 * loopback).  The particle order inside a cell is canonical (by id), so an N-slab run reproduces the single-GPU
 * run bit for bit.  The reference has no multi-device path; this is the scale-out row of SURVEY.md 8(e). */
typedef struct ws_transport {
    uint64_t struct_size; /* sizeof(ws_transport) of the host's build: ws_slab_create refuses a table that is shorter
                             than the one it was compiled with (a version-1 host's three-callback struct) instead of
                             calling through whatever lies behind it */
    void *ctx;
    /* Stream-ordered exchange of nseg buffers with each x-neighbour, d = 0 (rank - 1) and d = 1 (rank + 1), as
     * ONE group of point-to-point transfers: for segment k send send_bytes[2k + d] bytes from DEVICE pointer
     * send_ptr[2k + d] and receive recv_bytes[2k + d] bytes into DEVICE pointer recv_ptr[2k + d].  Zero bytes =
     * no transfer.  Both sides list their segments in the same order.  Returns 0 on success. */
    int (*sendrecv)(void *ctx, uint32_t nseg, void *const send_ptr[], const uint64_t send_bytes[],
                    void *const recv_ptr[], const uint64_t recv_bytes[], void *stream);
    /* Stream-ordered all-gather of bytes_each DEVICE bytes per rank into recv_ptr[world_size * bytes_each]. */
    int (*allgather_dev)(void *ctx, const void *send_ptr, void *recv_ptr, uint64_t bytes_each, void *stream);
    /* Stream-ordered all-to-all of DEVICE buffers: bytes [r * bytes_each, (r + 1) * bytes_each) of send_ptr go to rank
     * r, which stores them at [this rank * bytes_each, ...) of its recv_ptr; the segment a rank addresses to itself is
     * copied too (RCCL: ncclAllToAll).  Once per step on the handle's stream: the per-destination messages for particles
     * that cross more than one slab in a step, each headed by the sender's status words -- so every rank also ends up
     * with every rank's status.  Required when world_size > 1. */
    int (*alltoall_dev)(void *ctx, const void *send_ptr, void *recv_ptr, uint64_t bytes_each, void *stream);
} ws_transport;

/* A ws_transport implemented inside the library with RCCL (ncclSend / ncclRecv groups with the two x-neighbours,
 * ncclAllToAll for the far messages and the status words, ncclAllGather for the host's collective reads), for hosts
 * that have no communication layer of their own.  librccl is loaded at run time.  Rank 0 draws a unique id and the host hands its 128 bytes to every rank (any out-of-band channel);
 * every rank then creates its transport -- a collective call -- on the device its slab will live on.  The
 * transport must outlive the slab handle created with it. */
#define WS_RCCL_UNIQUE_ID_BYTES 128
ws_status ws_rccl_unique_id(void *out128);
ws_status ws_rccl_transport_create(const void *unique_id, uint32_t rank, uint32_t world_size, int32_t device,
                                   ws_transport *out);
void ws_rccl_transport_destroy(ws_transport *t);
const char *ws_rccl_last_error(void);
/* Communicators the transport drives: 2 = the step's two streams (migration / all-to-all on the handle's stream, halos on
 * its communication stream) each keep to a communicator of their own (the second one is split off the first), so no
 * communicator ever sees operations from two streams; 1 = an RCCL without ncclCommSplit (or the developer build's WS_RCCL_SINGLE_COMM=1 hook). */
uint32_t ws_rccl_transport_communicators(const ws_transport *t);

/* A ws_transport for several slabs inside ONE process, one host thread per slab (one GPU or several): plain
 * device-to-device copies between the slabs' message buffers with a host rendezvous.  For hosts that drive all their GPUs
 * from one process, and for exercising the whole slab protocol on a one-GPU box (host/frame_loop.cpp, tests).  Every
 * call synchronises its stream: correct, not fast, and not capturable (WS_FLAG_GRAPH falls back to direct launches).  A
 * rank that is left alone in a collective for 300 s breaks the hub (every call then fails) instead of hanging.  The hub
 * must outlive its transports, a transport the handle created with it. */
ws_status ws_local_hub_create(uint32_t world_size, void **hub_out);
void ws_local_hub_destroy(void *hub);
ws_status ws_local_transport_create(void *hub, uint32_t rank, ws_transport *out);
void ws_local_transport_destroy(ws_transport *t);

/* Host-only: which slab owns each position (by the x cell of floor(x / h) in the global grid, equal
 * cell-count cuts S_r = r * nx / world_size).  out_rank holds n entries. */
ws_status ws_slab_assign(const ws_params *params, const float *pos_xyz, uint32_t n, uint32_t world_size,
                         uint32_t *out_rank);
/* Create the slab cfg->rank of cfg->world_size.  pos_xyz / ids: the n_local particles this slab owns at
 * t = 0 (as ws_slab_assign says) and their global ids; n_global: the reference's num_particles.  The
 * transport struct is copied; its ctx must outlive the handle. */
ws_status ws_slab_create(const ws_params *params, const float *pos_xyz, const uint32_t *ids, uint32_t n_local,
                         uint32_t n_global, const ws_device_cfg *cfg, const ws_transport *transport, ws_handle **out);
/* The particles this slab owns now (count varies with migration): up to cap records and their global
 * ids, in no particular order; *n_out = number owned.  Waits for enqueued steps. */
ws_status ws_slab_read_particles(ws_handle *h, ws_particle80 *out, uint32_t *out_ids, uint32_t cap, uint32_t *n_out);
/* COLLECTIVE re-cut: move the slab boundaries so that every slab owns about n_global / world_size particles again (cuts
 * stay on cell-layer boundaries; every slab keeps at least one layer).  ws_slab_assign's cuts give every slab the same
 * number of cell LAYERS, which is balanced while the fluid is spread evenly along x; a gravity with an x component
 * (the HUD can set any, src/hud.rs:151-162) piles it up at one end until that slab's capacity overruns.  A host that sees
 * the owned counts drift apart (ws_slab_counters / ws_num_particles) calls this on every rank at the same point; the
 * particles are redistributed like on a re-grid and the run continues bit-identically to a single handle.  Cheap
 * when nothing needs to move (one gather; every rank decides alike). */
ws_status ws_slab_rebalance(ws_handle *h);
/* (After a re-cut, a later ws_set_params that rebuilds the grid re-applies the equal-COUNT rule on the new grid instead of
 * falling back to equal layers.  After any load of a slab handle -- ws_reset, ws_write_particles, a re-grid, a re-cut --
 * the 80-byte record view reports zero density / pressure / acceleration until the next ws_step, as a freshly created
 * handle does; a single-GPU handle keeps the last step's values over a re-grid.) */
/* Host-only: the cuts ws_slab_rebalance chooses for an x-layer histogram (hist[nx] particles per cell layer; cuts_out holds
 * world_size + 1 entries, cuts_out[0] = 0, cuts_out[world_size] = nx, strictly increasing). */
ws_status ws_slab_balanced_cuts(const uint32_t *hist, uint32_t nx, uint32_t world_size, uint32_t *cuts_out);
/* Migration counters of this slab since it was created (cumulative over ws_reset, ws_write_particles, a re-grid and a
 * re-cut), as of the last migration that has run (waits for enqueued steps): out[0] = particles owned now, out[1] = particles that left, out[2] = particles that arrived, out[3] = of
 * those that left, the ones that crossed more than one slab in a step (the all-to-all route).  The reference is a
 * single-GPU program and has no counterpart; diagnostics for the host and for the tests. */
ws_status ws_slab_counters(ws_handle *h, uint64_t out[4]);

/* ---- reference-layout views of the sort (diagnostic; computed on demand by HIP
 *      kernels, never inside ws_step) ---------------------------------------------
 * What the reference's three index buffers hold after the same number of steps:
 *   keys[id]    = particle_cell_indicies: hash_cell(get_cell(predicted)) % N of the
 *                 predicted position the last step STARTED from
 *                 (assets/simulation.wgsl:121-141)
 *   perm[slot]  = particle_indicies: a permutation with keys[perm] ascending.  The
 *                 reference's bitonic network is unstable, so only the key SEQUENCE
 *                 keys[perm[.]] is comparable bit for bit; this library returns the
 *                 stable order (ties by ascending particle id).
 *   offsets[k]  = cell_offsets: first slot of key k or 999999999
 *                 (assets/bitonic_sort.wgsl:48-59)
 * Before the first step all three are the identity (src/fluid_compute.rs:306-308).
 * Any of the three output pointers may be NULL. */
ws_status ws_read_sort_view(ws_handle *h, uint32_t *keys_by_id, uint32_t *perm,
                            uint32_t *cell_offsets);

/* ---- introspection ------------------------------------------------------------- */
const char *ws_last_error(ws_handle *h);
uint32_t ws_num_particles(ws_handle *h);
uint64_t ws_steps_done(ws_handle *h);

/* Kernel ids for ws_profile_read (stable; names via ws_kernel_name). */
enum {
    WS_K_SCAN = 0,     /* cell-count exclusive scan (3 launches)                */
    WS_K_SCATTER = 1,  /* slot assignment inside each cell                      */
    WS_K_REORDER = 2,  /* stable in-cell order + physical SoA reorder           */
    WS_K_DENSITY = 3,  /* K4 update_density                                     */
    WS_K_FORCE = 4,    /* K5 update_pressure_force + K6 integrate + next K1 bin */
    WS_K_BIN = 5,      /* stand-alone cell binning (first step after upload)    */
    WS_K_COUNT = 6
};
const char *ws_kernel_name(uint32_t kernel_id);
/* With WS_FLAG_PROFILE: total milliseconds and launch count per kernel id since the
 * last ws_profile_reset, measured with HIP events on the handle's own stream.  Waits
 * for enqueued steps. */
ws_status ws_profile_read(ws_handle *h, uint32_t kernel_id, double *total_ms, uint64_t *launches);
ws_status ws_profile_reset(ws_handle *h);
/* Restrict WS_FLAG_PROFILE's events to the kernel ids whose bit is set in mask (default: all),
 * so that a timed region carries two events per step instead of two per kernel. */
ws_status ws_profile_select(ws_handle *h, uint32_t kernel_mask);
/* out[0] = cumulative particle-steps with more candidates than the accept mask holds (their waves took the full
 * sweep in the force kernel); out[1..3] = how many of the reference's cells (edge = smoothing radius) one cell of the
 * device grid spans along x, y, z -- 1 unless the reference-sized grid would exceed the cell budget (a small
 * smoothing radius in a big container), see ws_grid_dims; out[4] = steps replayed from a captured hipGraph
 * (WS_FLAG_GRAPH); slab handles: out[5] = the most particles one of this slab's boundary layers has held since the last
 * load and out[6] = the halo capacity it must stay under (ws_device_cfg.ghost_capacity), out[7] = the most particles
 * that left towards one neighbour in one step and out[8] = the migration message's capacity, out[9] = the most that crossed
 * more than one slab towards ONE destination rank in one step and out[10] = the capacity of a far message (one per destination); out[11..13] = the records the migration, halo
 * and far messages carry: with exact sizes (the default) what the LAST step's carried, with WS_FLAG_LAGGED_MESSAGES what
 * the NEXT step's will (sized from what every rank reported a few steps ago; the capacities with
 * WS_FLAG_FIXED_MESSAGES); 0 without peers; out[14] = how often ws_step has waited for message sizes so far (two per
 * step with exact sizes, never with WS_FLAG_LAGGED_MESSAGES); out[15] = 1 when the cost-guided tile schedule drives the
 * neighbour kernels (single-GPU handles of 2^18 <= n < 2^20 particles, not in a captured step: DESIGN.md 3). */
ws_status ws_read_stats(ws_handle *h, uint32_t out[16]);
/* Device cell grid actually in use (cells along x,y,z incl. padding).  The reference's N-bucket hashed table has the
 * same size for every smoothing radius (assets/simulation.wgsl:125-128); a dense grid does not, so when
 * container volume / h^3 exceeds the cell budget (max(16 N, 2^24) cells) the library merges cells along z, then y,
 * then x.  Results are unaffected (cell edges stay >= h; the distance test decides); no radius the reference accepts
 * makes ws_create / ws_set_params run out of table memory. */
ws_status ws_grid_dims(ws_handle *h, uint32_t dims[3]);

#ifdef __cplusplus
}
#endif
#endif
