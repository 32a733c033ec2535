// ws_internal.h -- shared between the C-ABI host code and the gfx950 kernels.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "wsfluid.h"

#include "ws_devhooks.h"

// Everything a kernel needs besides array pointers; passed by value (kernarg / SGPRs).
// Slab handles: where the force kernel's epilogue puts a particle whose new predicted position left the slab
// (one copy in device memory, WsDev::mig points at it)
struct WsMig {
    const uint32_t *cuts;  // world + 1 global x-layer cuts
    uint32_t world, me;
    uint32_t *dyn;         // DY_* words
    uint32_t *hole;        // indices the leavers vacate
    uint32_t hole_cap;
    uint32_t *sendL, *sendR;  // neighbour migration messages (header + mig_cap records of 64 B)
    uint32_t mig_cap;
    uint32_t *far;            // the messages (one per destination rank) for particles that cross several slabs
    uint32_t far_cap;
};

struct WsDev {
    // FluidStaticProps, src/fluid_compute.rs:41-51
    float dt, damping, h, target_density, pressure_scalar, near_pressure_scalar, viscosity;
    // SmoothingKernel, src/fluid_compute.rs:30-38
    float k_pow2, k_pow2_der, k_pow3, k_pow3_der, k_spikey;
    float grav[3];
    float ext_min[3], ext_max[3];
    // Largest f32 T with sqrtf(T) <= h: `!(d2 > T)` is bit-for-bit the reference's
    // `!(distance > smoothing_radius)` without taking the square root first.
    float d2_accept;
    // dense cell grid over the padded container: cell = floor(pred / h) (simulation.wgsl:121-123)
    int32_t org[3];  // grid origin in cell coordinates
    int32_t dim[3];  // grid cells along x, y, z (z fastest in memory, x slowest)
    // A grid cell is cm[c] of the reference's cells wide along axis c (1 everywhere unless the reference-sized grid
    // would not fit the cell budget: a small smoothing radius in a big container, ws_api.cpp derive_dev).  Edges stay
    // >= h, so the 27-cell search still covers every neighbour; candidates a wider cell adds fail the distance test.
    int32_t fdim[3];  // reference-sized cells along x, y, z of the GLOBAL grid (== dim when cm == 1 on one GPU)
    int32_t cm[3];
    uint32_t coarse;  // any cm != 1
    int32_t guard;   // guard entries in front of / behind cell_start
    uint32_t n;      // particles the kernels process: sorted indices [base, base + n)
    uint32_t base;   // first owned slot of the sorted arrays (0 on one GPU; ghosts sit in front of it in a slab)
    int32_t gdim_x;  // cells along x of the GLOBAL grid (== dim[0] on one GPU)
    int32_t xoff;    // global x index of local layer 0 (0 on one GPU)
    uint32_t ncells;
    uint32_t hash_n;  // the reference's `num_particles` in hash_cell (global N)
    // Slab handles only (nullptr / 0 on a single-GPU handle).  A slab's owned count changes with migration and is
    // known on the device alone: kernels are launched over `n` = a host-side UPPER BOUND and read the real count
    // from dyn[DY_N]; the density / force kernels cover the part of the owned range that range_sel names, resolved
    // on the device from the cell starts at lidx (ws_kernels.hip ws_range).
    const uint32_t *dyn;
    uint32_t range_sel;            // WS_RANGE_*
    uint32_t has_left, has_right;  // x-neighbours present
    uint32_t lidx[4];              // cell-start indices: layer 1 begin / end, layer nxl-2 begin / end
    const WsMig *mig;              // device copy; non-null = the force epilogue also does migration part 1
    uint32_t mig_limit;            // records the NEXT step's migration messages will carry (<= WsMig::mig_cap; 0 = all of it)
    uint32_t far_limit;            // ... and each of its far messages (<= WsMig::far_cap; 0 = all of it)
};

// words of the device block `dyn` of a slab handle
enum {
    DY_N = 0,        // owned particles
    DY_NHOLE,        // leavers of the step being migrated (all routes; per route they are counted in the message headers)
    DY_GL,           // ghosts staged in front of / behind the owned range
    DY_GR,
    DY_ERR,          // sticky WS_DYN_ERR_* bits
    DY_LEFT,         // cumulative: particles that left / arrived
    DY_ARRIVED,
    DY_FAR,          // cumulative: leavers that took the far route (crossed more than one slab in a step)
    DY_STEP,         // steps whose migration has run since the handle was created / reset (k_migrate_fill counts): the
                     // stamp of every message and the slot of the status ring come from here, not from the host, so a
                     // captured graph of the step replays unchanged
    DY_HALO_NOW,     // the larger of this step's two boundary-layer populations (k_halo_pack; k_migrate_fill hands it on and clears it)
    DY_PEAK_HALO,    // since the last load: the most particles a boundary layer of this slab held (records of a halo message)
    DY_PEAK_MIG,     // ... and the most particles that left towards ONE neighbour in one step (records of a migration message)
    DY_PEAK_FAR,     // ... and the most that crossed more than one slab in one step (records of the far message)
    // scratch of the multi-kernel migration fill (k_fill_*: a step that moves more than a few ten thousand particles)
    DY_F_FAR, DY_F_NTGT, DY_F_NSRC, DY_F_NOLD, DY_F_NNEW, DY_F_ARR, DY_F_LEAVE, DY_F_NL, DY_F_NR, DY_F_WANTED,
    WS_DYN_WORDS = 32
};
enum {
    WS_DYN_ERR_MIGRATION = 1u,  // more leavers than a migration message holds
    WS_DYN_ERR_CAPACITY = 2u,   // owned count would exceed the slab's capacity
    WS_DYN_ERR_HALO = 4u,       // a boundary layer holds more particles than a halo message
    WS_DYN_ERR_GHOSTS = 8u,     // more ghosts than the ghost range holds
    WS_DYN_ERR_STAMP = 16u,     // a message stamped with another step arrived: the transport delivered out of order
    WS_DYN_ERR_PRED = 32u       // an uploaded particle had to migrate with a predicted position its record cannot reproduce
};
enum { WS_RANGE_ALL = 0, WS_RANGE_EARLY = 1, WS_RANGE_LATE_LEFT = 2, WS_RANGE_LATE_RIGHT = 3, WS_RANGE_LATE_BOTH = 4 };
#define WS_HDR_WORDS_HOST 8u  // words of a message header (ws_kernels.hip WS_HDR_WORDS)
#define WS_MIG_REC_WORDS_HOST 8u  // = WS_MIG_REC_WORDS of ws_kernels.hip: {pos, id}, {vel, 0}
#define WS_HDR_UNKNOWN 0xFFFFFFFFu  // header words 4 / 5 before the first migration / halo of a particle set has been counted

// density / force kernel family (developer builds: WS_VARIANT=simple in the environment, for A/B tests)
enum { WS_VARIANT_SIMPLE = 0, WS_VARIANT_LISTED = 2 };

// Accept masks K4 writes and K5 walks: bit s of owned particle p (sorted order, p = index - base) = its s-th
// candidate in visit order is a neighbour.  words[w][p]: a wave's access to one word row is one coalesced segment.
struct WsMask {
    uint32_t *words;
    uint32_t stride;  // particles per row
};

// The particle set in the order of the last step: ONE 32-byte record {position, id | velocity, cell id} per particle
// (round 5; two arrays until then): k_reorder fetches a particle through its index with two 16-byte loads of ONE line
// (position, velocity and cell id used to be three scattered reads, a 128-byte line each once the fluid has mixed), and it
// is the shape a migrating particle travels in.  The force kernel's epilogue writes it with full-line stores (lane pairs
// exchange halves first: store_records).
struct WsSoA {
    float4 *pv;    // [2 i] = {position.xyz, particle id (bits)}, [2 i + 1] = {velocity.xyz, cell id (bits; = cid_cur[i])}
    float4 *pred;  // xyz = predicted_position: what an upload supplied; the step loop does not maintain it (k_reorder)
    // Optional (single-GPU handles): the particle's arrival rank inside its cell, as returned by the histogram's
    // atomic when the particle was binned.  With it the sort places particles without a second round of atomics.
    uint32_t *rank;
    __host__ __device__ float4 &pos(uint32_t i) const { return pv[2 * (size_t)i]; }
    __host__ __device__ float4 &vel(uint32_t i) const { return pv[2 * (size_t)i + 1]; }
};

// The test-only build (tests/libwsfluid_refcheck.so, -DWS_WITH_REFCHECK -Itests/refcheck) adds a validation mode
// whose declarations, kernels and host code all live under tests/refcheck/; the product build sees none of it.
#ifdef WS_WITH_REFCHECK
#include "ws_refcheck.h"
#endif

// planar copy of the sorted predicted positions (K4's radius tests)
// The cell-sorted copy.  Predicted position and velocity are interleaved: a neighbour's {pred.xyz, density}
// and {vel.xyz, near density} are the two halves of ONE 32-byte record, so the force kernel's two 16-B gathers per
// neighbour hit the same cache line (C3 dense state: -17 % kernel time against two separate arrays), and a halo
// message is one range.
struct WsSorted {
    float4 *pos;  // xyz = position, w = particle id (bits); only the integrator reads it
    float4 *pv;   // [2j] = {pred.xyz, density after K4}, [2j + 1] = {vel.xyz, near density after K4}
    __host__ __device__ float4 &pred(uint32_t j) const { return pv[2 * (size_t)j]; }
    __host__ __device__ float4 &vel(uint32_t j) const { return pv[2 * (size_t)j + 1]; }
    // The gathers of the force kernel's inner loop: a 32-bit byte offset from the (uniform) base lets the load
    // take its base from scalar registers, with no 64-bit address arithmetic per neighbour.  Needs
    // 32 * slots < 2^32, which ws_create / ws_slab_create check (WS_MAX_SLOTS).
    __device__ const float4 &pred_near(uint32_t j) const
    {
        return *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(pv) + j * 32u);
    }
    __device__ const float4 &vel_near(uint32_t j) const
    {
        return *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(pv) + j * 32u + 16u);
    }
};
#define WS_MAX_SLOTS (1u << 27)

struct WsXYZ {
    float *x = nullptr, *y = nullptr, *z = nullptr;
};

// Tile schedule of a full-range launch of K4 / K5 (single-GPU handles).  Workgroup b works for XCD b & 7 (the hardware
// deals workgroups round-robin over the XCDs) and takes the (b >> 3)-th entry of that XCD's part of `perm`:
// perm[split[x] .. split[x + 1]) = the tiles of one contiguous x-slab of the sorted order (so an XCD's L2 keeps seeing one
// neighbourhood), slabs cut so that every XCD gets the same COST, tiles inside a slab in descending cost classes (the
// launch drains through its cheapest tiles, not through whichever came last).  cost[tile] = how long the tile's
// workgroup ran in the previous step (100 MHz ticks), written by the kernels themselves; k_schedule turns it into
// split / perm beside the next step's sort phase.  Any schedule computes the same results.
struct WsSched {
    uint32_t *split = nullptr;  // 9 words
    uint32_t *perm = nullptr;   // ntiles words
    uint32_t *cost = nullptr;   // ntiles words
};

struct WsEventPair {
    uint32_t kernel;
    hipEvent_t a, b;
    bool counts = true;  // false: the second launch of a kernel id within one step (time is added, the launch count is per step)
};

struct ws_handle {
    int device = 0;
    uint32_t flags = 0;
    uint32_t prof_mask = 0xFFFFFFFFu;
    uint32_t n = 0;
    uint64_t steps = 0;
    ws_params params{};
    ws_smoothing_kernel sk{};
    WsDev dev{};
    hipStream_t stream = nullptr;
    hipEvent_t done = nullptr;
    bool done_recorded = false;

    WsSoA cur{};   // state in the order of the last step (written by the force kernel)
    WsSorted srt{};  // cell-sorted copy the density/force kernels read
    WsXYZ sxyz;                   // planar predicted positions in `srt` order
    uint32_t *cid_cur = nullptr;  // cell id per particle of `cur`
    uint32_t *cid_srt = nullptr;  // cell id per particle of `srt`
    int variant = WS_VARIANT_LISTED;  // density / near density ride in srt.pred(i).w / srt.vel(i).w
    float4 *accel = nullptr;      // acceleration in `srt` order
    uint32_t *slot_tmp = nullptr; // particle index per tentative slot
    uint32_t *count = nullptr;    // per-cell particle count (histogram)
    uint32_t *cursor = nullptr;   // per-cell fill cursor
    uint32_t *start = nullptr;    // guard + ncells + 1 + guard exclusive starts
    uint32_t *start_alloc = nullptr;  // its allocation (start sits 0..3 words in: see alloc_grid)
    uint32_t *bsum = nullptr;     // scan state (ticket + tile descriptors), zeroed at allocation
    WsMask mask = {nullptr, 0};   // accept masks (listed variant)
    bool ieee = false;            // WS_FLAG_IEEE_DIVISION: correctly rounded sqrt / division in the pair terms
    uint32_t *stats = nullptr;    // device counters: [0] particle-steps with more candidates than the mask holds
    uint8_t *mult = nullptr;      // 27 stencil multiplicities (hash aliasing), device
    bool alias = false;
    size_t grid_alloc_cells = 0;

    // staging for uploads / readback
    void *stage = nullptr;
    size_t stage_bytes = 0;
    // asynchronous position readback (ws_read_positions_begin / _end): own staging, own copy stream
    hipStream_t copy_stream = nullptr;
    hipEvent_t rb_gathered = nullptr, rb_done = nullptr;
    float *rb_stage = nullptr;
    size_t rb_bytes = 0;
    bool rb_inflight = false;
    // ws_read_positions_begin(h, NULL): two page-locked buffers the library owns, filled alternately
    float *rb_host[2] = {nullptr, nullptr};
    size_t rb_host_bytes = 0;
    int rb_fill = 0, rb_last = -1;  // the buffer the readback in flight fills; the one the last finished readback filled
    bool rb_owned = false;          // the readback in flight goes into rb_host[rb_fill]

    // reference-layout sort view (lazy)
    uint32_t *v_keys = nullptr, *v_perm = nullptr, *v_tmp = nullptr, *v_count = nullptr,
             *v_cursor = nullptr, *v_start = nullptr, *v_bsum = nullptr, *v_off = nullptr;

    // profiling
    std::vector<WsEventPair> pending;
    std::vector<hipEvent_t> pool;
    double prof_ms[WS_K_COUNT] = {0};
    uint64_t prof_cnt[WS_K_COUNT] = {0};

#ifdef WS_WITH_REFCHECK
    bool refmode = false;  // WS_FLAG_REFERENCE_ORDER (test-only build)
    WsRef ref;
#endif

    // Cost-guided tile schedule of the two neighbour kernels (single-GPU handles; ws_kernels.hip "tile schedule").
    // Two sets: step t's kernels read set t & 1 while k_schedule, running beside K5(t) on a stream of its own, writes
    // set (t + 1) & 1 from K4(t)'s costs and K5(t - 1)'s.
    WsSched sched4[2], sched5[2];          // K4's (64-particle tiles) and K5's (128-particle tiles); cost4 is shared
    uint32_t sched_tiles4 = 0, sched_tiles5 = 0;
    bool sched_on = false;                 // the schedule arrays exist and describe the current particle count
    uint32_t sched_classes = 8, sched_group = 64;  // cost classes; particles per group of tiles that is classed as one
    uint32_t sched_parity = 0;             // the set the NEXT step's kernels read
    hipStream_t sched_stream = nullptr;
    hipEvent_t ev_sched_in = nullptr, ev_sched_out = nullptr;

    // WS_FLAG_GRAPH on a single-GPU handle: the captured steady-state step
    hipGraphExec_t graph_exec = nullptr;
    bool graph_failed = false;
    uint64_t graph_steps = 0;

    // slab (multi-GPU) state; slab == nullptr on a single-GPU handle
    bool accel_stale = false;  // accel[] is behind the last step (computed on demand: refresh_accel)
    bool pred_stale = false;  // cur.pred is behind cur.pos / cur.vel (the step loop does not store it: k_reorder)
    struct WsSlab *slab = nullptr;
    bool own_stream = true;
    bool dead = false;  // a re-grid failed after the old tables were given up: every later ws_step refuses (err says why)

    std::string err;
};

struct WsGraphEntry {
    uint32_t n_bound = 0, mig_cur = 0, mig_next = 0, halo = 0, far_cur = 0, far_next = 0;  // what a captured step has baked in
    hipGraphExec_t exec = nullptr;
};
struct WsSlab {
    ws_transport tr{};
    uint32_t rank = 0, world = 1;
    uint32_t n_global = 0;
    // fixed capacities, known to both ends of every message
    uint32_t cap = 0;        // owned particles
    uint32_t halo_cap = 0;   // particles of one boundary layer = ghosts per side = records of a halo message
    uint32_t mig_cap = 0;    // records of a neighbour migration message
    uint32_t far_cap = 0;    // records of ONE far message (one per destination rank) for particles that cross several slabs
    uint32_t hole_cap = 0;   // leavers per step (all routes)
    uint32_t max_arrivals = 0;
    // Message SIZES follow the fluid (round 4).  The buffers keep their fixed capacities; what travels each step is a
    // prefix sized from what every rank reported three to four steps ago (header words 4 / 5 / 6 of the far
    // message = the status table): the same table on every rank, hence the same size at both ends of every exchange.
    uint32_t mig_limit_cur = 0;       // records of the migration messages exchanged by the step being enqueued
    uint32_t mig_limit_next = 0;      // ... by the next one (the force kernel of this step fills them: WsDev::mig_limit)
    uint32_t far_limit_cur = 0, far_limit_next = 0, want_far = 0;  // the same for the far messages
    uint32_t halo_limit = 0, halo_limit_next = 0;  // records of this step's / the next step's halo messages
    uint32_t want_mig = 0, want_halo = 0;   // maxima over all ranks (header words 4 / 5) and over the last eight tables
    uint32_t want_ring[3][8] = {};          // the last eight tables' maxima: migration, halo, far
    uint64_t arrivals_hist[4] = {0, 0, 0, 0};  // upper bounds of the arrivals of the last four steps (launch bound)
    uint32_t floor_mig = 4096, floor_far = 256, floor_halo = 8192;  // no message is sized below these (records)
    // WS_FLAG_EXACT_MESSAGES: every message carries exactly what its sender has for it -- ws_step waits for the counts
    // (one all-gather of four words per rank before the migration, one before the halos, behind the early K4)
    bool exact_messages = false;
    uint32_t *xs_send[2] = {nullptr, nullptr}, *xs_all[2] = {nullptr, nullptr}, *xs_host[2] = {nullptr, nullptr};  // [migration | halo]
    hipEvent_t ev_sizes[2] = {nullptr, nullptr};
    uint32_t *far_pack = nullptr;     // the far messages at the stride of what travels
    uint32_t exact_now[3] = {0, 0, 0};  // the records the last step's migration / halo / far messages carried
    uint32_t size_waits = 0;            // host waits for message sizes so far (two per step with exact sizes)
    uint32_t limit_hold = 0;          // steps for which the limits stay at the full capacities (after a load / parameter change)
    bool fixed_messages = false;      // WS_FLAG_FIXED_MESSAGES (and a captured step with peers): always the full capacities
    uint32_t mode_word = 0;           // every choice that fixes the step's collective sequence or its results: all ranks must agree (slab_agree_mode)
    std::vector<uint32_t> cuts;       // world + 1 global x-layer cuts
    uint32_t *cuts_dev = nullptr;
    uint32_t *dyn = nullptr;          // WS_DYN_WORDS device words (DY_*)
    WsMig *mig_dev = nullptr;         // device copy of the migration targets (WsDev::mig)
    uint32_t *hole = nullptr, *tgt = nullptr, *src = nullptr;  // migration index lists
    uint32_t *mig_sendL = nullptr, *mig_sendR = nullptr, *mig_recvL = nullptr, *mig_recvR = nullptr;
    uint32_t *far_send = nullptr, *far_all = nullptr;
    uint32_t *halo_sendL = nullptr, *halo_sendR = nullptr, *halo_recvL = nullptr, *halo_recvR = nullptr;
    bool binned = false;              // cid_cur / count describe the current owned set
    // what the host knows about the device state, two steps late
    uint32_t *status_dev = nullptr, *status_host = nullptr;  // ring of per-rank header rows (pinned on the host)
    hipEvent_t ev_status[4] = {nullptr, nullptr, nullptr, nullptr};
    uint32_t n_known = 0;             // owned count as of step n_known_step
    uint64_t n_known_step = 0;
    uint64_t t0 = 0;                  // h->steps when the owned set was last loaded (create / reset / write / re-grid):
                                      // status tables of earlier steps describe another particle set
    uint32_t cfg_ghost_capacity = 0;  // ws_device_cfg.ghost_capacity as given (0 = derive from the grid)
    // id-ordered global views (ws_read_positions & co. on a slab handle): count words, packed records of this rank,
    // every rank's records, the id-ordered result
    uint32_t *cnt_send = nullptr, *cnt_all = nullptr;
    uint32_t *g_send = nullptr, *g_all = nullptr, *g_out = nullptr;
    size_t g_send_bytes = 0, g_all_bytes = 0, g_out_bytes = 0;
    uint32_t *hist_dev = nullptr;     // x-layer histogram of a re-grid (one buffer, grown by doubling)
    uint32_t hist_cap = 0;
    uint32_t *agree_send = nullptr, *agree_all = nullptr;  // one word per rank: verdicts and the mode word (slab_agree_words)
    std::vector<uint32_t *> retired;  // outgrown gather buffers (freed with the handle: ws_slab.inc slab_grow)
    std::vector<uint32_t> counts;     // owned particles per rank as of the last gather
    std::vector<uint32_t> caps;       // ... and every rank's owned capacity (a re-grid decides for all ranks alike)
    uint64_t left_base = 0, arrived_base = 0, far_base = 0;  // cumulative counters carried over the loads (ws_slab_counters)
    bool balanced = false;            // the cuts come from ws_slab_rebalance (a re-grid re-balances on the new grid)
    uint32_t failed = 0;              // sticky WS_DYN_ERR_* bits seen on any rank
    bool comm_failed = false;         // a transport call failed: the handle is dead
    // halo / compute overlap: the halos travel on `comm` while the particles that need no ghosts compute
    hipStream_t comm = nullptr, copy = nullptr;
    hipEvent_t ev_sorted = nullptr, ev_halo_a = nullptr, ev_k4_late = nullptr, ev_halo_b = nullptr, ev_filled = nullptr;
    bool overlap = true;              // WS_FLAG_NO_OVERLAP turns it off
    bool last_split = false;          // the last step ran K4 / K5 as early + late launches (refresh_accel repeats that)
    // WS_FLAG_GRAPH: captured steps, one per (quantised) launch bound
    std::vector<struct WsGraphEntry> graphs;
    bool graph_failed = false;        // the step could not be captured (transport / runtime): direct launches
    bool capturing = false;           // a capture is being recorded: a transport refusal is not a communication failure
    uint64_t graph_steps = 0;         // steps replayed from a graph
    hipEvent_t ev_copied = nullptr;
    // cumulative statistics (refreshed by ws_slab_read_particles)
    uint64_t migrated_out = 0;
};

// ws_rccl.cpp (not part of the public header): no-op for a transport that is not the library's own
extern "C" void ws_rccl_transport_bind_stream(const ws_transport *t, void *stream);

// ---- kernel launchers (ws_kernels.hip) -------------------------------------------
void wsk_upload_positions(hipStream_t s, const float *xyz_dev, WsSoA cur, uint32_t n);
void wsk_upload_particles(hipStream_t s, const ws_particle80 *in_dev, WsSoA cur, uint32_t n);
void wsk_bin(hipStream_t s, const WsDev &d, WsSoA cur, uint32_t *cid, uint32_t *count);  // (cur / cid from the first particle to bin)
void wsk_place(hipStream_t s, const WsDev &d, const uint32_t *cid, const uint32_t *rank, const uint32_t *start, uint32_t *slot_tmp);
void wsk_scan(hipStream_t s, uint32_t *count, uint32_t *start_body, uint32_t *cursor, uint32_t *state, uint32_t nitems,
              bool zero_count, uint32_t base);
uint32_t wsk_scan_state_words(uint32_t nitems);
void wsk_scatter(hipStream_t s, const uint32_t *keys, uint32_t *cursor, uint32_t *slot_tmp, uint32_t n, const uint32_t *n_dev);
void wsk_reorder(hipStream_t s, const WsDev &d, const uint32_t *slot_tmp, const uint32_t *cid_cur, const uint32_t *start, WsSoA cur,
                 WsSorted srt, uint32_t *cid_srt, WsXYZ sxyz, bool recompute_pred);
void wsk_refresh_pred(hipStream_t s, const WsDev &d, WsSoA cur);
void wsk_density(hipStream_t s, const WsDev &d, const uint32_t *start, const uint32_t *cid_srt, WsSorted srt,
                 const uint8_t *mult, bool alias, int variant, bool ieee, uint32_t *stats, WsMask mask, WsXYZ sxyz,
                 const WsEventPair *ev = nullptr, WsSched sched = WsSched{});
uint32_t wsk_density_tile(void);  // particles per tile of the listed K4 / K5
uint32_t wsk_force_tile(uint32_t n);  // particles per K5 workgroup in a launch over n particles
uint32_t wsk_sched_grid(uint32_t ntiles);  // workgroups of a scheduled launch over ntiles tiles
// s5_costs: the set whose cost array holds the K5 costs to schedule from (the set being written is read by nobody)
void wsk_schedule(hipStream_t s, WsSched s4, uint32_t ntiles4, WsSched s5, WsSched s5_costs, uint32_t ntiles5, uint32_t nclasses,
                  uint32_t group_particles, uint32_t tile5);
uint32_t wsk_mask_words(void);
void wsk_force(hipStream_t s, const WsDev &d, const uint32_t *start, const uint32_t *cid_srt, WsSorted srt, WsSoA out,
               float4 *accel, uint32_t *cid_out, uint32_t *count, const uint8_t *mult, bool alias, int variant, bool ieee,
               WsMask mask, bool accel_only, const WsEventPair *ev = nullptr, WsSched sched = WsSched{});
void wsk_gather_positions(hipStream_t s, WsSoA cur, float *out_xyz, uint32_t n);
void wsk_gather_speeds(hipStream_t s, WsSoA cur, float *out, uint32_t n);
void wsk_gather_particles(hipStream_t s, const WsDev &d, WsSoA cur, WsSorted srt, const float4 *accel, bool have_step,
                          ws_particle80 *out, uint32_t n);
// reference-layout view
void wsk_view_keys(hipStream_t s, const WsDev &d, WsSorted srt, uint32_t *keys_by_id, uint32_t *count);
void wsk_view_count(hipStream_t s, const uint32_t *keys, uint32_t *count, uint32_t n);
void wsk_view_fix(hipStream_t s, const uint32_t *tmp, const uint32_t *keys, const uint32_t *start,
                  uint32_t *perm, uint32_t n);
void wsk_view_offsets(hipStream_t s, const uint32_t *start, uint32_t *off, uint32_t n);
void wsk_iota(hipStream_t s, uint32_t *p, uint32_t n);
void wsk_set_words4(hipStream_t s, uint32_t *p, uint32_t a, uint32_t b, uint32_t c, uint32_t d);
// slabs
void wsk_migrate_mark(hipStream_t s, const WsDev &d, WsSoA cur, uint32_t *cid_cur, uint32_t *count);
void wsk_migrate_fill(hipStream_t s, const WsDev &d, uint32_t world, uint32_t me, uint32_t cap, uint32_t *dyn,
                      const uint32_t *hole, const uint32_t *recvL, const uint32_t *recvR, uint32_t mig_cap,
                      const uint32_t *far_all, uint32_t far_cap, uint32_t *tgt, uint32_t *src, WsSoA cur, uint32_t *cid_cur,
                      uint32_t *count, uint32_t *status_ring, uint32_t status_slots, uint32_t hole_cap, uint32_t *sendL,
                      uint32_t *sendR, uint32_t *far_send, uint32_t far_next, uint64_t leave_bound);
void wsk_far_seal(hipStream_t s, uint32_t world, uint32_t far_cur, uint32_t *far_send, uint32_t *dyn);
void wsk_sizes(hipStream_t s, const uint32_t *a, const uint32_t *b, const uint32_t *c, const uint32_t *dyn, uint32_t *out);
void wsk_far_pack(hipStream_t s, uint32_t world, uint32_t far_cap, uint32_t far_n, const uint32_t *src, uint32_t *dst);
void wsk_halo_pack(hipStream_t s, const WsDev &d, const uint32_t *start, WsSorted srt, uint32_t *dyn, uint32_t rowy,
                   uint32_t halo_cap, uint32_t *sendL, uint32_t *sendR, bool densities);
void wsk_halo_unpack(hipStream_t s, const WsDev &d, uint32_t *start, WsSorted srt, WsXYZ sxyz, uint32_t *dyn, uint32_t rowy,
                     uint32_t nxl, uint32_t ghost_cap, const uint32_t *recvL, const uint32_t *recvR, bool densities);
void wsk_gather_slab(hipStream_t s, const WsDev &d, WsSoA cur, WsSorted srt, const float4 *accel, bool have_step,
                     ws_particle80 *out, uint32_t *ids);
void wsk_upload_positions_ids(hipStream_t s, const float *xyz_dev, const uint32_t *ids_dev, WsSoA cur, uint32_t n);
// id-ordered global views / loads of slab handles (ws_kernels.hip "slab handles: the id-ordered GLOBAL views")
enum { WS_PACK_POS_H = 0, WS_PACK_SPEED_H = 1, WS_PACK_RECORD_H = 2, WS_PACK_STATE_H = 3, WS_PACK_KEY_H = 4 };
uint32_t wsk_pack_words(int kind);  // payload words per record (the record carries one more: the particle id)
void wsk_slab_pack(hipStream_t s, const WsDev &d, int kind, WsSoA cur, WsSorted srt, const float4 *accel, bool have_step,
                   uint32_t *out);
void wsk_slab_unpack_by_id(hipStream_t s, const uint32_t *all, const uint32_t *cnt, uint32_t world, uint32_t max_n,
                           size_t stride_words, uint32_t pw, uint32_t n_global, uint32_t *out);
void wsk_slab_layer_hist(hipStream_t s, const WsDev &d, const uint32_t *all, const uint32_t *cnt, uint32_t world, uint32_t max_n,
                         size_t stride_words, uint32_t *hist);
void wsk_slab_select(hipStream_t s, const WsDev &d, int src, const uint32_t *cuts, uint32_t world, uint32_t me,
                     const void *chunk, uint32_t id0, uint32_t m, const uint32_t *cnt, size_t stride_words, WsSoA cur,
                     uint32_t cap, uint32_t *dyn);
