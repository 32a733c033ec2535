// ws_api.cpp -- the C ABI of include/wsfluid.h: handle lifetime, device buffers, the
// per-step launch sequence and the readback paths.  Host-only logic; all device work is
// in ws_kernels.hip.  The role played here is the one bevy_app_compute's
// AppComputeWorker plays in the reference (src/fluid_compute.rs:277-366,:393-397).
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <new>

#include "ws_internal.h"

namespace {

constexpr int kGridPad = 2;                     // empty cell layers around the container
constexpr uint64_t kMaxCells = 1ull << 29;      // hard cap of the cell tables (3 x u32 per cell -> 6 GiB); see derive_dev

thread_local std::string g_create_error;

// ws_slab.inc
void slab_free(ws_handle *h);
void slab_drop_graphs(WsSlab *S);
ws_status slab_step(ws_handle *h);
ws_status slab_settle(ws_handle *h);
ws_status slab_read_by_id(ws_handle *h, int kind, void *out);
ws_status slab_gather_by_id(ws_handle *h, int kind);
ws_status slab_reset(ws_handle *h, const float *pos_xyz);
ws_status slab_write_particles(ws_handle *h, const ws_particle80 *in);
ws_status slab_regrid(ws_handle *h, const ws_params *params, bool rebalance);

ws_status fail(ws_handle *h, ws_status st, const char *what, hipError_t e = hipSuccess)
{
    char buf[512];
    if (e != hipSuccess)
        snprintf(buf, sizeof buf, "%s: %s (%s)", what, hipGetErrorString(e), hipGetErrorName(e));
    else
        snprintf(buf, sizeof buf, "%s", what);
    if (h)
        h->err = buf;
    else
        g_create_error = buf;
    return st;
}

#define HIP_TRY(h, expr)                                                                     \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess)                                                                \
            return fail(h, e_ == hipErrorOutOfMemory ? WS_ERR_OUT_OF_MEMORY : WS_ERR_HIP, #expr, e_); \
    } while (0)

// A host <-> device copy on the HANDLE'S OWN stream, complete on return.  Never the legacy (null) stream: it
// synchronises implicitly with streams that are none of this handle's business, and it fails outright
// (hipErrorStreamCaptureImplicit) while any other thread of the host is capturing a hipGraph -- another handle with
// WS_FLAG_GRAPH, the host's own graphs.  Ordered behind whatever the handle has enqueued.
hipError_t copy_now(ws_handle *h, void *dst, const void *src, size_t bytes, hipMemcpyKind kind)
{
    const hipError_t e = hipMemcpyAsync(dst, src, bytes, kind, h->stream);
    return e != hipSuccess ? e : hipStreamSynchronize(h->stream);
}

// Rust f32::powi -> llvm.powi -> compiler-rt __powisf2 (square-and-multiply).
float powi_f32(float a, int b)
{
    float r = 1.0f;
    for (;;) {
        if (b & 1) r *= a;
        b /= 2;
        if (b == 0) break;
        a *= a;
    }
    return r;
}

// Largest float T with sqrtf(T) <= h, so that (d2 > T) == (sqrtf(d2) > h) for every d2.
float accept_threshold(float h)
{
    float t = h * h;
    while (sqrtf(t) > h) t = nextafterf(t, 0.0f);
    for (;;) {
        const float up = nextafterf(t, INFINITY);
        if (sqrtf(up) <= h)
            t = up;
        else
            break;
    }
    return t;
}

bool is_pow2(uint32_t n) { return n && !(n & (n - 1)); }

// hash_cell before its `% N`, assets/simulation.wgsl:125-128 (u32 wrap arithmetic)
uint32_t ref_linear(int x, int y, int z)
{
    return (uint32_t)x * 15823u + (uint32_t)y * 9737333u + (uint32_t)z * 440817757u;
}

uint32_t ref_hash_delta(int x, int y, int z, uint32_t n) { return ref_linear(x, y, z) % n; }

ws_status validate_params(ws_handle *h, const ws_params *p)
{
    if (!p) return fail(h, WS_ERR_INVALID_ARG, "params is NULL");
    if (!(p->smoothing_radius > 0.0f) || !isfinite(p->smoothing_radius))
        return fail(h, WS_ERR_INVALID_ARG, "smoothing_radius must be finite and > 0");
    for (int c = 0; c < 3; c++) {
        if (!isfinite(p->ext_min[c]) || !isfinite(p->ext_max[c]) || p->ext_min[c] > p->ext_max[c])
            return fail(h, WS_ERR_INVALID_ARG, "container ext_min/ext_max must be finite with ext_min <= ext_max");
    }
    return WS_OK;
}

// Fill the by-value kernel parameter block from the ABI params.
ws_status derive_dev(ws_handle *h, const ws_params &p, uint32_t n, WsDev *out)
{
    WsDev d{};
    d.dt = p.delta_time;
    d.damping = p.collision_damping;
    d.h = p.smoothing_radius;
    d.target_density = p.target_density;
    d.pressure_scalar = p.pressure_scalar;
    d.near_pressure_scalar = p.near_pressure_scalar;
    d.viscosity = p.viscosity_strength;
    ws_smoothing_kernel sk;
    ws_get_smoothing_kernel(&p, &sk);
    d.k_pow2 = sk.pow2;
    d.k_pow2_der = sk.pow2_der;
    d.k_pow3 = sk.pow3;
    d.k_pow3_der = sk.pow3_der;
    d.k_spikey = sk.spikey_pow3;
    for (int c = 0; c < 3; c++) {
        d.grav[c] = p.gravity[c];
        d.ext_min[c] = p.ext_min[c];
        d.ext_max[c] = p.ext_max[c];
    }
    d.d2_accept = accept_threshold(d.h);
    // Reference-sized cells (edge h) over the container padded by kGridPad cells.  Cell coordinates are carried as
    // f32 in the kernels (floorf(x / h) - org): keep them exactly representable.
    uint64_t cells = 1;
    for (int c = 0; c < 3; c++) {
        const double lo = floor((double)floorf(p.ext_min[c] / d.h)) - kGridPad;
        const double hi = floor((double)floorf(p.ext_max[c] / d.h)) + kGridPad;
        const double dim = hi - lo + 1.0;
        if (!(fabs(lo) < 8388608.0) || !(fabs(hi) < 8388608.0))
            return fail(h, WS_ERR_INVALID_ARG, "container / smoothing_radius out of range (more than 2^23 cells along an axis)");
        d.org[c] = (int32_t)lo;
        d.fdim[c] = d.dim[c] = (int32_t)dim;
        d.cm[c] = 1;
        cells *= (uint64_t)d.dim[c];  // < 2^69 / overflow-free: three factors below 2^24
    }
    // The cell budget.  The reference's N-bucket hashed table (simulation.wgsl:125-128) costs the same whatever the
    // smoothing radius; a dense grid of reference-sized cells does not (two HUD key presses take h from 0.25 to 0.05,
    // src/hud.rs:135-138: 125 x the cells).  When the reference-sized grid exceeds the budget, grid cells are MERGED
    // along z first (a (dx, dy) column stays one contiguous particle run, only longer), then y, then x, until it
    // fits: every cell edge stays >= h, so the 27-cell search still sees every neighbour, the order inside the
    // runs stays canonical (multi-GPU bit-identity), and the only cost is more candidates for the distance test.
    // No smoothing radius the reference accepts runs out of table memory.
    uint64_t budget = std::max<uint64_t>(16ull * n, 1ull << 24);
    if (const char *v = WS_DEV_ENV("WS_CELL_BUDGET")) budget = std::max<uint64_t>(strtoull(v, nullptr, 10), 64);  // developer builds: force the merge on small domains
    budget = std::min<uint64_t>(budget, kMaxCells);
    for (int c = 2; c >= 0 && cells > budget; c--) {
        const uint64_t others = cells / (uint64_t)d.dim[c];
        // smallest merge factor that brings this axis (and with it the grid) under the budget, at most the whole axis
        uint64_t want = std::max<uint64_t>(budget / std::max<uint64_t>(others, 1), 1);  // grid cells this axis may keep
        uint64_t m = ((uint64_t)d.fdim[c] + want - 1) / want;
        // y and z keep at least three grid cells: the nine runs of a particle are spans [cc - 1, cc + 1] of the
        // LINEAR cell index around cc = c + dx * rowy + dy * rowz, which are disjoint only if dim2 >= 3 and dim1 >= 3
        // (fewer, and z + 1 of one (dx, dy) column IS z - 1 of the next: neighbours would be counted twice)
        const uint64_t m_max = c == 0 ? (uint64_t)d.fdim[c] : (uint64_t)d.fdim[c] / 3;
        m = std::min<uint64_t>(std::max<uint64_t>(m, 1), std::max<uint64_t>(m_max, 1));
        d.cm[c] = (int32_t)m;
        d.dim[c] = (int32_t)(((uint64_t)d.fdim[c] + m - 1) / m);
        cells = others * (uint64_t)d.dim[c];
    }
    d.coarse = (d.cm[0] != 1 || d.cm[1] != 1 || d.cm[2] != 1) ? 1u : 0u;
    if (cells > kMaxCells) return fail(h, WS_ERR_OUT_OF_MEMORY, "cell grid too large even with merged cells");
    d.ncells = (uint32_t)cells;
    d.guard = d.dim[1] * d.dim[2] + d.dim[2] + 1;
    d.n = n;
    d.base = 0;
    d.gdim_x = d.dim[0];
    d.xoff = 0;
    d.hash_n = n;
    h->sk = sk;
    *out = d;
    return WS_OK;
}

void free_grid(ws_handle *h)
{
    hipFree(h->count);
    hipFree(h->cursor);
    hipFree(h->start_alloc);
    hipFree(h->bsum);
    h->count = h->cursor = h->start = h->start_alloc = h->bsum = nullptr;
    h->grid_alloc_cells = 0;
}

ws_status alloc_grid(ws_handle *h)
{
    const WsDev &d = h->dev;
    // (developer builds: WS_FAIL_REGRID=1 makes a RE-grid's allocation fail -- the tests' way to a dead handle)
    if (h->steps > 0 && WS_DEV_ENV("WS_FAIL_REGRID")) return fail(h, WS_ERR_OUT_OF_MEMORY, "cell tables: allocation failure forced by WS_FAIL_REGRID");
    if (h->grid_alloc_cells < d.ncells) {
        free_grid(h);
        HIP_TRY(h, hipMalloc(&h->count, (size_t)d.ncells * 4));
        HIP_TRY(h, hipMalloc(&h->cursor, (size_t)d.ncells * 4));
        h->grid_alloc_cells = d.ncells;
    }
    // start is sized by the guard too, which depends on the dims: always (re)build it
    hipFree(h->start_alloc);
    hipFree(h->bsum);
    h->start = h->start_alloc = h->bsum = nullptr;
    const size_t nstart = (size_t)d.ncells + 2 * (size_t)d.guard + 2;
    // the scan writes start's body (from entry `guard`) 16 B at a time: shift the array so that entry is aligned
    HIP_TRY(h, hipMalloc(&h->start_alloc, (nstart + 4) * 4));
    h->start = h->start_alloc + (4u - (uint32_t)d.guard % 4u) % 4u;
    HIP_TRY(h, hipMalloc(&h->bsum, (size_t)wsk_scan_state_words(d.ncells) * 4));
    HIP_TRY(h, hipMemsetAsync(h->bsum, 0, (size_t)wsk_scan_state_words(d.ncells) * 4, h->stream));  // same stream as its users
    // constant parts of cell_start: front guard = 0, [ncells] and the back guard = n
    HIP_TRY(h, hipMemsetAsync(h->start, 0, (size_t)d.guard * 4, h->stream));
    std::vector<uint32_t> tail((size_t)d.guard + 2, d.n);
    HIP_TRY(h, hipMemcpyAsync(h->start + d.guard + d.ncells, tail.data(), tail.size() * 4, hipMemcpyHostToDevice,
                              h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    HIP_TRY(h, hipMemsetAsync(h->count, 0, (size_t)d.ncells * 4, h->stream));
    return WS_OK;
}

// Does the reference's N-bucket table put two cells of one 27-stencil into the same bucket?  Then a neighbour
// is visited once per aliasing stencil cell (the reference counts it that often) and the kernels take the
// multiplicity path (`alias`).  Power-of-two N: hash_cell is linear mod N, the multiplicity depends on the
// cell difference only and is tabulated here.  Any other N: `% N` follows a wrap mod 2^32, so two stencil cells
// whose linear forms differ by delta can alias iff delta, delta - 2^32 or delta + 2^32 is a multiple of N (which
// of them applies depends on the cell); if none of the 27 x 26 differences qualifies no particle can ever see an
// aliased bucket, otherwise the kernels count the aliasing offsets per pair (alias_mult).
ws_status upload_mult(ws_handle *h)
{
    uint8_t m[27];
    bool alias = false;
    const uint32_t n = h->dev.hash_n;
    for (int dx = -1; dx <= 1; dx++)
        for (int dy = -1; dy <= 1; dy++)
            for (int dz = -1; dz <= 1; dz++) {
                uint32_t cnt = 0;
                for (int ox = -1; ox <= 1; ox++)
                    for (int oy = -1; oy <= 1; oy++)
                        for (int oz = -1; oz <= 1; oz++) {
                            if (is_pow2(n)) {
                                if (ref_hash_delta(ox, oy, oz, n) == ref_hash_delta(dx, dy, dz, n)) cnt++;
                                continue;
                            }
                            if (ox == dx && oy == dy && oz == dz) {
                                cnt++;
                                continue;
                            }
                            const int64_t delta = (int64_t)ref_linear(ox, oy, oz) - (int64_t)ref_linear(dx, dy, dz);
                            for (int64_t k = -1; k <= 1; k++)
                                if ((delta + k * ((int64_t)1 << 32)) % (int64_t)n == 0) alias = true;
                        }
                if (cnt > 1) alias = true;
                m[(dx + 1) * 9 + (dy + 1) * 3 + (dz + 1)] = (uint8_t)(cnt ? cnt : 1);
            }
    h->alias = alias;
    if (!h->mult) HIP_TRY(h, hipMalloc(&h->mult, 32));
    HIP_TRY(h, copy_now(h, h->mult, m, 27, hipMemcpyHostToDevice));
    return WS_OK;
}

// kernel family and pair arithmetic of a new handle (flags already set)
void configure_kernels(ws_handle *h)
{
    if (const char *v = WS_DEV_ENV("WS_VARIANT")) h->variant = strcmp(v, "simple") == 0 ? WS_VARIANT_SIMPLE : WS_VARIANT_LISTED;
    h->ieee = (h->flags & WS_FLAG_IEEE_DIVISION) != 0;
}

ws_status ensure_stage(ws_handle *h, size_t bytes)
{
    if (h->stage_bytes >= bytes) return WS_OK;
    hipFree(h->stage);
    h->stage = nullptr;
    h->stage_bytes = 0;
    HIP_TRY(h, hipMalloc(&h->stage, bytes));
    h->stage_bytes = bytes;
    return WS_OK;
}

// A handle whose re-grid failed after the old tables were given up (h->dead; ws_set_params, slab_regrid): its cell
// tables, sorted arrays or -- on a slab -- its particle set are missing or half rebuilt.  EVERY entry point that would
// launch over them, or enter a collective its peers may skip, refuses with the reason instead (only ws_step
// did; a frame loop reads before it steps).  What still works: ws_destroy, ws_last_error, ws_ready, the profile and
// statistics getters, and ws_read_positions_end / _view of a readback that had completed.
#define WS_DEAD_CHECK(h)                                               \
    do {                                                               \
        if ((h)->dead) return fail((h), WS_ERR_HIP, (h)->err.c_str()); \
    } while (0)

// The test-only build also contains a validation mode that runs the reference's six passes literally
// (tests/refcheck/): its handles are handed over at the top of every entry point.  Nothing of it exists in the product.
#ifdef WS_WITH_REFCHECK
#include "ws_refcheck_host.inc"
#define WS_REF_DISPATCH(h, call)       \
    do {                               \
        if ((h)->refmode) return call; \
    } while (0)
#else
#define WS_REF_DISPATCH(h, call) ((void)0)
#endif

// ---- profiling ---------------------------------------------------------------------
hipEvent_t get_event(ws_handle *h)
{
    if (!h->pool.empty()) {
        hipEvent_t e = h->pool.back();
        h->pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;  // the bracket is then skipped, the step is not
    return e;
}

struct Prof {
    ws_handle *h;
    bool on;
    WsEventPair p{};
    Prof(ws_handle *h_, uint32_t k) : h(h_), on((h_->flags & WS_FLAG_PROFILE) != 0 && ((h_->prof_mask >> k) & 1u))
    {
        if (!on) return;
        p.kernel = k;
        p.a = get_event(h);
        p.b = get_event(h);
        if (!p.a || !p.b || hipEventRecord(p.a, h->stream) != hipSuccess) {  // out of events: time nothing, run everything
            if (p.a) h->pool.push_back(p.a);
            if (p.b) h->pool.push_back(p.b);
            on = false;
        }
    }
    ~Prof()
    {
        if (!on) return;
        if (hipEventRecord(p.b, h->stream) == hipSuccess) {
            h->pending.push_back(p);
        } else {
            h->pool.push_back(p.a);
            h->pool.push_back(p.b);
        }
    }
};

// The same for ONE kernel launch: the pair goes to the launch (hipExtLaunchKernelGGL start / stop events) instead of
// being recorded around it.  The pair is queued for reading only if a launch actually took it (events()); otherwise
// it goes back to the pool -- a pair no launch has stamped would read as the elapsed time of an earlier use.
struct ProfLaunch {
    ws_handle *h;
    bool on, taken = false;
    WsEventPair p{};
    ProfLaunch(ws_handle *h_, uint32_t k, bool counts = true)
        : h(h_), on((h_->flags & WS_FLAG_PROFILE) != 0 && ((h_->prof_mask >> k) & 1u))
    {
        if (!on) return;
        p.kernel = k;
        p.counts = counts;
        p.a = get_event(h);
        p.b = get_event(h);
        if (!p.a || !p.b) {
            if (p.a) h->pool.push_back(p.a);
            if (p.b) h->pool.push_back(p.b);
            on = false;
        }
    }
    // called by the launcher at the moment it launches
    const WsEventPair *events()
    {
        if (!on) return nullptr;
        taken = true;
        return &p;
    }
    ~ProfLaunch()
    {
        if (!on) return;
        if (taken) {
            h->pending.push_back(p);
        } else {
            h->pool.push_back(p.a);
            h->pool.push_back(p.b);
        }
    }
};

void drain_profile(ws_handle *h)
{
    for (auto &p : h->pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            h->prof_ms[p.kernel] += ms;
            h->prof_cnt[p.kernel] += p.counts ? 1 : 0;
        }
        h->pool.push_back(p.a);
        h->pool.push_back(p.b);
    }
    h->pending.clear();
}

// WS_FLAG_PROFILE with a host that never calls ws_sync: read the event pairs back before the list (and the event
// pool behind it) grows without bound.  Profile mode only; costs one stream synchronisation per 4096 pairs.
void bound_pending(ws_handle *h)
{
    if (h->pending.size() < 4096) return;
    if (hipStreamSynchronize(h->stream) == hipSuccess) drain_profile(h);
}

// The tile schedule of a single-GPU handle's K4 / K5 (WsSched, ws_internal.h): split / perm / cost per kernel in two
// sets, a stream and two events for k_schedule.  Built once here with no costs known: equal shares, tiles in their own
// order.  Only handles of WS_SCHED_MIN_PARTICLES <= n < WS_SCHED_END_PARTICLES take it (profiles/r05/sched/; the
// range re-measured on the round's final kernels: sched_range_final_kernels.log):
//   * there (C2: 262 144 particles; 2^19) a neighbour kernel is one or two rounds of workgroups -- most of the grid is
//     resident at once -- so it lasts as long as its slowest workgroup lives, and starting the expensive tiles first is
//     worth 10-19 % of K4 and 5-15 % of K5 (step -7.6 % settled at C2, -7 % at 2^19, -5 % sparse);
//   * below (the reference's own 65 536 particles: +12 %; 2^17: K4 -8 %, K5 -3 %, step +4 %) the step is launch-bound and
//     the extra launch with its two cross-stream events costs 5-8 us of a 75-110 us step;
//   * above (2^20 settled: K5 +8 %, step +3.8 %; 2^21: +6 %; C3, C4) an XCD works through many rounds, the equal static
//     shares are within 4 % of balanced, and every reordering of the tiles costs more in L2 locality than the shorter
//     drain returns (K5 +4 ... +13 %).  (2^20 in the sparse state would still gain 4 %: the settled state decides.)
#define WS_SCHED_MIN_PARTICLES (1u << 18)
#define WS_SCHED_END_PARTICLES (1u << 20)

void free_schedule(ws_handle *h)
{
    if (h->sched_stream) {
        hipStreamSynchronize(h->sched_stream);
        hipStreamDestroy(h->sched_stream);
    }
    if (h->ev_sched_in) hipEventDestroy(h->ev_sched_in);
    if (h->ev_sched_out) hipEventDestroy(h->ev_sched_out);
    hipFree(h->sched4[0].cost);  // (shared by both of K4's sets)
    for (int k = 0; k < 2; k++) {
        hipFree(h->sched4[k].split); hipFree(h->sched4[k].perm);
        hipFree(h->sched5[k].split); hipFree(h->sched5[k].perm); hipFree(h->sched5[k].cost);
        h->sched4[k] = h->sched5[k] = WsSched{};
    }
    h->sched_stream = nullptr;
    h->ev_sched_in = h->ev_sched_out = nullptr;
    h->sched_on = false;
}

ws_status alloc_schedule(ws_handle *h)
{
    if (h->variant != WS_VARIANT_LISTED || h->slab) return WS_OK;
    uint32_t min_n = WS_SCHED_MIN_PARTICLES, max_n = WS_SCHED_END_PARTICLES - 1u;
    if (const char *v = WS_DEV_ENV("WS_TILE_SCHEDULE")) {  // developer builds: 0 = never, 1 = at every size (A/B runs)
        min_n = 0;
        max_n = atoi(v) ? 0xFFFFFFFFu : 0u;
    }
    if (h->n > max_n || h->n < min_n) return WS_OK;
    if (const char *v = WS_DEV_ENV("WS_SCHED_CLASSES")) h->sched_classes = (uint32_t)atoi(v);
    if (const char *v = WS_DEV_ENV("WS_SCHED_GROUP")) h->sched_group = (uint32_t)atoi(v);
    h->sched_tiles4 = (h->n + wsk_density_tile() - 1u) / wsk_density_tile();
    h->sched_tiles5 = (h->n + wsk_force_tile(h->n) - 1u) / wsk_force_tile(h->n);
    uint32_t *cost4 = nullptr;
    HIP_TRY(h, hipMalloc(&cost4, (size_t)h->sched_tiles4 * 4));
    HIP_TRY(h, hipMemsetAsync(cost4, 0, (size_t)h->sched_tiles4 * 4, h->stream));
    for (int k = 0; k < 2; k++) {
        h->sched4[k].cost = cost4;
        HIP_TRY(h, hipMalloc(&h->sched4[k].split, 16 * 4));
        HIP_TRY(h, hipMalloc(&h->sched4[k].perm, (size_t)h->sched_tiles4 * 4));
        HIP_TRY(h, hipMalloc(&h->sched5[k].split, 16 * 4));
        HIP_TRY(h, hipMalloc(&h->sched5[k].perm, (size_t)h->sched_tiles5 * 4));
        HIP_TRY(h, hipMalloc(&h->sched5[k].cost, (size_t)h->sched_tiles5 * 4));
        HIP_TRY(h, hipMemsetAsync(h->sched5[k].cost, 0, (size_t)h->sched_tiles5 * 4, h->stream));
    }
    HIP_TRY(h, hipStreamCreateWithFlags(&h->sched_stream, hipStreamNonBlocking));
    HIP_TRY(h, hipEventCreateWithFlags(&h->ev_sched_in, hipEventDisableTiming));
    HIP_TRY(h, hipEventCreateWithFlags(&h->ev_sched_out, hipEventDisableTiming));
    for (int k = 0; k < 2; k++)  // (costs all unknown: both sets come out as the equal shares)
        wsk_schedule(h->stream, h->sched4[k], h->sched_tiles4, h->sched5[k], h->sched5[k], h->sched_tiles5, h->sched_classes, h->sched_group,
                     wsk_force_tile(h->n));
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    h->sched_parity = 0;
    h->sched_on = true;
    return WS_OK;
}

// The single-GPU step: scan -> scatter -> reorder -> K4 -> K5+K6+K1'.  Enqueued directly or captured.
void enqueue_step(ws_handle *h)
{
    const WsDev &d = h->dev;
    hipStream_t s = h->stream;
    // (not in a captured step: k_schedule joins the NEXT step, and the two sets alternate -- a replayed graph has one baked in)
    const bool sched = h->sched_on && !h->alias && h->variant == WS_VARIANT_LISTED && !(h->flags & WS_FLAG_GRAPH);
    const uint32_t par = h->sched_parity;
    {
        Prof p(h, WS_K_SCAN);
        // no fill cursors: particles are placed by the ranks they drew when they were binned
        wsk_scan(s, h->count, h->start + d.guard, nullptr, h->bsum, d.ncells, true, 0);
    }
    {
        Prof p(h, WS_K_SCATTER);
        wsk_place(s, d, h->cid_cur, h->cur.rank, h->start, h->slot_tmp);
    }
    {
        Prof p(h, WS_K_REORDER);
        wsk_reorder(s, d, h->slot_tmp, h->cid_cur, h->start, h->cur, h->srt, h->cid_srt, h->sxyz, h->pred_stale);
    }
    if (sched) hipStreamWaitEvent(s, h->ev_sched_out, 0);  // (the previous step's k_schedule: long done; nothing before the first)
    {
        // one launch each: timed by events the launch itself carries (no event packets between the kernels)
        ProfLaunch p(h, WS_K_DENSITY);
        wsk_density(s, d, h->start, h->cid_srt, h->srt, h->mult, h->alias, h->variant, h->ieee, h->stats, h->mask, h->sxyz,
                    p.events(), sched ? h->sched4[par] : WsSched{});
    }
    if (sched) {
        // next step's schedule, beside this step's K5 (16 workgroups; K5 is not bound by what they need): from the costs
        // K4 has just measured and the ones the PREVIOUS step's K5 left in the other set
        hipEventRecord(h->ev_sched_in, s);
        hipStreamWaitEvent(h->sched_stream, h->ev_sched_in, 0);
        wsk_schedule(h->sched_stream, h->sched4[par ^ 1u], h->sched_tiles4, h->sched5[par ^ 1u], h->sched5[par ^ 1u], h->sched_tiles5,
                     h->sched_classes, h->sched_group, wsk_force_tile(h->n));
        hipEventRecord(h->ev_sched_out, h->sched_stream);
        h->sched_parity = par ^ 1u;
    }
    {
        ProfLaunch p(h, WS_K_FORCE);
        wsk_force(s, d, h->start, h->cid_srt, h->srt, h->cur, h->accel, h->cid_cur, h->count, h->mult, h->alias,
                  h->variant, h->ieee, h->mask, false, p.events(), sched ? h->sched5[par] : WsSched{});
    }
    h->pred_stale = true;   // the epilogue stores position and velocity only (k_reorder)
    h->accel_stale = true;  // ... and no accelerations (refresh_accel)
}

// cur.pred is not maintained by the step loop (k_reorder recomputes it): bring it up to date for a reader off the loop
void refresh_pred(ws_handle *h, const WsDev &d)
{
    if (!h->pred_stale) return;
    wsk_refresh_pred(h->stream, d, h->cur);
    h->pred_stale = false;
}

// The accelerations of the last step, for the 80-byte record views: the force kernel once more over the sorted state
// that step left behind (cell table, sorted records with densities, accept masks: all intact until the next step
// starts), storing nothing but accel[].  Not profiled: it is not part of a step.
ws_status refresh_accel(ws_handle *h)
{
    if (!h->accel_stale) return WS_OK;
    WsDev d = h->dev;
    auto pass = [&](const WsDev &dd) {
        if (dd.n)
            wsk_force(h->stream, dd, h->start, h->cid_srt, h->srt, h->cur, h->accel, h->cid_cur, h->count, h->mult, h->alias,
                      h->variant, h->ieee, h->mask, true);
    };
    if (h->slab) {
        const WsSlab *S = h->slab;
        d.dyn = S->dyn;
        d.n = S->cap;  // upper bound; the kernel reads the owned count from the device
        if (S->last_split) {
            // the step ran K4 / K5 as an early and a late launch, and the early ones number their candidates on runs cut
            // to the owned range (ws_kernels.hip ws_cut): walk the same masks the same way
            d.range_sel = WS_RANGE_EARLY;
            pass(d);
            d.range_sel = d.has_left && d.has_right ? WS_RANGE_LATE_BOTH : d.has_left ? WS_RANGE_LATE_LEFT : WS_RANGE_LATE_RIGHT;
            d.n = (uint32_t)std::min<uint64_t>(S->cap, 2ull * S->halo_cap);
            pass(d);
        } else {
            d.range_sel = WS_RANGE_ALL;
            pass(d);
        }
    } else {
        pass(d);
    }
    HIP_TRY(h, hipGetLastError());
    h->accel_stale = false;
    return WS_OK;
}

// Put `cur` (freshly uploaded, or the current state after a re-grid) into the "binned" state every ws_step starts from.
ws_status bin_current(ws_handle *h)
{
    refresh_pred(h, h->dev);
    HIP_TRY(h, hipMemsetAsync(h->count, 0, (size_t)h->dev.ncells * 4, h->stream));
    {
        Prof pr(h, WS_K_BIN);
        wsk_bin(h->stream, h->dev, h->cur, h->cid_cur, h->count);
    }
    HIP_TRY(h, hipGetLastError());
    return WS_OK;
}

ws_status upload_positions(ws_handle *h, const float *pos_xyz)
{
    const size_t bytes = (size_t)h->n * 12;
    ws_status st = ensure_stage(h, bytes);
    if (st) return st;
    HIP_TRY(h, hipMemcpyAsync(h->stage, pos_xyz, bytes, hipMemcpyHostToDevice, h->stream));
    wsk_upload_positions(h->stream, (const float *)h->stage, h->cur, h->n);
    HIP_TRY(h, hipGetLastError());
    h->pred_stale = false;
    h->accel_stale = false;  // no step yet: the views report zero acceleration
    st = bin_current(h);
    if (st) return st;
    // the caller's buffer must not be referenced after return
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    h->steps = 0;
    return WS_OK;
}

// captured steps hold kernel arguments by value (WsDev) and array pointers: anything that changes either drops them
void drop_graphs(ws_handle *h)
{
    if (h->graph_exec) hipGraphExecDestroy(h->graph_exec);
    h->graph_exec = nullptr;
    if (h->slab) slab_drop_graphs(h->slab);
}

void free_particle_arrays(ws_handle *h)
{
    drop_graphs(h);
    hipFree(h->cur.pv); hipFree(h->cur.pred); hipFree(h->cur.rank);
    hipFree(h->srt.pos); hipFree(h->srt.pv);
    hipFree(h->sxyz.x); hipFree(h->sxyz.y); hipFree(h->sxyz.z);
    hipFree(h->cid_cur); hipFree(h->cid_srt); hipFree(h->accel);
    hipFree(h->slot_tmp); hipFree(h->mask.words);
    h->cur = WsSoA{};
    h->srt = WsSorted{};
    h->sxyz = WsXYZ{};
    h->cid_cur = h->cid_srt = h->slot_tmp = nullptr;
    h->accel = nullptr;
    h->mask = WsMask{nullptr, 0};
}

void free_all(ws_handle *h)
{
    if (h->stream) hipStreamSynchronize(h->stream);
    if (h->copy_stream) {
        hipStreamSynchronize(h->copy_stream);
        hipStreamDestroy(h->copy_stream);
    }
    free_schedule(h);
    if (h->rb_gathered) hipEventDestroy(h->rb_gathered);
    if (h->rb_done) hipEventDestroy(h->rb_done);
    hipFree(h->rb_stage);
    for (float *b : h->rb_host)
        if (b) hipHostFree(b);
    drain_profile(h);
    for (auto e : h->pool) hipEventDestroy(e);
    free_grid(h);
    free_particle_arrays(h);
    hipFree(h->stats); hipFree(h->mult); hipFree(h->stage);
    hipFree(h->v_keys); hipFree(h->v_perm); hipFree(h->v_tmp); hipFree(h->v_count);
    hipFree(h->v_cursor); hipFree(h->v_start); hipFree(h->v_bsum); hipFree(h->v_off);
#ifdef WS_WITH_REFCHECK
    ref_free(h);
#endif
    slab_free(h);
    if (h->done) hipEventDestroy(h->done);
    if (h->stream && h->own_stream) hipStreamDestroy(h->stream);
}

}  // namespace

// ======================================================================================
// host-side functions of the path
// ======================================================================================
extern "C" {

uint32_t ws_abi_version(void) { return WS_ABI_VERSION; }

const char *ws_status_string(ws_status s)
{
    switch (s) {
        case WS_OK: return "ok";
        case WS_ERR_INVALID_ARG: return "invalid argument";
        case WS_ERR_NO_DEVICE: return "no usable gfx950 device";
        case WS_ERR_OUT_OF_MEMORY: return "out of device memory";
        case WS_ERR_HIP: return "HIP runtime error";
        case WS_ERR_COMM: return "RCCL communication error";
        case WS_ERR_UNSUPPORTED: return "unsupported";
        case WS_ERR_NOT_READY: return "not ready";
    }
    return "unknown status";
}

const char *ws_kernel_name(uint32_t k)
{
    static const char *names[WS_K_COUNT] = {"cell_scan", "cell_scatter", "reorder", "density", "force_integrate_bin",
                                            "bin"};
    return k < WS_K_COUNT ? names[k] : "?";
}

// src/fluid_compute.rs:20-27,:67-79; src/gravity.rs:6,:29-33; src/fluid_container.rs:8-9
ws_status ws_default_params(ws_params *out)
{
    if (!out) return WS_ERR_INVALID_ARG;
    memset(out, 0, sizeof *out);
    out->delta_time = 1.0f / 60.0f;
    out->collision_damping = 0.95f;
    out->smoothing_radius = 0.25f;
    out->target_density = 10.0f;
    out->pressure_scalar = 22.0f;
    out->near_pressure_scalar = 2.0f;
    out->viscosity_strength = 0.1f;
    out->gravity[1] = -9.8f;
    const float position[3] = {0.f, 0.f, 0.f}, size[3] = {16.f, 9.f, 9.f};
    return ws_get_ext(position, size, 0.1f, out->ext_min, out->ext_max);
}

// src/fluid_compute.rs:55-63
ws_status ws_get_smoothing_kernel(const ws_params *p, ws_smoothing_kernel *out)
{
    if (!p || !out) return WS_ERR_INVALID_ARG;
    const float PI = 3.14159265358979323846f;  // std::f32::consts::PI
    const float h = p->smoothing_radius;
    out->pow2 = 15.0f / (2.0f * PI * powi_f32(h, 5));
    out->pow2_der = 15.0f / (PI * powi_f32(h, 5));
    out->pow3 = 15.0f / (PI * powi_f32(h, 6));
    out->pow3_der = 45.0f / (PI * powi_f32(h, 6));
    out->spikey_pow3 = 315.0f / (64.0f * PI * powi_f32(h, 9));
    return WS_OK;
}

// src/helpers.rs:3-20
ws_status ws_cube_fluid(uint32_t ni, uint32_t nj, uint32_t nk, float r, float *out)
{
    if (!out) return WS_ERR_INVALID_ARG;
    const float ox = r - (float)ni * r, oy = r - (float)nj * r, oz = r - (float)nk * r;
    const float diam = r * 2.0f;
    size_t o = 0;
    for (uint32_t i = 0; i < ni; i++)
        for (uint32_t j = 0; j < nj; j++)
            for (uint32_t k = 0; k < nk; k++) {
                out[o++] = (float)i * diam + ox;
                out[o++] = (float)j * diam + oy;
                out[o++] = (float)k * diam + oz;
            }
    return WS_OK;
}

// src/fluid_container.rs:42-50
ws_status ws_get_ext(const float position[3], const float size[3], float padding, float ext_min[4], float ext_max[4])
{
    if (!position || !size || !ext_min || !ext_max) return WS_ERR_INVALID_ARG;
    for (int c = 0; c < 3; c++) {
        const float half = size[c] / 2.0f;
        ext_min[c] = (position[c] - half) + padding;
        ext_max[c] = (position[c] + half) - padding;
    }
    ext_min[3] = ext_max[3] = 0.0f;
    return WS_OK;
}

// src/fluid_compute.rs:251-273
uint32_t ws_bit_sorter_stage_count(uint32_t data_length)
{
    uint32_t k = 0;
    uint64_t p = 1;
    while (p < data_length) {
        p <<= 1;
        k++;
    }
    return k * (k + 1) / 2;
}

// ======================================================================================
// lifetime
// ======================================================================================
ws_status ws_create(const ws_params *params, const float *pos_xyz, uint32_t n, const ws_device_cfg *cfg,
                    ws_handle **out)
{
    if (!out) return WS_ERR_INVALID_ARG;
    *out = nullptr;
    if (!pos_xyz || n == 0) return fail(nullptr, WS_ERR_INVALID_ARG, "pos_xyz is NULL or n == 0");
    if (n > WS_MAX_SLOTS - 16u) return fail(nullptr, WS_ERR_INVALID_ARG, "more than 2^27 particles on one device");
    ws_status st = validate_params(nullptr, params);
    if (st) return st;
    if (cfg && cfg->world_size > 1) return fail(nullptr, WS_ERR_UNSUPPORTED, "multi-GPU slabs are created with ws_slab_create");
#ifndef WS_WITH_REFCHECK
    if (cfg && (cfg->flags & WS_FLAG_REFERENCE_ORDER))
        return fail(nullptr, WS_ERR_UNSUPPORTED,
                    "WS_FLAG_REFERENCE_ORDER is a validation mode of the test-only build (tests/libwsfluid_refcheck.so)");
#endif

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(nullptr, WS_ERR_NO_DEVICE, "no HIP device visible (this library has no CPU fallback)");
    const int device = cfg ? cfg->device : 0;
    if (device < 0 || device >= ndev) return fail(nullptr, WS_ERR_NO_DEVICE, "device index out of range");
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return fail(nullptr, WS_ERR_NO_DEVICE, "hipGetDeviceProperties failed");
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, WS_ERR_NO_DEVICE, "device is not gfx950 (kernels are built for MI355X only)");

    ws_handle *h = new (std::nothrow) ws_handle();
    if (!h) return fail(nullptr, WS_ERR_OUT_OF_MEMORY, "host allocation failed");
    h->device = device;
    h->flags = cfg ? cfg->flags : 0;
    h->n = n;
    h->params = *params;
    configure_kernels(h);
    if (h->flags & WS_FLAG_GRAPH) h->flags &= ~(uint32_t)WS_FLAG_PROFILE;  // event brackets are not part of a captured step

    auto bail = [&](ws_status s) {
        g_create_error = h->err;
        free_all(h);
        delete h;
        return s;
    };
#define CREATE_TRY(expr)                  \
    do {                                  \
        ws_status s_ = (expr);            \
        if (s_) return bail(s_);          \
    } while (0)
#define CREATE_HIP(expr)                                                                   \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess)                                                              \
            return bail(fail(h, e_ == hipErrorOutOfMemory ? WS_ERR_OUT_OF_MEMORY : WS_ERR_HIP, #expr, e_)); \
    } while (0)

    CREATE_HIP(hipSetDevice(device));
    CREATE_HIP(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    CREATE_HIP(hipEventCreateWithFlags(&h->done, hipEventDisableTiming));
    CREATE_TRY(derive_dev(h, *params, n, &h->dev));
#ifdef WS_WITH_REFCHECK
    if (h->flags & WS_FLAG_REFERENCE_ORDER) {
        CREATE_TRY(ref_alloc(h));
        CREATE_TRY(ref_upload_positions(h, pos_xyz));
        *out = h;
        return WS_OK;
    }
#endif
    // +16 entries: phase 1 trips read up to U-1 slots past a run's end (masked, but must be mapped)
    const size_t n16 = ((size_t)n + 16) * 16;
    CREATE_HIP(hipMalloc(&h->cur.pv, 2 * n16));
    CREATE_HIP(hipMalloc(&h->cur.pred, n16));
    CREATE_HIP(hipMalloc(&h->cur.rank, (size_t)n * 4));
    CREATE_HIP(hipMalloc(&h->srt.pos, n16));
    CREATE_HIP(hipMalloc(&h->srt.pv, 2 * n16));
    CREATE_HIP(hipMalloc(&h->sxyz.x, n16 / 4)); CREATE_HIP(hipMalloc(&h->sxyz.y, n16 / 4)); CREATE_HIP(hipMalloc(&h->sxyz.z, n16 / 4));
    CREATE_HIP(hipMemsetAsync(h->sxyz.x, 0, n16 / 4, h->stream)); CREATE_HIP(hipMemsetAsync(h->sxyz.y, 0, n16 / 4, h->stream)); CREATE_HIP(hipMemsetAsync(h->sxyz.z, 0, n16 / 4, h->stream));
    CREATE_HIP(hipMalloc(&h->cid_cur, (size_t)n * 4));
    CREATE_HIP(hipMalloc(&h->cid_srt, (size_t)n * 4));
    CREATE_HIP(hipMalloc(&h->accel, n16));
    CREATE_HIP(hipMalloc(&h->slot_tmp, (size_t)n * 4));
    if (h->variant == WS_VARIANT_LISTED) {
        h->mask.stride = n;
        CREATE_HIP(hipMalloc(&h->mask.words, (size_t)wsk_mask_words() * n * 4));
    }
    CREATE_HIP(hipMalloc(&h->stats, 64));
    CREATE_HIP(hipMemsetAsync(h->stats, 0, 64, h->stream));
    CREATE_TRY(alloc_grid(h));
    CREATE_TRY(upload_mult(h));
    CREATE_TRY(upload_positions(h, pos_xyz));
    CREATE_TRY(alloc_schedule(h));
#undef CREATE_TRY
#undef CREATE_HIP
    *out = h;
    return WS_OK;
}

ws_status ws_destroy(ws_handle *h)
{
    if (!h) return WS_OK;
    hipSetDevice(h->device);
    free_all(h);
    delete h;
    return WS_OK;
}

const char *ws_last_error(ws_handle *h) { return h ? h->err.c_str() : g_create_error.c_str(); }
uint32_t ws_num_particles(ws_handle *h) { return h ? h->n : 0; }
uint64_t ws_steps_done(ws_handle *h) { return h ? h->steps : 0; }

// ======================================================================================
// per frame
// ======================================================================================
ws_status ws_step(ws_handle *h)
{
    if (!h) return WS_ERR_INVALID_ARG;
    HIP_TRY(h, hipSetDevice(h->device));
    if (h->slab) return slab_step(h);
    WS_REF_DISPATCH(h, ref_step(h));
    if (h->dead) return fail(h, WS_ERR_HIP, h->err.c_str());
    hipStream_t s = h->stream;
    bound_pending(h);
    // WS_FLAG_GRAPH: the five launches as one captured graph (the reference replays its pass graph the same way,
    // src/fluid_compute.rs:309-363,:396).  Off by default: measured no faster than the direct launches, which already
    // pipeline on the stream (round 1: C1 0.061 vs 0.055 ms/step, C2 0.107 vs 0.100, C3 equal; round 3: DESIGN.md).
    // The first step after an upload takes the caller's predicted positions (k_reorder<false>) and is launched directly.
    if ((h->flags & WS_FLAG_GRAPH) && !h->graph_failed && h->pred_stale) {
        if (!h->graph_exec) {
            hipGraph_t gr = nullptr;
            // (relaxed mode: a host may drive several handles from several threads -- in-process slabs, or simply two fluids --
            // and in the other modes one thread's capture makes ANOTHER thread's legacy-stream hipMemcpy fail with
            // hipErrorStreamCaptureImplicit; this thread itself issues nothing but launches, events and async copies here)
            hipError_t e = hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed);
            if (e == hipSuccess) {
                enqueue_step(h);
                e = hipStreamEndCapture(s, &gr);
                if (e == hipSuccess) e = hipGraphInstantiate(&h->graph_exec, gr, nullptr, nullptr, 0);
                if (gr) hipGraphDestroy(gr);
            }
            if (e != hipSuccess || !h->graph_exec) {
                (void)hipGetLastError();
                h->graph_failed = true;
                h->graph_exec = nullptr;
                h->err = std::string("WS_FLAG_GRAPH: capture failed (") + hipGetErrorName(e) + "), using direct launches";
            }
        }
        if (h->graph_exec) {
            HIP_TRY(h, hipGraphLaunch(h->graph_exec, s));
            h->pred_stale = true;
            h->accel_stale = true;
            h->graph_steps++;
            HIP_TRY(h, hipEventRecord(h->done, s));
            h->done_recorded = true;
            h->steps++;
            return WS_OK;
        }
    }
    enqueue_step(h);
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipEventRecord(h->done, s));
    h->done_recorded = true;
    h->steps++;
    return WS_OK;
}

ws_status ws_ready(ws_handle *h, int *ready)
{
    if (!h || !ready) return WS_ERR_INVALID_ARG;
    if (!h->done_recorded) {
        *ready = 1;
        return WS_OK;
    }
    const hipError_t e = hipEventQuery(h->done);
    if (e == hipSuccess)
        *ready = 1;
    else if (e == hipErrorNotReady)
        *ready = 0;
    else
        return fail(h, WS_ERR_HIP, "hipEventQuery", e);
    return WS_OK;
}

ws_status ws_sync(ws_handle *h)
{
    if (!h) return WS_ERR_INVALID_ARG;
    WS_DEAD_CHECK(h);
    HIP_TRY(h, hipSetDevice(h->device));
    if (h->slab) {  // also the communication stream, and the device-side error bits of every rank seen so far
        const ws_status st = slab_settle(h);
        drain_profile(h);
        return st;
    }
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    drain_profile(h);
    return WS_OK;
}

ws_status ws_set_params(ws_handle *h, const ws_params *params)
{
    if (!h) return WS_ERR_INVALID_ARG;
    WS_DEAD_CHECK(h);
    ws_status st = validate_params(h, params);
    if (st) return st;
    HIP_TRY(h, hipSetDevice(h->device));
    WsDev nd;
    st = derive_dev(h, *params, h->slab ? h->slab->n_global : h->n, &nd);  // (the cell budget follows the GLOBAL count)
    if (st) return st;
    const WsDev &od = h->dev;
    bool regrid = memcmp(nd.org, od.org, sizeof nd.org) || memcmp(nd.dim, od.dim, sizeof nd.dim) || nd.h != od.h;
    if (h->slab) {
        // keep the slab's local grid; only the global y/z extents and the radius can be compared
        regrid = nd.h != od.h || nd.dim[1] != od.dim[1] || nd.dim[2] != od.dim[2] || nd.dim[0] != od.gdim_x ||
                 memcmp(nd.org, od.org, sizeof nd.org);
        nd.dim[0] = od.dim[0]; nd.gdim_x = od.gdim_x; nd.xoff = od.xoff; nd.ncells = od.ncells; nd.guard = od.guard;
        nd.base = od.base; nd.n = od.n; nd.hash_n = od.hash_n;
        nd.has_left = od.has_left; nd.has_right = od.has_right; nd.mig = od.mig;
        memcpy(nd.lidx, od.lidx, sizeof nd.lidx);
    }
    WS_REF_DISPATCH(h, ref_set_params(h, params, nd));
    // (slab handles: COLLECTIVE when the cell size or the container changes -- every rank must make the same call)
    if (h->slab && regrid) return slab_regrid(h, params, false);
    // The last step's accelerations are computed on demand from the state and the parameters that step used: if a
    // parameter they depend on changes (or the grid goes away), compute them now.  (A host that pushes unchanged
    // parameters every frame, as the reference's update() does, pays nothing.)
    const bool accel_params_changed =
        nd.h != od.h || nd.target_density != od.target_density || nd.pressure_scalar != od.pressure_scalar ||
        nd.near_pressure_scalar != od.near_pressure_scalar || nd.viscosity != od.viscosity || nd.k_pow2_der != od.k_pow2_der ||
        nd.k_pow3_der != od.k_pow3_der || nd.k_spikey != od.k_spikey || nd.d2_accept != od.d2_accept;
    if (regrid || accel_params_changed) {
        st = refresh_accel(h);
        if (st) return st;
    }
    h->params = *params;
    if (!regrid) {
        // (update() pushes the same parameters every frame, src/fluid_compute.rs:479-481: a captured step survives that)
        if (memcmp(&h->dev, &nd, sizeof nd) != 0) {
            drop_graphs(h);
            // new dynamics (a gravity flip can set the whole fluid in motion within a few steps): a slab's messages
            // travel at full capacity until the counts of the new regime have come round (every rank makes this call)
            if (h->slab) h->slab->limit_hold = 8;
        }
        h->dev = nd;  // by-value kernel argument: picked up by the next ws_step
        return WS_OK;
    }
    drop_graphs(h);
    // cell size or container changed: rebuild the grid tables and re-bin the current state
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    h->dev = nd;
    st = alloc_grid(h);
    if (!st) st = bin_current(h);
    if (st) {
        // the old cell tables are gone and the new ones are not there: no ws_step may launch on this handle again
        h->dead = true;
        h->err = "re-grid failed after the old cell tables were given up (" + h->err + "); the handle is unusable";
    }
    return st;
}

ws_status ws_read_positions(ws_handle *h, float *out_xyz)
{
    if (!h || (!out_xyz && !h->slab)) return WS_ERR_INVALID_ARG;
    WS_DEAD_CHECK(h);
    HIP_TRY(h, hipSetDevice(h->device));
    if (h->slab) return slab_read_by_id(h, WS_PACK_POS_H, out_xyz);  // collective: all n_global positions, by id
    WS_REF_DISPATCH(h, ref_read_positions(h, out_xyz));
    const size_t bytes = (size_t)h->n * 12;
    ws_status st = ensure_stage(h, bytes);
    if (st) return st;
    wsk_gather_positions(h->stream, h->cur, (float *)h->stage, h->n);
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipMemcpyAsync(out_xyz, h->stage, bytes, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    drain_profile(h);
    return WS_OK;
}

// The frame loop's readback, overlapped with the next step: `begin` enqueues the id-order gather behind the steps
// enqueued so far (a ~65 us kernel on the library's stream) and starts the device->host copy on a separate copy
// stream; steps enqueued after `begin` run while the copy is in flight; `end` waits for the copy alone.
// out_xyz == NULL: into one of two page-locked buffers the library owns, alternately (ws_read_positions_view).
ws_status ws_read_positions_begin(ws_handle *h, float *out_xyz)
{
    if (!h) return WS_ERR_INVALID_ARG;
    WS_DEAD_CHECK(h);
    WS_REF_DISPATCH(h, fail(h, WS_ERR_UNSUPPORTED, "no asynchronous readback in the reference-order validation mode"));
    if (h->rb_inflight) return fail(h, WS_ERR_INVALID_ARG, "a readback is already in flight (call ws_read_positions_end)");
    HIP_TRY(h, hipSetDevice(h->device));
    const size_t bytes = (size_t)(h->slab ? h->slab->n_global : h->n) * 12;
    if (!h->copy_stream) {
        // A stream of another PRIORITY than the step's: the runtime multiplexes the streams of one priority onto a few
        // hardware queues (round-robin, 4 by default), and a copy stream that lands on the step stream's queue puts its
        // wait-for-the-copy barrier in front of the next step's kernels -- copy and step then run in series.  Queues
        // are pooled per priority, so this stream can never share one with the step.
        int lo = 0, hi = 0;
        const char *pr = WS_DEV_ENV("WS_COPY_STREAM_PRIORITY");  // "default" = same priority as the step's stream (A/B runs)
        if (pr && strcmp(pr, "default") == 0) {
            HIP_TRY(h, hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking));
        } else {
            HIP_TRY(h, hipDeviceGetStreamPriorityRange(&lo, &hi));  // (numerically lowest = highest priority)
            HIP_TRY(h, hipStreamCreateWithPriority(&h->copy_stream, hipStreamNonBlocking, hi));
        }
        HIP_TRY(h, hipEventCreateWithFlags(&h->rb_gathered, hipEventDisableTiming));
        HIP_TRY(h, hipEventCreateWithFlags(&h->rb_done, hipEventDisableTiming));
    }
    const bool owned = out_xyz == nullptr;
    if (owned) {
        if (h->rb_host_bytes < bytes) {
            for (float *&b : h->rb_host) {
                if (b) hipHostFree(b);
                b = nullptr;
            }
            h->rb_host_bytes = 0;
            h->rb_last = -1;
            for (float *&b : h->rb_host) HIP_TRY(h, hipHostMalloc((void **)&b, bytes, hipHostMallocDefault));
            h->rb_host_bytes = bytes;
        }
        h->rb_fill = h->rb_last == 0 ? 1 : 0;  // never the buffer the host may still be reading
        out_xyz = h->rb_host[h->rb_fill];
    }
    const float *src;
    if (h->slab) {
        // the collective gather (it settles the enqueued steps: the counts travel through the host), then the copy of
        // the id-ordered result overlaps with the steps enqueued after this call, as on a single handle
        const ws_status st = slab_gather_by_id(h, WS_PACK_POS_H);
        if (st) return st;
        src = reinterpret_cast<const float *>(h->slab->g_out);
    } else {
        if (h->rb_bytes < bytes) {
            hipFree(h->rb_stage);
            h->rb_stage = nullptr;
            h->rb_bytes = 0;
            HIP_TRY(h, hipMalloc(&h->rb_stage, bytes));
            h->rb_bytes = bytes;
        }
        wsk_gather_positions(h->stream, h->cur, h->rb_stage, h->n);
        HIP_TRY(h, hipGetLastError());
        src = h->rb_stage;
    }
    HIP_TRY(h, hipEventRecord(h->rb_gathered, h->stream));
    HIP_TRY(h, hipStreamWaitEvent(h->copy_stream, h->rb_gathered, 0));
    // hipMemcpyAsync, i.e. an SDMA engine: it reads HBM and writes PCIe beside the shader's memory path.  (Measured in
    // round 4, profiles/r04/readback: a copy KERNEL storing into the mapped host buffer -- even one of only 64 workgroups
    // -- slows the step's short bandwidth-bound kernels while it runs, k_scan 21 -> 170-560 us, k_place 31 -> 220-590 us;
    // the runtime's own blit kernel, which it substitutes for SDMA whenever a profiler is attached, does the same.)
    HIP_TRY(h, hipMemcpyAsync(out_xyz, src, bytes, hipMemcpyDeviceToHost, h->copy_stream));
    HIP_TRY(h, hipEventRecord(h->rb_done, h->copy_stream));
    h->rb_inflight = true;
    h->rb_owned = owned;
    return WS_OK;
}

ws_status ws_read_positions_end(ws_handle *h)
{
    if (!h) return WS_ERR_INVALID_ARG;
    if (!h->rb_inflight) return fail(h, WS_ERR_INVALID_ARG, "no readback in flight");
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipEventSynchronize(h->rb_done));
    h->rb_inflight = false;
    if (h->rb_owned) h->rb_last = h->rb_fill;
    return WS_OK;
}

ws_status ws_read_positions_view(ws_handle *h, const float **out_xyz)
{
    if (!h || !out_xyz) return WS_ERR_INVALID_ARG;
    *out_xyz = nullptr;
    if (h->rb_last < 0) return fail(h, WS_ERR_INVALID_ARG, "no finished readback into the library's buffers (ws_read_positions_begin(h, NULL) / _end)");
    *out_xyz = h->rb_host[h->rb_last];
    return WS_OK;
}

ws_status ws_read_speeds(ws_handle *h, float *out_speed)
{
    if (!h || (!out_speed && !h->slab)) return WS_ERR_INVALID_ARG;
    WS_DEAD_CHECK(h);
    HIP_TRY(h, hipSetDevice(h->device));
    if (h->slab) return slab_read_by_id(h, WS_PACK_SPEED_H, out_speed);
    WS_REF_DISPATCH(h, ref_read_speeds(h, out_speed));
    const size_t bytes = (size_t)h->n * 4;
    ws_status st = ensure_stage(h, bytes);
    if (st) return st;
    wsk_gather_speeds(h->stream, h->cur, (float *)h->stage, h->n);
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipMemcpyAsync(out_speed, h->stage, bytes, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    drain_profile(h);
    return WS_OK;
}

// Page-lock / release a host buffer the caller owns and keeps alive (e.g. the Vec update() reads positions
// into every frame): a device->host copy into pinned memory runs at PCIe rate (C3: 0.97 ms for 50 MB) instead
// of through the runtime's pageable staging (9.8 ms).  Explicit on purpose: pinning caller memory behind its
// back is unsafe -- the caller may free it while it is still registered.
ws_status ws_pin_host_buffer(ws_handle *h, void *ptr, uint64_t bytes)
{
    if (!h || !ptr || !bytes) return WS_ERR_INVALID_ARG;
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipHostRegister(ptr, (size_t)bytes, hipHostRegisterDefault));
    return WS_OK;
}

ws_status ws_unpin_host_buffer(ws_handle *h, void *ptr)
{
    if (!h || !ptr) return WS_ERR_INVALID_ARG;
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipHostUnregister(ptr));
    return WS_OK;
}

ws_status ws_read_particles(ws_handle *h, ws_particle80 *out)
{
    if (!h || (!out && !h->slab)) return WS_ERR_INVALID_ARG;
    WS_DEAD_CHECK(h);
    HIP_TRY(h, hipSetDevice(h->device));
    if (h->slab) return slab_read_by_id(h, WS_PACK_RECORD_H, out);  // collective: all n_global records, by id
    WS_REF_DISPATCH(h, ref_read_particles(h, out));
    const size_t bytes = (size_t)h->n * sizeof(ws_particle80);
    ws_status st = ensure_stage(h, bytes);
    if (st) return st;
    refresh_pred(h, h->dev);
    st = refresh_accel(h);
    if (st) return st;
    wsk_gather_particles(h->stream, h->dev, h->cur, h->srt, h->accel, h->steps > 0, (ws_particle80 *)h->stage, h->n);
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipMemcpyAsync(out, h->stage, bytes, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    drain_profile(h);
    return WS_OK;
}

ws_status ws_reset(ws_handle *h, const float *pos_xyz)
{
    if (!h || !pos_xyz) return WS_ERR_INVALID_ARG;
    WS_DEAD_CHECK(h);
    HIP_TRY(h, hipSetDevice(h->device));
    if (h->slab) return slab_reset(h, pos_xyz);  // pos_xyz: ALL n_global positions, on every rank
    WS_REF_DISPATCH(h, ref_upload_positions(h, pos_xyz));
    return upload_positions(h, pos_xyz);
}

ws_status ws_write_particles(ws_handle *h, const ws_particle80 *in)
{
    if (!h || !in) return WS_ERR_INVALID_ARG;
    WS_DEAD_CHECK(h);
    HIP_TRY(h, hipSetDevice(h->device));
    if (h->slab) return slab_write_particles(h, in);  // in: ALL n_global records, on every rank
    WS_REF_DISPATCH(h, ref_load(h, in, false));  // write_slice("particles") leaves the index buffers alone
    const size_t bytes = (size_t)h->n * sizeof(ws_particle80);
    ws_status st = ensure_stage(h, bytes);
    if (st) return st;
    HIP_TRY(h, hipMemcpyAsync(h->stage, in, bytes, hipMemcpyHostToDevice, h->stream));
    wsk_upload_particles(h->stream, (const ws_particle80 *)h->stage, h->cur, h->n);
    HIP_TRY(h, hipGetLastError());
    h->pred_stale = false;  // the caller's predicted positions, as they are
    h->accel_stale = false;
    st = bin_current(h);
    if (st) return st;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    h->steps = 0;
    return WS_OK;
}

// ======================================================================================
// reference-layout sort view
// ======================================================================================
ws_status ws_read_sort_view(ws_handle *h, uint32_t *keys_by_id, uint32_t *perm, uint32_t *cell_offsets)
{
    if (!h) return WS_ERR_INVALID_ARG;
    WS_DEAD_CHECK(h);
    HIP_TRY(h, hipSetDevice(h->device));
    // a slab handle answers for the WHOLE domain (collective: the keys of every rank's particles are gathered by id)
    const uint32_t n = h->slab ? h->slab->n_global : h->n;
    const bool stepped = h->slab ? h->steps > h->slab->t0 : h->steps > 0;
    hipStream_t s = h->stream;
    WS_REF_DISPATCH(h, ref_read_sort_view(h, keys_by_id, perm, cell_offsets));
    if (!h->v_keys) {
        HIP_TRY(h, hipMalloc(&h->v_keys, (size_t)n * 4));
        HIP_TRY(h, hipMalloc(&h->v_perm, (size_t)n * 4));
        HIP_TRY(h, hipMalloc(&h->v_tmp, (size_t)n * 4));
        HIP_TRY(h, hipMalloc(&h->v_count, (size_t)n * 4));
        HIP_TRY(h, hipMalloc(&h->v_cursor, (size_t)n * 4));
        HIP_TRY(h, hipMalloc(&h->v_start, ((size_t)n + 1) * 4));
        HIP_TRY(h, hipMalloc(&h->v_off, (size_t)n * 4));
        HIP_TRY(h, hipMalloc(&h->v_bsum, (size_t)wsk_scan_state_words(n) * 4));
        HIP_TRY(h, hipMemsetAsync(h->v_bsum, 0, (size_t)wsk_scan_state_words(n) * 4, h->stream));
        HIP_TRY(h, copy_now(h, h->v_start + n, &n, 4, hipMemcpyHostToDevice));
    }
    if (!stepped) {
        // src/fluid_compute.rs:306-308: all three buffers start as the identity
        if (h->slab) HIP_TRY(h, hipStreamSynchronize(s));
        wsk_iota(s, h->v_keys, n);
        HIP_TRY(h, hipGetLastError());
        HIP_TRY(h, hipStreamSynchronize(s));
        for (uint32_t *dst : {keys_by_id, perm, cell_offsets})
            if (dst) HIP_TRY(h, copy_now(h, dst, h->v_keys, (size_t)n * 4, hipMemcpyDeviceToHost));
        return WS_OK;
    }
    HIP_TRY(h, hipMemsetAsync(h->v_count, 0, (size_t)n * 4, s));
    const uint32_t *keys = h->v_keys;
    if (h->slab) {
        const ws_status st = slab_gather_by_id(h, WS_PACK_KEY_H);
        if (st) return st;
        keys = h->slab->g_out;
        wsk_view_count(s, keys, h->v_count, n);
    } else {
        // the predicted positions the last step started from live in the sorted copy
        wsk_view_keys(s, h->dev, h->srt, h->v_keys, h->v_count);
    }
    wsk_scan(s, h->v_count, h->v_start, h->v_cursor, h->v_bsum, n, false, 0);
    wsk_scatter(s, keys, h->v_cursor, h->v_tmp, n, nullptr);
    wsk_view_fix(s, h->v_tmp, keys, h->v_start, h->v_perm, n);
    wsk_view_offsets(s, h->v_start, h->v_off, n);
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipStreamSynchronize(s));
    if (keys_by_id) HIP_TRY(h, copy_now(h, keys_by_id, keys, (size_t)n * 4, hipMemcpyDeviceToHost));
    if (perm) HIP_TRY(h, copy_now(h, perm, h->v_perm, (size_t)n * 4, hipMemcpyDeviceToHost));
    if (cell_offsets) HIP_TRY(h, copy_now(h, cell_offsets, h->v_off, (size_t)n * 4, hipMemcpyDeviceToHost));
    return WS_OK;
}

// ======================================================================================
// introspection
// ======================================================================================
ws_status ws_profile_read(ws_handle *h, uint32_t k, double *total_ms, uint64_t *launches)
{
    if (!h || k >= WS_K_COUNT) return WS_ERR_INVALID_ARG;
    ws_status st = ws_sync(h);
    if (st) return st;
    if (total_ms) *total_ms = h->prof_ms[k];
    if (launches) *launches = h->prof_cnt[k];
    return WS_OK;
}

ws_status ws_profile_reset(ws_handle *h)
{
    if (!h) return WS_ERR_INVALID_ARG;
    ws_status st = ws_sync(h);
    if (st) return st;
    for (int k = 0; k < WS_K_COUNT; k++) {
        h->prof_ms[k] = 0;
        h->prof_cnt[k] = 0;
    }
    return WS_OK;
}

ws_status ws_profile_select(ws_handle *h, uint32_t kernel_mask)
{
    if (!h) return WS_ERR_INVALID_ARG;
    h->prof_mask = kernel_mask;
    return WS_OK;
}

ws_status ws_read_stats(ws_handle *h, uint32_t out[16])
{
    if (!h || !out) return WS_ERR_INVALID_ARG;
    ws_status st = ws_sync(h);
    if (st) return st;
    HIP_TRY(h, copy_now(h, out, h->stats, 64, hipMemcpyDeviceToHost));
    for (int c = 0; c < 3; c++) out[1 + c] = (uint32_t)h->dev.cm[c];  // reference cells merged per grid cell (host-side)
    out[4] = (uint32_t)(h->slab ? h->slab->graph_steps : h->graph_steps);  // steps replayed from a captured graph
    out[15] = (h->sched_on && !h->alias && !(h->flags & WS_FLAG_GRAPH)) ? 1u : 0u;  // the cost-guided tile schedule drives K4 / K5
    if (h->slab) {  // how close the fixed message capacities came to overrunning, since the last load
        uint32_t dyn[WS_DYN_WORDS];
        HIP_TRY(h, copy_now(h, dyn, h->slab->dyn, sizeof dyn, hipMemcpyDeviceToHost));
        out[5] = dyn[DY_PEAK_HALO];
        out[6] = h->slab->halo_cap;
        out[7] = dyn[DY_PEAK_MIG];
        out[8] = h->slab->mig_cap;
        out[9] = dyn[DY_PEAK_FAR];
        out[10] = h->slab->far_cap;
        out[11] = h->slab->mig_limit_next;   // what the next step's messages will carry (they follow the fluid)
        out[12] = h->slab->halo_limit_next;
        out[13] = h->slab->far_limit_next;
        out[14] = h->slab->size_waits;
        if (h->slab->world == 1) {
            out[11] = out[12] = out[13] = 0;  // no peers, no messages
        } else if (h->slab->exact_messages) {  // ... what the last step's messages carried
            out[11] = h->slab->exact_now[0];
            out[12] = h->slab->exact_now[1];
            out[13] = h->slab->exact_now[2];
        }
    }
    return WS_OK;
}

ws_status ws_grid_dims(ws_handle *h, uint32_t dims[3])
{
    if (!h || !dims) return WS_ERR_INVALID_ARG;
    for (int c = 0; c < 3; c++) dims[c] = (uint32_t)h->dev.dim[c];
    return WS_OK;
}

}  // extern "C"

#include "ws_slab.inc"
