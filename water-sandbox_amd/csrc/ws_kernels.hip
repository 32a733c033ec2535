// ws_kernels.hip -- gfx950 (CDNA4) kernels of the SPH fluid step.
//
// Written for MI355X only: 64-lane wavefronts, SoA particle arrays in HBM, dense
// z-fastest cell grid so that the three z-neighbour cells of any (dx,dy) column are
// ONE contiguous particle run (9 runs per particle instead of 27 bucket walks).
// No MFMA: the step is gather/scan/sort work bounded by HBM/L2/LDS and VALU.
//
// Arithmetic contract: this file is compiled with -ffp-contract=off, so every operation below is one IEEE
// binary32 operation, never a contracted multiply-add.  Hashing, cell assignment, the radius test, the density
// terms and the integrator are written in the evaluation order of the reference WGSL.  The pair terms of the force
// come in two forms (template parameter IEEE): WS_FLAG_IEEE_DIVISION = the reference's expression tree operation
// for operation with correctly rounded sqrt / division; default = hardware v_sqrt_f32 / v_rcp_f32 (1 ULP) with
// the scalar factors of a pair collected before they meet the direction vector (force_pair).  What differs from
// the reference otherwise is only the ORDER in which neighbours are visited (dense-grid order instead of
// hashed-bucket order).  Every parity test applies ONE tolerance to both forms.
#include <stdlib.h>

#include <algorithm>
#include <type_traits>

#include "ws_internal.h"

#include <hip/hip_ext.h>  // hipExtLaunchKernelGGL: a launch with its own start / stop events


#pragma clang fp contract(off)

#define WS_BLOCK 256
// Cells per thread of the scan.  The look-back chain is ~1 us per 64 tiles, so few large tiles (C3, 5.7 M cells:
// 8 / 16 / 32 / 64 / 96 / 128 items -> 54 / 39 / 27 / 23 / 22.5 / 25 us); but a small grid wants more, smaller tiles to
// fill the chip (C2 and the reference's default 65 536 particles: 64 items cost +2.7 us of a 54-75 us step).  Two
// instantiations, chosen by the grid size.
#define WS_SCAN_ITEMS_SMALL 32
#define WS_SCAN_ITEMS_LARGE 64
#define WS_SCAN_LARGE_FROM (1u << 21)  // cells
#define WS_SCAN_TILE_MIN (WS_BLOCK * WS_SCAN_ITEMS_SMALL)

static inline uint32_t cdiv(uint32_t a, uint32_t b) { return (a + b - 1) / b; }

// Workgroup -> tile map of the reorder pass and the two neighbour kernels.  Blocks are dealt round-robin over the 8 XCDs
// (b and b + 8 share one), each with its own L2: tile = f(b) gives every XCD one CONTIGUOUS eighth of the sorted order (an
// x-slab of the domain), so a tile's neighbour records are fetched into one L2 instead of into all eight (round 2) --
// and the SAME eighth in k_reorder, K4 and K5 (round 5), walked in alternating directions: k_reorder ascending, K4
// DESCENDING (`reverse`), K5 ascending.  So every kernel starts on what the kernel before it wrote LAST on this very
// XCD -- the end of the range k_reorder has just filled is still in this XCD's L2 when K4 begins there, the accept masks
// and densities K4 wrote last are where K5 begins (C3: K4 -9 % sparse, K5 -4 % settled; steps -3.2 % / -2.5 %; C2, the
// reference's 65 536: -2.5 %; C4 settled -1.5 %; profiles/r05/ab/ab_zigzag_*.log).  Speed / traffic only: any
// placement and any order compute the same thing.
__device__ __forceinline__ uint32_t xcd_tile(uint32_t b, uint32_t nt, bool reverse = false)
{
    const uint32_t xcd = b & 7u, q = nt >> 3, r = nt & 7u;
    const uint32_t first = xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q, mine = q + (xcd < r ? 1u : 0u);
    const uint32_t j = b >> 3;
    return first + (reverse ? mine - 1u - j : j);
}

// ---------------------------------------------------------------------------------
// cell helpers
// ---------------------------------------------------------------------------------

// get_cell, assets/simulation.wgsl:121-123: vec3<i32>(floor(position / h)) with an
// IEEE divide (not a reciprocal multiply), then clamped into the padded dense grid.
// Clamping is monotone per axis, so two particles whose true cells are adjacent stay
// in adjacent-or-equal grid cells: the 27-cell search over grid cells visits a superset
// of the reference's candidates and the exact distance test decides, as it does there.
// Global x layer (grid cells along x of the whole domain) of a position: the slab cuts are expressed in these.
__device__ __forceinline__ int grid_layer_x(const WsDev &d, float x)
{
    const float fx = floorf(x / d.h) - (float)d.org[0];
    int gxg = (int)fminf(fmaxf(fx, 0.0f), (float)(d.fdim[0] - 1));  // fmaxf/fminf also squash NaN to the low border
    if (d.coarse) gxg /= d.cm[0];
    return gxg;
}

__device__ __forceinline__ uint32_t grid_cell(const WsDev &d, float x, float y, float z)
{
    const float fy = floorf(y / d.h) - (float)d.org[1];
    const float fz = floorf(z / d.h) - (float)d.org[2];
    // x: clamped in GLOBAL grid coordinates, then shifted into this handle's local (slab) grid and
    // clamped again so a particle that left the slab still bins inside the local tables
    const int gxg = grid_layer_x(d, x);
    int gy = (int)fminf(fmaxf(fy, 0.0f), (float)(d.fdim[1] - 1));
    int gz = (int)fminf(fmaxf(fz, 0.0f), (float)(d.fdim[2] - 1));
    if (d.coarse) {  // (wave-uniform; integer division only on the rare handles whose grid cells are merged)
        gy /= d.cm[1];
        gz /= d.cm[2];
    }
    const int gx = min(max(gxg - d.xoff, 0), d.dim[0] - 1);
    return (uint32_t)((gx * d.dim[1] + gy) * d.dim[2] + gz);
}

// hash_cell(get_cell(p)), assets/simulation.wgsl:121-128 (u32 wrap arithmetic).
__device__ __forceinline__ uint32_t ref_hash_key(const WsDev &d, float x, float y, float z)
{
    const uint32_t cx = (uint32_t)(int32_t)floorf(x / d.h);
    const uint32_t cy = (uint32_t)(int32_t)floorf(y / d.h);
    const uint32_t cz = (uint32_t)(int32_t)floorf(z / d.h);
    return (cx * 15823u + cy * 9737333u + cz * 440817757u) % d.hash_n;
}

// ---------------------------------------------------------------------------------
// particle counts and ranges that only the device knows (slab handles; see WsDev::dyn)
// ---------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t ws_n(const WsDev &d) { return d.dyn ? d.dyn[DY_N] : d.n; }

// The part of the sorted order a density / force launch covers, as a span of `len` consecutive VIRTUAL offsets:
// offset v is the particle lo + v, plus `jump` from offset `split` on.  Single-GPU handle: all of it, no jump.
// Slab handle: the owned range, or one of the pieces the halo / compute overlap cuts it into -- x is the slowest axis
// of the sort, so "the owned layer next to a neighbour slab" is a contiguous range delimited by cell starts: EARLY =
// the particles whose searches touch no ghost layer, LATE_LEFT / LATE_RIGHT = layers 1 / nxl-2, LATE_BOTH = the two
// of them in one launch (the left layer, then a jump over the early range to the right layer).
struct WsSpan {
    uint32_t lo, len, split, jump;
};
__device__ __forceinline__ WsSpan ws_span(const WsDev &d, const uint32_t *__restrict__ start)
{
    WsSpan sp = {d.base, d.n, 0xFFFFFFFFu, 0u};
    if (!d.dyn) return sp;
    const uint32_t n = d.dyn[DY_N], end = d.base + n;
    sp.len = n;
    if (d.range_sel == WS_RANGE_ALL) return sp;
    const uint32_t a = d.has_left ? start[d.lidx[1]] : d.base;  // first particle that needs no left ghosts
    const uint32_t b = d.has_right ? start[d.lidx[2]] : end;     // first particle of the right boundary layer
    if (d.range_sel == WS_RANGE_EARLY) {
        sp.lo = a;
        sp.len = b - a;
    } else if (d.range_sel == WS_RANGE_LATE_LEFT) {
        sp.len = a - d.base;
    } else if (d.range_sel == WS_RANGE_LATE_RIGHT) {
        sp.lo = b;
        sp.len = end - b;
    } else {  // WS_RANGE_LATE_BOTH
        sp.split = a - d.base;
        sp.jump = b - a;
        sp.len = (a - d.base) + (end - b);
    }
    return sp;
}
__device__ __forceinline__ uint32_t span_at(const WsSpan &sp, uint32_t v) { return sp.lo + v + (v >= sp.split ? sp.jump : 0u); }

// The EARLY launches of a slab step (the particles whose searches touch no ghost layer) run while the halo exchange
// on the other stream is still rewriting the ghost layers' cell starts (k_halo_unpack).  No true neighbour of an
// early particle lies in a ghost layer -- but the linearised stencil of a particle clamped into a border row of the
// grid (y row 0 or dim1 - 1, z column 0 or dim2 - 1: predicted positions that overshot the two cells of padding)
// wraps around into the previous / next layer's cells, which for the first / last early layer is a ghost layer.
// Those wrapped cells hold no neighbour (they are a whole grid extent away), but their run lengths number the
// candidates of the accept masks: K4 (before the unpack) and K5 (after it) must count the same.  So early launches
// cut every run to the owned range [base, base + n): a ghost layer's starts clamp to the same ends whatever they
// hold.  The listed kernels have an instantiation of their own for early launches (template parameter CUT), so that
// every other launch -- all of a single-GPU handle's -- compiles to exactly the code without it (a run-time switch,
// though wave-uniform, cost the dense force kernel 4 % through register allocation alone); the simple kernels test
// at run time.
struct WsCut {
    uint32_t lo, hi;
    bool on;
};
__device__ __forceinline__ WsCut ws_cut(const WsDev &d)
{
    WsCut c = {0u, 0xFFFFFFFFu, false};
    if (d.dyn && d.range_sel == WS_RANGE_EARLY) {
        c.lo = d.base;
        c.hi = d.base + d.dyn[DY_N];
        c.on = true;
    }
    return c;
}
template <bool CUT>
__device__ __forceinline__ void cut_run_t(const WsDev &d, uint32_t &b, uint32_t &e)
{
    if constexpr (CUT) {
        const uint32_t lo = d.base, hi = d.base + d.dyn[DY_N];
        b = min(max(b, lo), hi);
        e = min(max(e, lo), hi);
    }
}
__device__ __forceinline__ void cut_run(const WsCut &c, uint32_t &b, uint32_t &e)
{
    if (c.on) {
        b = min(max(b, c.lo), c.hi);
        e = min(max(e, c.lo), c.hi);
    }
}

// ---------------------------------------------------------------------------------
// uploads
// ---------------------------------------------------------------------------------

// FluidParticle::make_vec_from_positions, src/fluid_compute.rs:118-130.
__global__ void __launch_bounds__(WS_BLOCK) k_upload_positions(const float *__restrict__ xyz, WsSoA cur,
                                                               uint32_t n)
{
    const uint32_t i = blockIdx.x * WS_BLOCK + threadIdx.x;
    if (i >= n) return;
    const float x = xyz[3 * (size_t)i], y = xyz[3 * (size_t)i + 1], z = xyz[3 * (size_t)i + 2];
    cur.pos(i) = make_float4(x, y, z, __uint_as_float(i));
    cur.vel(i) = make_float4(0.f, 0.f, 0.f, 0.f);
    cur.pred[i] = make_float4(x, y, z, 0.f);
}

__global__ void __launch_bounds__(WS_BLOCK) k_upload_particles(const ws_particle80 *__restrict__ in, WsSoA cur,
                                                               uint32_t n)
{
    const uint32_t i = blockIdx.x * WS_BLOCK + threadIdx.x;
    if (i >= n) return;
    const float4 *rec = reinterpret_cast<const float4 *>(in + i);
    const float4 p = rec[0], v = rec[2], q = rec[4];
    cur.pos(i) = make_float4(p.x, p.y, p.z, __uint_as_float(i));
    cur.vel(i) = make_float4(v.x, v.y, v.z, 0.f);
    cur.pred[i] = make_float4(q.x, q.y, q.z, 0.f);
}

void wsk_upload_positions(hipStream_t s, const float *xyz_dev, WsSoA cur, uint32_t n)
{
    hipLaunchKernelGGL(k_upload_positions, dim3(cdiv(n, WS_BLOCK)), dim3(WS_BLOCK), 0, s, xyz_dev, cur, n);
}
void wsk_upload_particles(hipStream_t s, const ws_particle80 *in_dev, WsSoA cur, uint32_t n)
{
    hipLaunchKernelGGL(k_upload_particles, dim3(cdiv(n, WS_BLOCK)), dim3(WS_BLOCK), 0, s, in_dev, cur, n);
}

// ---------------------------------------------------------------------------------
// K1': cell binning (stand-alone form; the steady-state form is fused into k_force)
// ---------------------------------------------------------------------------------
#define WS_DEAD 0xFFFFFFFFu  // cell id of a slot whose particle has left the slab (slab handles)

// (the cell id is kept twice: cid[i], which k_place streams, and the w lane of the particle's velocity record, which
// k_reorder fetches with the record -- every writer of one writes the other)
__global__ void __launch_bounds__(WS_BLOCK) k_bin(WsDev d, WsSoA cur, uint32_t *__restrict__ cid, uint32_t *__restrict__ count)
{
    const uint32_t i = blockIdx.x * WS_BLOCK + threadIdx.x;
    if (i >= ws_n(d)) return;
    const float4 p = cur.pred[i];
    const uint32_t c = grid_cell(d, p.x, p.y, p.z);
    cid[i] = c;
    cur.vel(i).w = __uint_as_float(c);
    const uint32_t r = atomicAdd(&count[c], 1u);
    if (cur.rank) cur.rank[i] = r;
}

void wsk_bin(hipStream_t s, const WsDev &d, WsSoA cur, uint32_t *cid, uint32_t *count)
{
    hipLaunchKernelGGL(k_bin, dim3(cdiv(d.n, WS_BLOCK)), dim3(WS_BLOCK), 0, s, d, cur, cid, count);
}

// ---------------------------------------------------------------------------------
// exclusive scan of the per-cell counts -> cell starts (K3's result:
// calculate_cell_offsets, assets/bitonic_sort.wgsl:48-59, as a prefix sum)
// ---------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v)
{
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(v, o, 64);
        if (lane >= o) v += t;
    }
    return v;
}

// Wave-level aggregation of `atomicAdd(&table[key], 1)`: consecutive lanes with the same key form a
// run; the run's first lane adds the run length once and every member gets base + its offset in the run.
// Particles are processed in (previous) cell order, so in a dense fluid runs are ~7 lanes long and the
// atomic traffic on the per-cell counters drops accordingly.  Returns this lane's value of the counter
// (what its own atomicAdd(.., 1) would have returned in SOME serialisation).  `active` lanes only.
__device__ __forceinline__ uint32_t wave_run_atomic_inc(uint32_t *__restrict__ table, uint32_t key, bool active)
{
    // May be called from divergent code: lanes that are not executing simply do not appear in `act`
    // (their shuffled key is never trusted: a lane whose predecessor is not in `act` starts a run).
    const int lane = threadIdx.x & 63;
    const unsigned long long act = __ballot(active);
    const uint32_t prev = __shfl_up(key, 1, 64);
    const bool prev_active = lane > 0 && ((act >> (lane - 1)) & 1ull);
    const bool head = active && (!prev_active || key != prev);
    const unsigned long long heads = __ballot(head);
    if (!active) return 0u;
    const unsigned long long upto = heads & ((lane == 63) ? ~0ull : ((2ull << lane) - 1ull));
    const int hl = 63 - __builtin_clzll(upto);                      // my run's head lane
    const unsigned long long above = (hl == 63) ? 0ull : (~((2ull << hl) - 1ull));
    // the run ends at the next head or at the first inactive lane above the head
    const unsigned long long stop = (heads | ~act) & above;
    const int end = stop ? __builtin_ctzll(stop) : 64;
    uint32_t base = 0;
    if (lane == hl) base = atomicAdd(&table[key], (uint32_t)(end - hl));
    base = __shfl(base, hl, 64);
    return base + (uint32_t)(lane - hl);
}

// One-pass exclusive scan (decoupled look-back): tile t publishes its aggregate, then walks back over its
// predecessors' descriptors until it meets an inclusive prefix, and publishes its own.  Tiles take their
// index from a ticket counter, so every predecessor of a running tile is itself running or done: the
// wait cannot deadlock.  state[0..1] = a 64-bit ticket counter, never reset: every launch on a buffer draws exactly
// ntiles tickets, so ticket / ntiles numbers the launches (the `epoch`) and ticket % ntiles the tiles -- nothing of
// this comes from the host, so a captured hipGraph of the step replays it unchanged.  64-bit descriptors from
// state[2]: hi = epoch << 2 | status (1 = aggregate, 2 = inclusive prefix), lo = value.  A descriptor of
// another epoch reads as "not yet written", so nothing is cleared between launches.  A descriptor carries its
// whole message in one 64-bit word, so relaxed device-scope atomics suffice (acquire / release at agent scope
// would write back and invalidate the XCD's L2 around every access: measured 8x slower).
template <bool ZERO, int WS_SCAN_ITEMS>
__global__ void __launch_bounds__(WS_BLOCK) k_scan(uint32_t *__restrict__ count, uint32_t nitems,
                                                   uint32_t *__restrict__ state, uint32_t *__restrict__ start,
                                                   uint32_t *__restrict__ cursor, uint32_t start_offset, uint32_t ntiles)
{
    __shared__ uint32_t s_tile, s_epoch, s_prefix;
    if (threadIdx.x == 0) {
        const unsigned long long ticket = atomicAdd(reinterpret_cast<unsigned long long *>(state), 1ull);
        s_tile = (uint32_t)(ticket % ntiles);
        s_epoch = (uint32_t)(ticket / ntiles) + 1u;  // modulo 2^32: only ever compared for equality with its neighbours
    }
    __syncthreads();
    const uint32_t tile = s_tile, epoch = s_epoch;
    // A tile is WS_SCAN_ITEMS / 4 chunks of WS_BLOCK uint4: thread t owns uint4 number t of every chunk, so
    // every load and store of the tile is a coalesced 16 B per lane.
    constexpr int Q = WS_SCAN_ITEMS / 4;
    constexpr uint32_t WS_SCAN_TILE = WS_BLOCK * WS_SCAN_ITEMS;
    const uint32_t tbase = tile * WS_SCAN_TILE;
    uint4 v[Q];
    uint32_t excl[Q], total = 0;
    // all of the tile's loads first (one memory round trip, not Q of them) ...
#pragma unroll
    for (int q = 0; q < Q; q++) {
        const uint32_t at = tbase + (q * WS_BLOCK + threadIdx.x) * 4u;
        if (at + 4u <= nitems) {
            v[q] = *reinterpret_cast<const uint4 *>(count + at);
        } else {
            v[q].x = at < nitems ? count[at] : 0u;
            v[q].y = at + 1u < nitems ? count[at + 1u] : 0u;
            v[q].z = at + 2u < nitems ? count[at + 2u] : 0u;
            v[q].w = 0u;
        }
    }
    // ... then the Q chunk scans with ONE barrier: wave scans in registers, the waves' sums through LDS
    {
        __shared__ uint32_t wsum[Q][WS_BLOCK / 64];
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
        for (int q = 0; q < Q; q++) {
            excl[q] = wave_incl_scan(v[q].x + v[q].y + v[q].z + v[q].w);
            if (lane == 63) wsum[q][wave] = excl[q];
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < Q; q++) {
            uint32_t base = 0, tot = 0;
#pragma unroll
            for (int w = 0; w < WS_BLOCK / 64; w++) {
                const uint32_t sw = wsum[q][w];
                if (w < wave) base += sw;
                tot += sw;
            }
            excl[q] = total + base + excl[q] - (v[q].x + v[q].y + v[q].z + v[q].w);
            total += tot;
        }
    }
    if (threadIdx.x < 64) {
        // wave 0 looks back 64 predecessors at a time (one memory round trip per window)
        unsigned long long *desc = reinterpret_cast<unsigned long long *>(state + 2);
        const uint32_t e30 = epoch & 0x3FFFFFFFu;
        const unsigned long long tag = (unsigned long long)(e30 << 2) << 32;
        const int lane = (int)threadIdx.x;
        uint32_t prefix = 0;
        if (tile != 0u) {
            if (lane == 0)
                __hip_atomic_store(&desc[tile], tag | (1ull << 32) | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (int32_t newest = (int32_t)tile - 1;;) {
                const int32_t p = newest - lane;  // lane 0 = the nearest predecessor of this window
                // tiles "before the first" count as an inclusive prefix of 0, which ends every walk
                const unsigned long long dsc =
                    p >= 0 ? __hip_atomic_load(&desc[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (tag | (2ull << 32));
                const uint32_t hi = (uint32_t)(dsc >> 32);
                const bool ready = (hi >> 2) == e30 && (hi & 3u) != 0u;
                const unsigned long long inclusive = __ballot(ready && (hi & 3u) == 2u);
                const int stop = inclusive ? __ffsll((long long)inclusive) - 1 : 63;  // last lane whose value counts
                const unsigned long long needed = stop == 63 ? ~0ull : (2ull << stop) - 1ull;
                if (__ballot(!ready) & needed) {
                    __builtin_amdgcn_s_sleep(1);
                    continue;  // somebody this walk depends on has not published yet: same window again
                }
                uint32_t v = lane <= stop ? (uint32_t)dsc : 0u;
#pragma unroll
                for (int sh = 32; sh >= 1; sh >>= 1) v += __shfl_xor(v, sh, 64);
                prefix += v;
                if (inclusive) break;
                newest -= 64;
            }
        }
        if (lane == 0) {
            __hip_atomic_store(&desc[tile], tag | (2ull << 32) | (prefix + total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_prefix = prefix;
        }
    }
    __syncthreads();
    const uint32_t carry = s_prefix + start_offset;
#pragma unroll
    for (int q = 0; q < Q; q++) {
        const uint32_t at = tbase + (q * WS_BLOCK + threadIdx.x) * 4u;
        uint4 r;
        r.x = carry + excl[q];
        r.y = r.x + v[q].x;
        r.z = r.y + v[q].y;
        r.w = r.z + v[q].z;
        if (at + 4u <= nitems) {
            *reinterpret_cast<uint4 *>(start + at) = r;
            if (cursor) *reinterpret_cast<uint4 *>(cursor + at) = r;
            if (ZERO) *reinterpret_cast<uint4 *>(count + at) = make_uint4(0u, 0u, 0u, 0u);
        } else {
            const uint32_t rr[3] = {r.x, r.y, r.z};
            for (uint32_t k = 0; k < 3u && at + k < nitems; k++) {
                start[at + k] = rr[k];
                if (cursor) cursor[at + k] = rr[k];
                if (ZERO) count[at + k] = 0u;
            }
        }
    }
}

// words of scan state for `nitems` items (zero-initialised once by the owner)
uint32_t wsk_scan_state_words(uint32_t nitems) { return 2u + 2u * cdiv(nitems, WS_SCAN_TILE_MIN); }  // (the smaller tile: more descriptors)

// start_body points at the first real entry (after the guard); entry [nitems] and the
// guards are constant and written once by the host.
void wsk_scan(hipStream_t s, uint32_t *count, uint32_t *start_body, uint32_t *cursor, uint32_t *state, uint32_t nitems,
              bool zero_count, uint32_t base)
{
    // (the tile size is a function of the buffer's length, so every launch on a state buffer draws the same number of
    // tickets -- what the epoch arithmetic of the kernel relies on)
    const bool large = nitems >= WS_SCAN_LARGE_FROM;
    const uint32_t ntiles = cdiv(nitems, WS_BLOCK * (large ? WS_SCAN_ITEMS_LARGE : WS_SCAN_ITEMS_SMALL));
#define WS_SCAN_LAUNCH(Z, I) \
    hipLaunchKernelGGL((k_scan<Z, I>), dim3(ntiles), dim3(WS_BLOCK), 0, s, count, nitems, state, start_body, cursor, base, ntiles)
    if (zero_count) {
        if (large) WS_SCAN_LAUNCH(true, WS_SCAN_ITEMS_LARGE);
        else WS_SCAN_LAUNCH(true, WS_SCAN_ITEMS_SMALL);
    } else {
        if (large) WS_SCAN_LAUNCH(false, WS_SCAN_ITEMS_LARGE);
        else WS_SCAN_LAUNCH(false, WS_SCAN_ITEMS_SMALL);
    }
#undef WS_SCAN_LAUNCH
}

// ---------------------------------------------------------------------------------
// K2': counting sort by cell.  k_place puts every particle's INDEX into a tentative slot of its cell (cell start + the
// rank it drew when it was binned: arrival order); k_reorder then fetches the records through the indices and ranks the
// members of each cell by PARTICLE ID, so the order inside a cell is canonical: it depends on neither the previous
// order nor, across GPUs, on the order in which migrated particles arrived.  Summation order -- hence every float -- is
// therefore a function of the particle set alone, and an N-GPU run reproduces the 1-GPU run bit for bit.
// Slots are absolute indices into the sorted arrays (cell starts already include d.base).
// Round 5: the ids the ranking compares come out of the records k_reorder fetches anyway (a workgroup shares them through
// LDS: cell-mates are neighbours in the tentative order), so k_place no longer reads the 16-byte position records for
// their id lane, nor scatters a second word per particle (12 B read + 4 B scattered write per particle instead of 24 + 8).
// ---------------------------------------------------------------------------------
// Placement through per-cell fill cursors (a slab's first step after an upload, where k_migrate_mark took the leavers
// out of the histogram and the ranks drawn at binning have holes; and the reference-layout sort view, by bucket):
// arrival order inside the cell, one returning atomic per run of lanes.
__global__ void __launch_bounds__(WS_BLOCK) k_scatter(const uint32_t *__restrict__ keys, uint32_t *__restrict__ cursor,
                                                      uint32_t *__restrict__ slot_tmp, uint32_t n,
                                                      const uint32_t *__restrict__ n_dev)
{
    const uint32_t i = blockIdx.x * WS_BLOCK + threadIdx.x;
    bool active = i < (n_dev ? *n_dev : n);  // slab handles: n is an upper bound, the count lives on the device
    if (active && keys[i] == WS_DEAD) active = false;  // only after a migration overrun (the step is already flagged invalid)
    const uint32_t slot = wave_run_atomic_inc(cursor, active ? keys[i] : 0u, active);
    if (active) slot_tmp[slot] = i;
}

// The same placement from ranks taken when the particles were binned (the force kernel's epilogue, k_bin, or
// k_migrate_fill for a slab's arrivals): slot = cell start + rank, no atomics.  The ranks of a cell are gap-free
// on a slab as well: a particle that leaves is never binned (force epilogue), one that arrives draws the next rank.
// cid / rank are indexed from the first owned particle; slots are absolute.
__global__ void __launch_bounds__(WS_BLOCK) k_place(WsDev d, const uint32_t *__restrict__ cid,
                                                    const uint32_t *__restrict__ rank,
                                                    const uint32_t *__restrict__ start, uint32_t *__restrict__ slot_tmp)
{
    const uint32_t i = blockIdx.x * WS_BLOCK + threadIdx.x;
    if (i >= ws_n(d)) return;
    const uint32_t c = cid[i];
    if (c == WS_DEAD) return;  // only after a migration overrun (the step is already flagged invalid)
    slot_tmp[start[d.guard + c] + rank[i]] = i;
}

void wsk_place(hipStream_t s, const WsDev &d, const uint32_t *cid, const uint32_t *rank, const uint32_t *start, uint32_t *slot_tmp)
{
    hipLaunchKernelGGL(k_place, dim3(cdiv(d.n, WS_BLOCK)), dim3(WS_BLOCK), 0, s, d, cid, rank, start, slot_tmp);
}

void wsk_scatter(hipStream_t s, const uint32_t *keys, uint32_t *cursor, uint32_t *slot_tmp, uint32_t n, const uint32_t *n_dev)
{
    hipLaunchKernelGGL(k_scatter, dim3(cdiv(n, WS_BLOCK)), dim3(WS_BLOCK), 0, s, keys, cursor, slot_tmp, n, n_dev);
}

// `cur` and cid_cur are indexed from the first owned particle (the caller passes offset pointers);
// srt / cid_srt / slot_tmp are indexed absolutely.
// RECOMPUTE_PRED: the force kernel's epilogue does not store the predicted positions it bins by -- they are
// pos + vel * look-ahead of the values it does store, and the same two operations here give the same bits
// (simulation.wgsl:307).  After an upload (ws_create, ws_reset, ws_write_particles, a slab's first step) the stored
// `cur.pred` is used instead: the caller's predicted positions are taken as they are.
#define WS_LOOKAHEAD 0.02f  // 1. / 50., simulation.wgsl:3
typedef uint32_t nd_u4u __attribute__((ext_vector_type(4), aligned(4)));  // 4 consecutive words, 4-byte aligned
#ifndef WS_REORDER_BLOCK
#define WS_REORDER_BLOCK 512  // 256 / 512 / 1024: C3 sparse step 0.4615 / 0.4602 / 0.4611 ms, C4 settled 9.411 / 9.387 / 9.371 (a cell that straddles the edge costs two dependent gathers per mate)
#endif
template <bool RECOMPUTE_PRED>
__global__ void __launch_bounds__(WS_REORDER_BLOCK) k_reorder(WsDev d, const uint32_t *__restrict__ slot_tmp,
                                                      const uint32_t *__restrict__ cid_cur,
                                                      const uint32_t *__restrict__ start, WsSoA cur, WsSorted srt,
                                                      uint32_t *__restrict__ cid_srt, WsXYZ sxyz)
{
    __shared__ uint32_t s_id[WS_REORDER_BLOCK];  // the ids of this workgroup's slots (0xFFFFFFFF: nothing there)
    // (single-GPU handles: every XCD fills the contiguous eighth of the sorted order it will work on in K4 / K5 -- xcd_tile)
    const uint32_t blk = d.dyn ? blockIdx.x : xcd_tile(blockIdx.x, gridDim.x);
    const uint32_t k = blk * WS_REORDER_BLOCK + threadIdx.x;
    const uint32_t s = d.base + k, s0 = d.base + blk * WS_REORDER_BLOCK;
    bool active = k < ws_n(d);
    uint32_t i = 0, c = WS_DEAD;
    float4 p = make_float4(0.f, 0.f, 0.f, __uint_as_float(0xFFFFFFFFu)), v = p;
    if (active) {
        i = slot_tmp[s];
        p = cur.pos(i);  // the two halves of ONE 32-byte record: position with the id, velocity with the cell id
        v = cur.vel(i);
        c = __float_as_uint(v.w);
        v.w = 0.f;
        active = c != WS_DEAD;  // a stale slot after a migration overrun (the step is already flagged invalid)
        if (!active) p.w = __uint_as_float(0xFFFFFFFFu);
    }
    const uint32_t id = __float_as_uint(p.w);
    s_id[threadIdx.x] = id;
    __syncthreads();
    if (!active) return;
    const uint32_t b = start[d.guard + c], e = start[d.guard + c + 1];
    uint32_t rank = 0;
    for (uint32_t t = b; t < e; t++) {
        // cell-mates are neighbours in the tentative order: all but those of a cell that straddles the workgroup's edge
        // sit in LDS; for the others the id comes the way this thread's own did
        const uint32_t o = t - s0;
        const uint32_t other = o < (uint32_t)WS_REORDER_BLOCK ? s_id[o] : __float_as_uint(cur.pos(slot_tmp[t]).w);
        rank += other < id ? 1u : 0u;
    }
    const uint32_t dst = b + rank;
    float4 q;
    if constexpr (RECOMPUTE_PRED)
        q = make_float4(p.x + v.x * WS_LOOKAHEAD, p.y + v.y * WS_LOOKAHEAD, p.z + v.z * WS_LOOKAHEAD, 0.f);
    else
        q = cur.pred[i];
    srt.pos[dst] = p;
    srt.vel(dst) = v;
    srt.pred(dst) = q;
    // K4's radius tests read the predicted positions as three planar arrays: one 16-B load there
    // fetches x (or y, z) of FOUR consecutive candidates
    sxyz.x[dst] = q.x;
    sxyz.y[dst] = q.y;
    sxyz.z[dst] = q.z;
    cid_srt[dst] = c;
}

void wsk_reorder(hipStream_t s, const WsDev &d, const uint32_t *slot_tmp, const uint32_t *cid_cur, const uint32_t *start, WsSoA cur,
                 WsSorted srt, uint32_t *cid_srt, WsXYZ sxyz, bool recompute_pred)
{
    if (recompute_pred)
        hipLaunchKernelGGL(k_reorder<true>, dim3(cdiv(d.n, WS_REORDER_BLOCK)), dim3(WS_REORDER_BLOCK), 0, s, d, slot_tmp, cid_cur, start, cur, srt,
                           cid_srt, sxyz);
    else
        hipLaunchKernelGGL(k_reorder<false>, dim3(cdiv(d.n, WS_REORDER_BLOCK)), dim3(WS_REORDER_BLOCK), 0, s, d, slot_tmp, cid_cur, start, cur, srt,
                           cid_srt, sxyz);
}

// cur.pred of the owned particles from their stored position and velocity (see k_reorder): for the paths off the
// step loop that read it -- the 80-byte record views, re-binning after a parameter change
__global__ void __launch_bounds__(WS_BLOCK) k_refresh_pred(WsDev d, WsSoA cur)
{
    const uint32_t k = blockIdx.x * WS_BLOCK + threadIdx.x;
    if (k >= ws_n(d)) return;
    const uint32_t i = d.base + k;
    const float4 p = cur.pos(i), v = cur.vel(i);
    cur.pred[i] = make_float4(p.x + v.x * WS_LOOKAHEAD, p.y + v.y * WS_LOOKAHEAD, p.z + v.z * WS_LOOKAHEAD, 0.f);
}

void wsk_refresh_pred(hipStream_t s, const WsDev &d, WsSoA cur)
{
    if (d.n) hipLaunchKernelGGL(k_refresh_pred, dim3(cdiv(d.n, WS_BLOCK)), dim3(WS_BLOCK), 0, s, d, cur);
}

// ---------------------------------------------------------------------------------
// smoothing kernels, assets/simulation.wgsl:93-117 (evaluation order as written)
// ---------------------------------------------------------------------------------
__device__ __forceinline__ float sk_density(const WsDev &d, float dst)
{
    const float v = d.h - dst;
    return v * v * d.k_pow2;
}
__device__ __forceinline__ float sk_near(const WsDev &d, float dst)
{
    const float v = d.h - dst;
    return v * v * v * d.k_pow3;
}
__device__ __forceinline__ float sk_der(const WsDev &d, float dst) { return (dst - d.h) * d.k_pow2_der; }
__device__ __forceinline__ float sk_der_near(const WsDev &d, float dst)
{
    const float v = dst - d.h;
    return v * v * d.k_pow3_der;
}
__device__ __forceinline__ float sk_visc(const WsDev &d, float dst)
{
    const float v = d.h * d.h - dst * dst;
    return v * v * v * d.k_spikey;
}

// How many of the 27 stencil offsets of the reference's hashed table land on the
// bucket of a neighbour whose true cell differs by (dx,dy,dz): 1 for every N where no
// two stencil cells alias (all the benchmark sizes), more for tiny N.  For a power-of-two N
// hash_cell is linear mod N and the count depends on the cell difference only (27-entry table);
// for any other N the `% N` after the 2^32 wrap makes it depend on the cell itself, and the
// count is taken literally: the stencil offsets o of a's cell with hash(cell_a + o) == hash(cell_b).
__device__ __forceinline__ uint32_t alias_mult(const WsDev &d, const uint8_t *__restrict__ mult, float4 a, float4 b)
{
    const int ax = (int)floorf(a.x / d.h), ay = (int)floorf(a.y / d.h), az = (int)floorf(a.z / d.h);
    const int bx = (int)floorf(b.x / d.h), by = (int)floorf(b.y / d.h), bz = (int)floorf(b.z / d.h);
    const int dx = bx - ax, dy = by - ay, dz = bz - az;
    if (dx < -1 || dx > 1 || dy < -1 || dy > 1 || dz < -1 || dz > 1) return 0u;
    if ((d.hash_n & (d.hash_n - 1u)) == 0u) return mult[(dx + 1) * 9 + (dy + 1) * 3 + (dz + 1)];
    const uint32_t hb = ((uint32_t)bx * 15823u + (uint32_t)by * 9737333u + (uint32_t)bz * 440817757u) % d.hash_n;
    uint32_t cnt = 0;
    for (int ox = -1; ox <= 1; ox++)
        for (int oy = -1; oy <= 1; oy++)
            for (int oz = -1; oz <= 1; oz++) {
                const uint32_t ho = ((uint32_t)(ax + ox) * 15823u + (uint32_t)(ay + oy) * 9737333u +
                                     (uint32_t)(az + oz) * 440817757u) % d.hash_n;
                cnt += ho == hb ? 1u : 0u;
            }
    return cnt;
}

// ---------------------------------------------------------------------------------
// per-pair arithmetic of K4 / K5, shared by every kernel variant.
// ---------------------------------------------------------------------------------

// sqrt and division of the pair terms.  IEEE = true: correctly rounded, in the reference's operation order -- the
// arithmetic of a CPU restatement (and always of the reference-order validation mode).  IEEE = false (default of
// the production kernels): the hardware's v_sqrt_f32 / v_rcp_f32 (1 ULP each), x / y evaluated as x * rcp(y) --
// within the accuracy the reference's shading language itself grants its GPU (WGSL: x / y 2.5 ULP, sqrt as
// 1 / inverseSqrt at 2 ULP) -- and, in the force pair, the scalar factors collected first (force_pair).  ~9
// correctly rounded divisions and a square root cost ~110 of the ~180 instructions of one IEEE force pair.
template <bool IEEE>
__device__ __forceinline__ float ws_sqrt(float x)
{
    if constexpr (IEEE) return sqrtf(x);
    else return __builtin_amdgcn_sqrtf(x);
}

template <bool IEEE>
struct WsDivisor {
    float v;
    __device__ __forceinline__ explicit WsDivisor(float den) : v(IEEE ? den : __builtin_amdgcn_rcpf(den)) {}
    __device__ __forceinline__ float operator()(float num) const { return IEEE ? num / v : num * v; }
};

// simulation.wgsl:176-183; d2 has already passed the radius test
template <bool IEEE>
__device__ __forceinline__ void density_pair(const WsDev &d, float d2, float &density, float &near_density,
                                             uint32_t mult)
{
    const float dst = ws_sqrt<IEEE>(d2);
    const float w = sk_density(d, dst), wn = sk_near(d, dst);
    for (uint32_t r = 0; r < mult; r++) {
        density += w;
        near_density += wn;
    }
}

struct ForceAcc {
    float pfx, pfy, pfz, vfx, vfy, vfz;
};

// simulation.wgsl:238-263.  (ex,ey,ez) = neighbour.pred - own.pred, d2 its squared length.
// nrho_x / nrho_y = the neighbour's density / near density; its pressures are recomputed
// from them (simulation.wgsl:192-193: the same two IEEE operations K4 would have stored).
template <bool IEEE>
__device__ __forceinline__ void force_pair(const WsDev &d, float ex, float ey, float ez, float d2, float nrho_x,
                                           float nrho_y, float4 nvel, float4 vel, float pressure, float near_pressure,
                                           ForceAcc &a, uint32_t mult)
{
    const float dst = ws_sqrt<IEEE>(d2);
    const float npress = d.pressure_scalar * (nrho_x - d.target_density);
    const float nnear = d.near_pressure_scalar * nrho_y;
    const float slope = sk_der(d, dst);
    const float shared = (pressure + npress) / 2.f;
    const float slope_near = sk_der_near(d, dst);
    const float shared_near = (near_pressure + nnear) / 2.f;
    if constexpr (IEEE) {
        // the reference's expression tree, operation for operation: ((dir * shared) * slope) / density per term
        const WsDivisor<IEEE> by_dst(dst), by_rho(nrho_x), by_near_rho(nrho_y);
        if (dst > 0.f) {
            ex = by_dst(ex);
            ey = by_dst(ey);
            ez = by_dst(ez);
        } else {
            ex = 0.f;
            ey = 1.f;
            ez = 0.f;
        }
        const float ax = by_rho(ex * shared * slope), ay = by_rho(ey * shared * slope), az = by_rho(ez * shared * slope);
        const float bx = by_near_rho(ex * shared_near * slope_near), by = by_near_rho(ey * shared_near * slope_near),
                    bz = by_near_rho(ez * shared_near * slope_near);
        const float visc = sk_visc(d, dst);
        const float wx = (nvel.x - vel.x) * visc, wy = (nvel.y - vel.y) * visc, wz = (nvel.z - vel.z) * visc;
        for (uint32_t r = 0; r < mult; r++) {
            a.pfx += ax; a.pfy += ay; a.pfz += az;
            a.pfx += bx; a.pfy += by; a.pfz += bz;
            a.vfx += wx; a.vfy += wy; a.vfz += wz;
        }
    } else {
        // Default arithmetic: the same pair terms with the scalar factors collected before they meet the direction
        // vector -- dir * (shared * slope / rho + shared_near * slope_near / rho_near), divisions as reciprocal
        // multiplies (hardware v_rcp_f32 / v_sqrt_f32, 1 ULP each) -- 17 of the ~65 operations of a pair fewer.  Each
        // term differs from the reference's expression tree by a few ULP; the sums stay inside the tolerance every
        // parity test applies to both arithmetics.  (Also measured: the neighbour's reciprocals prepared
        // once per particle by K4 instead -- 9 operations fewer still, no faster: K5 is then no longer bound by them.)
        const float sa = shared * slope * __builtin_amdgcn_rcpf(nrho_x);
        const float sb = shared_near * slope_near * __builtin_amdgcn_rcpf(nrho_y);
        const float f = sa + sb;
        const float g = f * __builtin_amdgcn_rcpf(dst);
        const bool apart = dst > 0.f;  // coincident particles: direction (0, 1, 0), simulation.wgsl:243-246
        const float fx = apart ? ex * g : 0.f, fy = apart ? ey * g : f, fz = apart ? ez * g : 0.f;
        const float hv = d.h * d.h - d2;
        const float visc = hv * hv * hv * d.k_spikey;
        const float wx = (nvel.x - vel.x) * visc, wy = (nvel.y - vel.y) * visc, wz = (nvel.z - vel.z) * visc;
        for (uint32_t r = 0; r < mult; r++) {
            a.pfx += fx; a.pfy += fy; a.pfz += fz;
            a.vfx += wx; a.vfy += wy; a.vfz += wz;
        }
    }
}

// K4 epilogue (simulation.wgsl:186-194).  Density and near density ride in the w lanes of the
// particle's sorted predicted position and velocity, so K5 gets {pred.xyz, density} and
// {vel.xyz, near density} of a neighbour in two 16-B loads.  Other workgroups are still
// reading pred.xyz while this w is written: different floats, never the same memory location.
__device__ __forceinline__ void density_store(float density, float near_density, uint32_t i, WsSorted srt)
{
    density = density + 0.00001f;  // DENSITY_PADDING, simulation.wgsl:4,187-188
    near_density = near_density + 0.00001f;
    srt.pred(i).w = density;
    srt.vel(i).w = near_density;
}

// ---- slab handles: migration, part 1 -------------------------------------------------------------------------

// message header: 8 words in front of every fixed-capacity message
//   [0] records in the message   [1] sender's sticky error bits   [2] sender's owned count   [3] step stamp
//   [4] the most records either of the sender's two migration messages of the previous exchange wanted to carry
//   [5] the larger of the sender's two boundary-layer populations at the previous step's halo
//   [6] the most records any ONE of the sender's far messages of this exchange wanted to carry   [7] reserved
// ([1..6] matter in the far messages only -- one per destination rank, exchanged all-to-all, each headed by the same
// status words: their headers are the per-step status table of all ranks; from [4] / [5] / [6] every rank sizes the
// next steps' messages -- the same table everywhere, hence the same sizes)
#define WS_HDR_WORDS 8u
// A migrating particle travels as 32 bytes: {position, id} and {velocity, 0}.  Its predicted position is not sent: it is
// position + velocity * look-ahead of exactly these values (simulation.wgsl:307), the receiver computes the same bits
// (as k_reorder does every step), and its destination is the message it travels in.
#define WS_MIG_REC_WORDS 8u

// The particle in slot i (not in the histogram, its cell id already WS_DEAD) left the slab: write its 32-byte record
// {pos+id, vel} into the message for its route -- left neighbour, right neighbour, or the "far"
// message addressed to its destination rank (one per rank, laid end to end at the stride of the size the next exchange
// will have: a particle that crosses more than one slab in a step) -- and remember the hole it leaves.  A message's record count lives in its header word 0 and is counted there directly (the receiver clamps
// it to the capacity; the other header words were written by the previous k_migrate_fill).  Order is free
// everywhere here: the sort is canonical.
__device__ __forceinline__ void migrate_out(const WsDev &d, const WsMig &m, uint32_t i, float4 pos, float4 vel, float4 pred)
{
    const uint32_t gxg = (uint32_t)grid_layer_x(d, pred.x);
    uint32_t dest = 0;
    while (dest + 1 < m.world && gxg >= m.cuts[dest + 1]) dest++;
    const uint32_t hs = atomicAdd(&m.dyn[DY_NHOLE], 1u);
    if (hs < m.hole_cap) m.hole[hs] = i;
    uint32_t *msg;
    uint32_t cap;
    const uint32_t mig_cap = d.mig_limit ? min(d.mig_limit, m.mig_cap) : m.mig_cap;  // what the next exchange will carry
    if (dest + 1u == m.me) {
        msg = m.sendL; cap = mig_cap;
    } else if (dest == m.me + 1u) {
        msg = m.sendR; cap = mig_cap;
    } else {
        cap = d.far_limit ? min(d.far_limit, m.far_cap) : m.far_cap;
        msg = m.far + (size_t)min(dest, m.world - 1u) * (WS_HDR_WORDS + (size_t)cap * WS_MIG_REC_WORDS);
        atomicAdd(&m.dyn[DY_FAR], 1u);
    }
    const uint32_t slot = atomicAdd(&msg[0], 1u);
    if (slot >= cap || dest == m.me) {  // the message is full (or the grids disagree): the particle is lost, the step is invalid -- say so
        atomicOr(&m.dyn[DY_ERR], WS_DYN_ERR_MIGRATION);
        return;
    }
    float4 *rec = reinterpret_cast<float4 *>(msg + WS_HDR_WORDS) + 2 * (size_t)slot;
    rec[0] = pos;
    rec[1] = vel;
}

// The new record of a particle -- {position, id}, {velocity, cell id} -- as eight scalars, and whether there is one to
// store (not in the ACCEL_ONLY pass, not for shadow lanes).
struct WsNewRecord {
    float px, py, pz, pw, vx, vy, vz, vw;
    int have;
};

// Store the wave's new records with FULL-LINE stores.  A lane holds both halves of its particle's 32-byte record; stored
// as they are, the two 16-byte stores of a wave would each touch every second half-line (measured: K5 +13 % in the sparse
// state).  So lanes exchange halves first: for each half of the wave, lane l takes half (l & 1) of the record of lane
// 32 h + (l >> 1) -- consecutive lanes then write consecutive 16 bytes (the wave's particles are consecutive slots except
// across the jump of a slab's two-range span, which only splits one store in two).  Every lane of the wave must call.
__device__ __forceinline__ void store_records(const WsDev &d, WsSoA out, uint32_t i, const WsNewRecord &r)
{
    const int lane = threadIdx.x & 63;
    const bool odd = (lane & 1) != 0;
    // the lanes that have a record are a prefix of the wave (lanes past the end of the range shadow its last particle)
    const int nhave = __popcll(__ballot(r.have != 0));
    // ... and their particles are consecutive slots, except across the jump of a slab's two-range span
    const uint32_t i0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)i);
    auto from = [](int byte_addr, float v) { return __int_as_float(__builtin_amdgcn_ds_bpermute(byte_addr, __float_as_int(v))); };
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const int src = 32 * h + (lane >> 1), a = src << 2;
        const uint32_t si = d.dyn ? (uint32_t)__builtin_amdgcn_ds_bpermute(a, (int)i) : i0 + (uint32_t)src;
        const float px = from(a, r.px), py = from(a, r.py), pz = from(a, r.pz), pw = from(a, r.pw);
        const float vx = from(a, r.vx), vy = from(a, r.vy), vz = from(a, r.vz), vw = from(a, r.vw);
        const float4 w = make_float4(odd ? vx : px, odd ? vy : py, odd ? vz : pz, odd ? vw : pw);
        if (src < nhave) out.pv[2 * (size_t)si + (odd ? 1u : 0u)] = w;
    }
}

// K5 epilogue (simulation.wgsl:265-268) + K6 integrate (:279-309) + next step's K1 binning.
// The acceleration itself (the reference's `acceleration` field, read by nothing but the 80-byte record view) is not
// stored by the step: ACCEL_ONLY = the same kernel run again over the same sorted state, on demand, storing just
// that (ws_read_particles; same code, same inputs, same visit order: the bits the step used).
// Leaves the particle's new record in `rec` (the caller stores the wave's records together: store_records).
template <bool ACCEL_ONLY>
__device__ __forceinline__ void force_integrate_bin(const WsDev &d, const ForceAcc &a, float rho_x, float4 vel, uint32_t i,
                                                    const float4 *__restrict__ pos, WsSoA out, float4 *__restrict__ accel,
                                                    uint32_t *__restrict__ cid_out, uint32_t *__restrict__ count,
                                                    WsNewRecord &rec)
{
    const float accx = a.pfx / rho_x + a.vfx * d.viscosity;
    const float accy = a.pfy / rho_x + a.vfy * d.viscosity;
    const float accz = a.pfz / rho_x + a.vfz * d.viscosity;
    if constexpr (ACCEL_ONLY) {
        accel[i] = make_float4(accx, accy, accz, 0.f);
        return;
    }

    const float4 p0 = pos[i];
    float vx = vel.x + (d.grav[0] + accx) * d.dt;
    float vy = vel.y + (d.grav[1] + accy) * d.dt;
    float vz = vel.z + (d.grav[2] + accz) * d.dt;
    float px = p0.x + vx * d.dt;
    float py = p0.y + vy * d.dt;
    float pz = p0.z + vz * d.dt;
    const float nd = -1.f * d.damping;
    if (px < d.ext_min[0]) { vx *= nd; px = d.ext_min[0]; } else if (px > d.ext_max[0]) { vx *= nd; px = d.ext_max[0]; }
    if (py < d.ext_min[1]) { vy *= nd; py = d.ext_min[1]; } else if (py > d.ext_max[1]) { vy *= nd; py = d.ext_max[1]; }
    if (pz < d.ext_min[2]) { vz *= nd; pz = d.ext_min[2]; } else if (pz > d.ext_max[2]) { vz *= nd; pz = d.ext_max[2]; }
    const float qx = px + vx * WS_LOOKAHEAD, qy = py + vy * WS_LOOKAHEAD, qz = pz + vz * WS_LOOKAHEAD;
    // (the predicted position is not stored: k_reorder recomputes it from the record, same bits)

    // next step's hash_particles (simulation.wgsl:130-141) on the dense grid
    const uint32_t nc = grid_cell(d, qx, qy, qz);
    // On a slab handle a particle whose new cell lies in a ghost layer (the local grid clamps x into its layers) has
    // left the slab: it is not binned, its record goes into a migration message instead (rare: a handful per step).
    bool stays = true;
    if (d.mig) {
        const uint32_t rowy = (uint32_t)(d.dim[1] * d.dim[2]);
        stays = nc >= rowy && nc < (uint32_t)(d.dim[0] - 1) * rowy;
    }
    const uint32_t cell = stays ? nc : WS_DEAD;
    cid_out[i] = cell;
    rec.px = px; rec.py = py; rec.pz = pz; rec.pw = p0.w;
    rec.vx = vx; rec.vy = vy; rec.vz = vz; rec.vw = __uint_as_float(cell);
    rec.have = 1;
    const uint32_t r = wave_run_atomic_inc(count, nc, stays);  // one atomic per run of lanes that moved into the same cell
    if (out.rank) out.rank[i] = r;  // arrival rank inside the cell: the sort's tentative slot, without more atomics
    if (!stays) migrate_out(d, *d.mig, i, make_float4(px, py, pz, p0.w), make_float4(vx, vy, vz, 0.f), make_float4(qx, qy, qz, 0.f));
}

// ---------------------------------------------------------------------------------
// variant "simple": one lane per particle, one candidate per loop trip, pair arithmetic
// inline.  The reference-shaped baseline of the A/B tests, the path used when the
// reference's hashed table would alias inside one stencil (tiny N: multiplicity table),
// and the fallback of the listed K5 for particles whose neighbour list overflowed.
// ---------------------------------------------------------------------------------
// CUTMODE: 0 = runs as they are, 1 = cut to the owned range (the early launches' instantiation of the listed kernels),
// 2 = decide at run time (the simple kernels: one instantiation serves every range)
template <bool ALIAS, bool IEEE, int CUTMODE = 2>
__device__ __forceinline__ void density_sweep_simple(const WsDev &d, const uint32_t *__restrict__ start, WsSorted srt,
                                                     const uint8_t *__restrict__ mult, float4 o, int c, float &density,
                                                     float &near_density)
{
    const int rowz = d.dim[2], rowy = d.dim[1] * d.dim[2];
    WsCut cut = {0u, 0xFFFFFFFFu, false};
    if constexpr (CUTMODE == 2) cut = ws_cut(d);
    for (int dx = -1; dx <= 1; dx++) {
        for (int dy = -1; dy <= 1; dy++) {
            const int cc = d.guard + c + dx * rowy + dy * rowz;
            uint32_t b = start[cc - 1], e = start[cc + 2];
            if constexpr (CUTMODE == 2) cut_run(cut, b, e);
            else cut_run_t<CUTMODE == 1>(d, b, e);
            for (uint32_t j = b; j < e; j++) {
                const float4 q = srt.pred(j);
                const float ex = q.x - o.x, ey = q.y - o.y, ez = q.z - o.z;
                const float d2 = ex * ex + ey * ey + ez * ez;
                if (d2 > d.d2_accept) continue;
                density_pair<IEEE>(d, d2, density, near_density, ALIAS ? alias_mult(d, mult, o, q) : 1u);
            }
        }
    }
}

template <bool ALIAS, bool IEEE, int CUTMODE = 2>
__device__ __forceinline__ void force_sweep_simple(const WsDev &d, const uint32_t *__restrict__ start, WsSorted srt,
                                                   const uint8_t *__restrict__ mult, uint32_t i, float4 o, float4 vel, int c,
                                                   float pressure, float near_pressure, ForceAcc &acc)
{
    const int rowz = d.dim[2], rowy = d.dim[1] * d.dim[2];
    WsCut cut = {0u, 0xFFFFFFFFu, false};
    if constexpr (CUTMODE == 2) cut = ws_cut(d);
    for (int dx = -1; dx <= 1; dx++) {
        for (int dy = -1; dy <= 1; dy++) {
            const int cc = d.guard + c + dx * rowy + dy * rowz;
            uint32_t b = start[cc - 1], e = start[cc + 2];
            if constexpr (CUTMODE == 2) cut_run(cut, b, e);
            else cut_run_t<CUTMODE == 1>(d, b, e);
            for (uint32_t j = b; j < e; j++) {
                if (j == i) continue;  // `particle_index == neighbour_index`, simulation.wgsl:232
                const float4 q = srt.pred(j);
                const float ex = q.x - o.x, ey = q.y - o.y, ez = q.z - o.z;
                const float d2 = ex * ex + ey * ey + ez * ez;
                if (d2 > d.d2_accept) continue;
                const float4 nvel = srt.vel(j);
                force_pair<IEEE>(d, ex, ey, ez, d2, q.w, nvel.w, nvel, vel, pressure, near_pressure, acc,
                                 ALIAS ? alias_mult(d, mult, o, q) : 1u);
            }
        }
    }
}

template <bool ALIAS, bool IEEE>
__global__ void __launch_bounds__(WS_BLOCK) k_density_simple(WsDev d, const uint32_t *__restrict__ start,
                                                             const uint32_t *__restrict__ cid_srt, WsSorted srt,
                                                             const uint8_t *__restrict__ mult)
{
    const WsSpan sp = ws_span(d, start);
    const uint32_t v = blockIdx.x * WS_BLOCK + threadIdx.x;
    if (v >= sp.len) return;
    const uint32_t i = span_at(sp, v);
    float density = 0.f, near_density = 0.f;
    density_sweep_simple<ALIAS, IEEE>(d, start, srt, mult, srt.pred(i), (int)cid_srt[i], density, near_density);
    density_store(density, near_density, i, srt);
}

template <bool ALIAS, bool IEEE, bool ACCEL_ONLY>
__global__ void __launch_bounds__(WS_BLOCK) k_force_simple(WsDev d, const uint32_t *__restrict__ start,
                                                           const uint32_t *__restrict__ cid_srt, WsSorted srt, WsSoA out,
                                                           float4 *__restrict__ accel, uint32_t *__restrict__ cid_out,
                                                           uint32_t *__restrict__ count, const uint8_t *__restrict__ mult)
{
    const WsSpan sp = ws_span(d, start);
    const uint32_t v = blockIdx.x * WS_BLOCK + threadIdx.x;
    if (sp.len == 0u || (blockIdx.x * WS_BLOCK + (threadIdx.x & ~63u)) >= sp.len) return;  // (whole waves only: a wave stores its records together)
    const bool valid = v < sp.len;
    const uint32_t i = span_at(sp, valid ? v : sp.len - 1u);  // lanes past the end shadow the last particle
    const float4 o = srt.pred(i);    // w = own density
    const float4 vel = srt.vel(i);   // w = own near density
    WsNewRecord rec = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0};
    if (valid) {
        const float pressure = d.pressure_scalar * (o.w - d.target_density);
        const float near_pressure = d.near_pressure_scalar * vel.w;
        ForceAcc acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        force_sweep_simple<ALIAS, IEEE>(d, start, srt, mult, i, o, vel, (int)cid_srt[i], pressure, near_pressure, acc);
        force_integrate_bin<ACCEL_ONLY>(d, acc, o.w, vel, i, srt.pos, out, accel, cid_out, count, rec);
    }
    if constexpr (!ACCEL_ONLY) store_records(d, out, i, rec);
}

// ---------------------------------------------------------------------------------
// variant "listed" (default): the MI355X kernels.
//
// The cell grid is z-fastest, so the 27 cells a particle searches are 9 contiguous particle
// runs (one per (dx,dy) column, three z-cells each) instead of 27 hashed bucket walks.
// One lane per particle; consecutive lanes are consecutive particles of the cell-sorted
// order, i.e. they sit in the same or neighbouring cells, so their candidate loads are the same
// or adjacent addresses and L1 serves them as broadcasts.
//
// K4 (density) makes the ONE pass over the candidates (~6x more of them than neighbours):
//   phase 1  radius test only, 4 candidates per trip from the planar x/y/z arrays; every accepted
//            candidate's d2 is pushed onto a per-lane list in LDS ([slot][lane]: conflict-free;
//            the store is unconditional, only the slot advance depends on the test), and ONE BIT
//            PER CANDIDATE in visit order goes to the particle's accept mask
//            (mask[word][particle]: coalesced, candidates / 8 bytes per particle);
//   phase 2  when some lane of the wave is blocked on a full list, every lane consumes its list:
//            the density terms (sqrt + kernel polynomials) on the accepted candidates only.
// K5 (force) accepts exactly the same candidates (same positions, same threshold; it only drops the
// particle itself) in the same order, so it does no radius tests: each lane iterates over the set
// bits of its mask, maps the bit number to the neighbour's index through a 9-entry per-lane run table
// in LDS, gathers {pred, density} and {vel, near density} (2 x 16 B, issued one neighbour ahead) and
// does the pair arithmetic -- then integrates and bins for the next step.
// A particle with more than 32 * ND_MASK_WORDS candidates keeps no mask and takes the simple sweep in K5 instead
// (same visit order, same operations); the other lanes of its wave still walk their masks.
// Visit order = (dx, dy, z, slot) ascending in every variant: all of them produce the same sums
// bit for bit.
// ---------------------------------------------------------------------------------
#ifndef ND_P
#define ND_P 64            // particles per K4 workgroup.  One wave: in the collapsing cloud neighbour counts vary
#endif                     // 10x between tiles, and a finished wave frees its slot at once (step 60: 0.38 -> 0.29 ms)
#ifndef ND_K
#define ND_K 16            // list fill level that triggers a flush
#endif
#define ND_ROWS (ND_K + 3) // a trip of 4 candidates may start at fill level K-1
#ifndef ND_MASK_WORDS
#define ND_MASK_WORDS 64   // 2048 candidates per particle (256 B of mask rows each; only the words in use are touched)
#endif


typedef float nd_f4 __attribute__((ext_vector_type(4)));
typedef float nd_f4u __attribute__((ext_vector_type(4), aligned(4)));  // 4 consecutive floats, 4-byte aligned

// Both ends of a run in ONE 16-byte load: the run of column cell `cc` is [start[cc - 1], start[cc + 2]), four
// consecutive table entries (the texture addresser is the busiest unit of both neighbour kernels in the sparse state;
// this halves their cell-table loads)
__device__ __forceinline__ void run_bounds(const uint32_t *__restrict__ start, int cc, uint32_t &b, uint32_t &e)
{
    const nd_u4u v = *reinterpret_cast<const nd_u4u *>(start + (cc - 1));
    b = v[0];
    e = v[3];
}

// K4's phase 1 over one run on the planar arrays: 4 candidates per trip from three 16-B loads, the
// squared distances as packed f32 vector arithmetic (same IEEE operations per candidate, same order).
// Lanes past their run's end keep loading in-bounds slots (the planes are padded) and are masked
// out of the accept test.  note(nvalid, bits): the trip tested nvalid candidates, bit u = candidate u accepted.
template <class Push, class Phase2, class Note>
__device__ __forceinline__ void nd_run_planar(const WsDev &d, float4 o, uint32_t j, uint32_t e, uint32_t &cnt,
                                              WsXYZ p, Push &&push, Phase2 &&phase2, Note &&note)
{
    // the walk keeps two numbers per lane: the byte offset of the next candidate in the planes (32-bit, from the three
    // uniform plane bases: no 64-bit address arithmetic) and how many candidates of the run remain
    uint32_t off = j * 4u;
    int32_t rem = (int32_t)(e - j);
    bool more = rem > 0;
    auto trip = [&]() {
        const nd_f4 X = *reinterpret_cast<const nd_f4u *>(reinterpret_cast<const char *>(p.x) + off);
        const nd_f4 Y = *reinterpret_cast<const nd_f4u *>(reinterpret_cast<const char *>(p.y) + off);
        const nd_f4 Z = *reinterpret_cast<const nd_f4u *>(reinterpret_cast<const char *>(p.z) + off);
        __builtin_amdgcn_sched_barrier(0);  // the three loads are issued before any is consumed
        const nd_f4 ex = X - o.x, ey = Y - o.y, ez = Z - o.z;
        const nd_f4 d2 = ex * ex + ey * ey + ez * ez;
        uint32_t bits = 0;
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const bool acc = (u < rem) && !(d2[u] > d.d2_accept);  // (u < rem: the candidate belongs to the run)
            push(cnt, d2[u]);  // branch-free: store always, advance the slot on accept
            cnt += acc ? 1u : 0u;
            bits |= (acc ? 1u : 0u) << u;
        }
        note((uint32_t)min(4, rem), bits);
        off += 16u;
        more = rem > 4;
        rem -= 4;
    };
    if (!__ballot(more && cnt + (uint32_t)rem > (uint32_t)ND_K)) {
        // no list of the wave can fill up in this run, even if every remaining candidate is accepted (the sparse
        // state, and short runs anywhere): the plain walk, each lane to the end of its run
        while (more) trip();
        return;
    }
    // Some list may fill up.  Wave-uniform walk: the moment a lane with candidates left has no room, EVERY lane
    // empties its list (phase 2) and all go on together.  (A lane that waited for the others to finish the run would
    // walk the rest of it alone afterwards -- whole trips of the wave for one lane: 101 instead of 83 trips per wave
    // in the settled C3 cloud.)  The order of the terms is the visit order whenever the lists are emptied.
    for (;;) {
        if (__ballot(more && cnt >= (uint32_t)ND_K)) {
            phase2(cnt);
            cnt = 0;
        }
        if (!__ballot(more)) break;
        if (more) trip();
    }
}

// One tile of K4: lane `threadIdx.x` works for particle iv (lanes past the end of their range shadow its last particle,
// valid = false).  `list`: the workgroup's LDS columns.
template <bool IEEE, bool CUT>
__device__ __forceinline__ void nd_tile(const WsDev &d, const uint32_t *__restrict__ start, const uint32_t *__restrict__ cid_srt,
                                        WsSorted srt, WsXYZ sxyz, WsMask mask, uint32_t *__restrict__ stats, float *list,
                                        const uint32_t iv, const bool valid, const uint32_t tid)
{
    const uint32_t i = iv;
    const float4 o = make_float4(sxyz.x[iv], sxyz.y[iv], sxyz.z[iv], 0.f);  // the planar copy: coalesced, same bits
    const int c = (int)cid_srt[iv];
    const int rowz = d.dim[2], rowy = d.dim[1] * d.dim[2];
    float density = 0.f, near_density = 0.f;
    // accept mask: bit `seq` = candidate number seq in visit order.  Trips append up to 4 bits to a 32-bit
    // accumulator; when a trip crosses into the next word the full word goes out (coalesced across lanes) and the
    // bits that spilled over start the next one -- the crossing is the rare path, one trip in eight.
    uint32_t acc32 = 0;
    uint32_t pos = 0, word = 0;  // candidates seen so far = 32 * word + pos
    uint32_t *mrow = mask.words + (iv - d.base);
    auto push = [&](uint32_t slot, float d2) { list[slot * ND_P + tid] = d2; };
    auto phase2 = [&](uint32_t cnt) {
        uint32_t k = 0;
        for (; k + 2u <= cnt; k += 2u) {  // two list entries per pass: their LDS reads and square roots overlap
            const float a = list[k * ND_P + tid], b = list[(k + 1u) * ND_P + tid];
            density_pair<IEEE>(d, a, density, near_density, 1u);
            density_pair<IEEE>(d, b, density, near_density, 1u);
        }
        if (k < cnt) density_pair<IEEE>(d, list[k * ND_P + tid], density, near_density, 1u);
    };
    auto note = [&](uint32_t nvalid, uint32_t bits) {
        acc32 |= bits << pos;
        pos += nvalid;
        if (pos >= 32u) {  // pos was >= 28, so the shift below is by 1..4
            if (word < ND_MASK_WORDS) mrow[(size_t)word * mask.stride] = acc32;
            pos -= 32u;
            acc32 = bits >> (nvalid - pos);  // the bits that spilled over: the trip's last `pos` ones
            word++;
        }
    };
    uint32_t cnt = 0;
    for (int p = 0; p < 3; p++) {  // dx = -1, 0, +1
        const int cc = d.guard + c + (p - 1) * rowy;
        // unconditional (iv is always a real particle), so the six loads go out together
        uint32_t b0, b1, b2, e0, e1, e2;
        run_bounds(start, cc - rowz, b0, e0);
        run_bounds(start, cc, b1, e1);
        run_bounds(start, cc + rowz, b2, e2);
        cut_run_t<CUT>(d, b0, e0);
        cut_run_t<CUT>(d, b1, e1);
        cut_run_t<CUT>(d, b2, e2);
        if (!valid) {
            e0 = b0;
            e1 = b1;
            e2 = b2;
        }
        nd_run_planar(d, o, b0, e0, cnt, sxyz, push, phase2, note);
        nd_run_planar(d, o, b1, e1, cnt, sxyz, push, phase2, note);
        nd_run_planar(d, o, b2, e2, cnt, sxyz, push, phase2, note);
    }
    phase2(cnt);
    if (valid) {
        if (pos && word < ND_MASK_WORDS) mrow[(size_t)word * mask.stride] = acc32;
        if (32u * word + pos > 32u * ND_MASK_WORDS) atomicAdd(&stats[0], 1u);  // rare by construction: one counter is enough
        density_store(density, near_density, i, srt);
    }
}

// SCHED = false: the whole range, or the span a slab launch names (ws_span); tiles dealt XCD-contiguously in equal shares.
// SCHED = true (full-range launches of a single-GPU handle): the cost-guided tile schedule (WsSched, ws_internal.h);
// the workgroup also measures how long its tile took -- next step's cost.
template <bool IEEE, bool CUT, bool SCHED>
__global__ void __launch_bounds__(ND_P) k_density_listed(WsDev d, const uint32_t *__restrict__ start,
                                                         const uint32_t *__restrict__ cid_srt, WsSorted srt, WsXYZ sxyz,
                                                         WsMask mask, uint32_t *__restrict__ stats, WsSched sched)
{
    __shared__ float list[ND_ROWS * ND_P];  // d2 of the accepted candidates
    if constexpr (SCHED) {
        const unsigned long long t0 = wall_clock64();
        const uint32_t x = blockIdx.x & 7u, j = blockIdx.x >> 3;
        const uint32_t s0 = sched.split[x], s1 = sched.split[x + 1];
        if (j >= s1 - s0) return;
        const uint32_t tile = sched.perm[s0 + j];
        const uint32_t v = tile * ND_P + threadIdx.x;
        const bool valid = v < d.n;
        nd_tile<IEEE, CUT>(d, start, cid_srt, srt, sxyz, mask, stats, list, d.base + (valid ? v : d.n - 1u), valid, threadIdx.x);
        if (threadIdx.x == 0) sched.cost[tile] = (uint32_t)min(wall_clock64() - t0, 0xFFFFFFull) + 1u;
    } else {
        const WsSpan sp = ws_span(d, start);
        const uint32_t ntiles = (sp.len + ND_P - 1u) / ND_P;  // <= gridDim.x: a slab launches over an upper bound
        if (blockIdx.x >= ntiles) return;
        const uint32_t v = xcd_tile(blockIdx.x, ntiles, true) * ND_P + threadIdx.x;  // (descending: see xcd_tile)
        const bool valid = v < sp.len;
        nd_tile<IEEE, CUT>(d, start, cid_srt, srt, sxyz, mask, stats, list, span_at(sp, valid ? v : sp.len - 1u), valid, threadIdx.x);  // lanes past the end shadow the last particle
    }
}

// Mask words per particle up to which a wave takes the in-step walk (nf_tile).  The walk gathers coherently (every
// lane inside the same pair of mask words) but keeps ONE neighbour in flight and pays, per pair, the busiest lane's
// bit count; the per-lane iterator keeps two in flight and pays the busiest lane's total.  A launch of many rounds of
// workgroups is bound by what its gathers cost the chip and gains from the walk in denser waves (word by word, 10
// against 8 words: K5 -1.8 % settled at C3, -1.3 % at C4; pair by pair, 12: -3.8 % more, 13: +-0, 14: +3 %); a launch
// of a round or two is as long as its slowest tile, and there the walk of a dense wave IS the slowest tile (10 words
// at 2^18 / 2^19 particles: K5 +24 % / +26 %; 6 against 8: -5 %).  2^20 and 2^21 particles: +-1 % either way.
// profiles/r05/ab/wordsync_by_launch_size.log, k5_walk_word_pairs.log
#ifndef NF_WORDSYNC_MAX
#define NF_WORDSYNC_MAX 6
#endif
#ifndef NF_WORDSYNC_MAX_BIG
#define NF_WORDSYNC_MAX_BIG 12       // ... in launches of at least
#endif
#ifndef NF_WORDSYNC_BIG_TILES
#define NF_WORDSYNC_BIG_TILES 16384  // tiles (2^21 particles: four and a half rounds of resident workgroups)
#endif
#ifndef NF_WORDSYNC_MAX_TINY
#define NF_WORDSYNC_MAX_TINY 4       // ... and in launches of fewer than
#endif
#ifndef NF_WORDSYNC_TINY_TILES
#define NF_WORDSYNC_TINY_TILES 2048  // tiles (the reference's 65 536 particles: every wave alone on its SIMD; 4 against 6 words: K5 -5 %, step -2 %; 2^17: +-0)
#endif
#ifndef NF_WORDSYNC_MAX_IEEE
#define NF_WORDSYNC_MAX_IEEE 4       // WS_FLAG_IEEE_DIVISION: a pair costs ~2.7 x the instructions, so does every lockstep trip of the walk (C3 settled K5, limit 2 / 4 / 6 / 8 / 10 / 12 words: 1.320 / 1.327 / 1.352 / 1.399 / 1.581 / 1.878 ms; tools/ieee_ab.sh)
#endif
// Particles per K5 workgroup (template parameter P of nf_tile / k_force_listed).  C3, 64 / 128 / 256 at step 60:
// 0.43 / 0.43 / 0.54 ms, step 200: 1.57 / 1.35 / 1.27 -- 128.  A launch of a round or two of workgroups is as long as
// its slowest tile, and one-wave tiles are shorter and deal finer: 64 below NF_SMALL_BELOW particles (K5 -11 % at the
// reference's 65 536, -5 % at 2^18 settled; +-0 at 2^19, +1.3 % at C3: profiles/r05/ab/k5_tile_by_launch_size.log)
#ifndef NF_P
#define NF_P 128
#endif
#ifndef NF_P_SMALL
#define NF_P_SMALL 64
#endif
#ifndef NF_SMALL_BELOW
#define NF_SMALL_BELOW (1u << 19)
#endif

// One tile of K5 (see nd_tile).  t_end / t_delta: the workgroup's LDS, a per-lane run table: candidate numbers
// [t_end[r-1], t_end[r]) belong to run r, neighbour = number + t_delta[r]
template <bool IEEE, bool ACCEL_ONLY, bool CUT, int P>
__device__ __forceinline__ void nf_tile(const WsDev &d, const uint32_t *__restrict__ start, const uint32_t *__restrict__ cid_srt,
                                        WsSorted srt, WsSoA out, float4 *__restrict__ accel, uint32_t *__restrict__ cid_out,
                                        uint32_t *__restrict__ count, WsMask mask, uint32_t *t_end, uint32_t *t_delta,
                                        const uint32_t iv, const bool valid, const uint32_t tid, const uint32_t wordsync_max)
{
    const uint32_t i = iv;
    const float4 o = srt.pred(iv);   // w = own density
    const float4 vel = srt.vel(iv);  // w = own near density
    const float pressure = d.pressure_scalar * (o.w - d.target_density);
    const float near_pressure = d.near_pressure_scalar * vel.w;
    ForceAcc acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int c = (int)cid_srt[iv];
    uint32_t total = 0;
    {
        // unconditional loads (iv is always a real particle), six in flight per plane
        const int rowz = d.dim[2], rowy = d.dim[1] * d.dim[2];
#pragma unroll
        for (int p = 0; p < 3; p++) {
            uint32_t b[3], e[3];
#pragma unroll
            for (int q = 0; q < 3; q++) {
                const int cc = d.guard + c + (p - 1) * rowy + (q - 1) * rowz;
                b[q] = start[cc - 1];  // (one 16-B load for both, as in K4: -4 % sparse, +4.5 % dense here)
                e[q] = start[cc + 2];
                cut_run_t<CUT>(d, b[q], e[q]);
            }
#pragma unroll
            for (int q = 0; q < 3; q++) {
                const int r = 3 * p + q;
                t_delta[r * P + tid] = b[q] - total;
                total += e[q] - b[q];
                t_end[r * P + tid] = total;
            }
        }
        t_end[9 * P + tid] = 0xFFFFFFFFu;
        if (!valid) total = 0;
    }
    // A particle with more candidates than the accept mask holds (> 2 048: a cell column of the settled floor layer in the
    // tall containers) has no mask and takes the plain sweep -- that particle alone: the other lanes of its wave walk
    // their masks (with the per-lane iterator: the in-step walk needs the whole wave).
    const bool over = total > 32u * ND_MASK_WORDS;
    const bool any_over = __ballot(over) != 0ull;
    if (!over) {
        const uint32_t *mrow = mask.words + (iv - d.base);
        const uint32_t nwords = (total + 31u) >> 5;
        uint32_t run = 0, end_r = t_end[tid], delta_r = t_delta[tid];
        const uint32_t self_s = i - t_delta[4 * P + tid];  // own candidate number (own cell = run 4)
      if (!any_over && !__ballot(nwords > wordsync_max)) {
        // A wave whose particles all have few candidates (the sparse state: 1-2 mask words each) walks its 64 masks
        // IN STEP, all lanes inside the same PAIR of mask words (64 candidates): inside a pair every lane takes its set
        // bits one per trip (the neighbours in visit order), so the wave's gathers stay inside a few cells.  The rare
        // work -- next pair, dropping the particle's own bit (simulation.wgsl:232 `particle_index == neighbour_index`)
        // -- is done by all lanes at once, once per pair; only the run switch stays per lane.  One neighbour's records
        // are in flight while the previous one computes.  The walk pays every pair's busiest lane (word by word, until
        // round 5, every WORD's: 9 x ~10 trips for 44 neighbours; a pair's busiest lane is less far above the average:
        // K5 -3.8 % settled at C3, -2.8 % at C4, -7 % at 65 536 particles; groups of four words lose the coherence
        // again: profiles/r05/ab/k5_walk_word_pairs.log), which is why a wave of denser particles takes the per-lane
        // iterator below (the limits: NF_WORDSYNC_MAX...).
        uint32_t wn0 = nwords ? mrow[0] : 0u, wn1 = nwords > 1u ? mrow[mask.stride] : 0u;  // one pair ahead
        bool pend = false;
        float4 q_p = o, nvel_p = vel;
        for (uint32_t w = 0; __ballot(w < nwords); w += 2u) {
            unsigned long long bits = (w < nwords ? (unsigned long long)wn0 : 0ull) | (w + 1u < nwords ? (unsigned long long)wn1 << 32 : 0ull);
            if ((w >> 1) == (self_s >> 6)) bits &= ~(1ull << (self_s & 63u));
            wn0 = w + 2u < nwords ? mrow[(size_t)(w + 2u) * mask.stride] : 0u;
            wn1 = w + 3u < nwords ? mrow[(size_t)(w + 3u) * mask.stride] : 0u;
            const uint32_t wbase = w << 5;
            while (__ballot(bits != 0ull)) {
                const bool has = bits != 0ull;
                float4 q_n = q_p, nvel_n = nvel_p;
                if (has) {
                    const uint32_t sc = wbase + (uint32_t)__ffsll((long long)bits) - 1u;
                    bits &= bits - 1ull;
                    while (sc >= end_r) {
                        run++;
                        end_r = t_end[run * P + tid];
                        delta_r = t_delta[min(run, 8u) * P + tid];
                    }
                    const uint32_t j = sc + delta_r;
                    q_n = srt.pred_near(j);
                    nvel_n = srt.vel_near(j);
                }
                if (pend) {
                    const float ex = q_p.x - o.x, ey = q_p.y - o.y, ez = q_p.z - o.z;
                    force_pair<IEEE>(d, ex, ey, ez, ex * ex + ey * ey + ez * ez, q_p.w, nvel_p.w, nvel_p, vel, pressure,
                                     near_pressure, acc, 1u);
                }
                pend = has;
                q_p = q_n;
                nvel_p = nvel_n;
            }
        }
        if (pend) {
            const float ex = q_p.x - o.x, ey = q_p.y - o.y, ez = q_p.z - o.z;
            force_pair<IEEE>(d, ex, ey, ez, ex * ex + ey * ey + ez * ez, q_p.w, nvel_p.w, nvel_p, vel, pressure, near_pressure,
                             acc, 1u);
        }
      } else {
        // Iterator over the set bits of the mask = the neighbours in visit order (the particle's own bit is dropped
        // when its word is fetched -- simulation.wgsl:232 `particle_index == neighbour_index`).
        uint32_t rest = 0, wbase = 0, widx = 0;
        uint32_t wnext = nwords ? mrow[0] : 0u;  // one mask word ahead
        // One pending-bits register per lane: a neighbour is ffs + clear + one compare against the end of the current
        // run.  The two rare steps -- next mask word, next run -- are short and independent, so a wave in which some
        // lane takes one of them on almost every trip (9 words and 9 runs per lane in the dense state) pays ~15
        // instructions for it, not the ~40 of a combined word-and-run segment computation.
        auto next = [&](uint32_t &j) -> bool {
            while (rest == 0u) {
                if (widx >= nwords) return false;
                rest = wnext;
                wbase = widx << 5;
                if (widx == (self_s >> 5)) rest &= ~(1u << (self_s & 31u));
                widx++;
                wnext = widx < nwords ? mrow[(size_t)widx * mask.stride] : 0u;
            }
            const uint32_t sc = wbase + (uint32_t)__ffs((int)rest) - 1u;
            rest &= rest - 1u;
            while (sc >= end_r) {
                run++;
                end_r = t_end[run * P + tid];
                delta_r = t_delta[min(run, 8u) * P + tid];
            }
            j = sc + delta_r;
            return true;
        };
        // Software pipeline: while neighbour k computes, the records of neighbour k+1 are in flight and the
        // iterator has already produced k+2.
        uint32_t j1 = iv, j2 = iv;  // next() leaves j alone when it fails: an exhausted lane keeps re-loading a valid record
        bool have0 = next(j1);
        float4 q_next = srt.pred_near(j1), nvel_next = srt.vel_near(j1);
        bool have1 = have0 && next(j2);
        while (have0) {
            const float4 q = q_next, nvel = nvel_next;
            q_next = srt.pred_near(j2);
            nvel_next = srt.vel_near(j2);
            have0 = have1;
            have1 = have1 && next(j2);
            const float ex = q.x - o.x, ey = q.y - o.y, ez = q.z - o.z;
            force_pair<IEEE>(d, ex, ey, ez, ex * ex + ey * ey + ez * ez, q.w, nvel.w, nvel, vel, pressure, near_pressure,
                             acc, 1u);
        }
      }
    } else if (valid) {
        force_sweep_simple<false, IEEE, CUT ? 1 : 0>(d, start, srt, nullptr, i, o, vel, c, pressure, near_pressure, acc);
    }
    WsNewRecord rec = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0};
    if (valid) force_integrate_bin<ACCEL_ONLY>(d, acc, o.w, vel, i, srt.pos, out, accel, cid_out, count, rec);
    if constexpr (!ACCEL_ONLY) store_records(d, out, i, rec);  // (every lane of the wave: the halves change lanes first)
}

// (SCHED: see k_density_listed)
template <bool IEEE, bool ACCEL_ONLY, bool CUT, bool SCHED, int P>
__global__ void __launch_bounds__(P) k_force_listed(WsDev d, const uint32_t *__restrict__ start,
                                                       const uint32_t *__restrict__ cid_srt, WsSorted srt, WsSoA out,
                                                       float4 *__restrict__ accel, uint32_t *__restrict__ cid_out,
                                                       uint32_t *__restrict__ count, WsMask mask, WsSched sched)
{
    __shared__ uint32_t t_end[10 * P];  // row 9: a sentinel no candidate number reaches
    __shared__ uint32_t t_delta[9 * P];
    if constexpr (SCHED) {
        const unsigned long long t0 = wall_clock64();
        const uint32_t x = blockIdx.x & 7u, j = blockIdx.x >> 3;
        const uint32_t s0 = sched.split[x], s1 = sched.split[x + 1];
        if (j >= s1 - s0) return;
        const uint32_t tile = sched.perm[s0 + j];
        const uint32_t v = tile * P + threadIdx.x;
        const bool valid = v < d.n;
        nf_tile<IEEE, ACCEL_ONLY, CUT, P>(d, start, cid_srt, srt, out, accel, cid_out, count, mask, t_end, t_delta,
                                       d.base + (valid ? v : d.n - 1u), valid, threadIdx.x, (uint32_t)(IEEE ? NF_WORDSYNC_MAX_IEEE : NF_WORDSYNC_MAX));  // (scheduled launches are small ones)
        __syncthreads();  // (both waves of the tile are done)
        if (threadIdx.x == 0) sched.cost[tile] = (uint32_t)min(wall_clock64() - t0, 0xFFFFFFull) + 1u;
    } else {
        const WsSpan sp = ws_span(d, start);
        const uint32_t ntiles = (sp.len + P - 1u) / P;  // <= gridDim.x: a slab launches over an upper bound
        if (blockIdx.x >= ntiles) return;
        const uint32_t v = xcd_tile(blockIdx.x, ntiles) * P + threadIdx.x;
        const bool valid = v < sp.len;
        nf_tile<IEEE, ACCEL_ONLY, CUT, P>(d, start, cid_srt, srt, out, accel, cid_out, count, mask, t_end, t_delta,
                                       span_at(sp, valid ? v : sp.len - 1u), valid, threadIdx.x,  // lanes past the end shadow the last particle
                                       IEEE ? (uint32_t)NF_WORDSYNC_MAX_IEEE : ntiles >= (uint32_t)NF_WORDSYNC_BIG_TILES ? (uint32_t)NF_WORDSYNC_MAX_BIG
                                                                     : ntiles >= (uint32_t)NF_WORDSYNC_TINY_TILES ? (uint32_t)NF_WORDSYNC_MAX : (uint32_t)NF_WORDSYNC_MAX_TINY);
    }
}

uint32_t wsk_mask_words(void) { return ND_MASK_WORDS; }

// ev (optional): the launch carries its own start / stop events (hipExtLaunchKernelGGL: the dispatch packet's
// completion signal is time-stamped) -- per-kernel timing without separate event packets in front of and behind the
// kernel, each of which costs the stream 7-9 us
#define WS_LAUNCH(kernel, grid, block, s, ev, ...)                                                          \
    do {                                                                                                    \
        if (ev) hipExtLaunchKernelGGL(kernel, grid, block, 0, s, (ev)->a, (ev)->b, 0, __VA_ARGS__);         \
        else hipLaunchKernelGGL(kernel, grid, block, 0, s, __VA_ARGS__);                                    \
    } while (0)

// workgroups of a scheduled launch: eight XCDs x the longest part of `perm` k_schedule ever gives one of them
uint32_t wsk_sched_grid(uint32_t ntiles) { return 8u * (ntiles / 8u + ntiles / 16u + 2u); }
uint32_t wsk_density_tile(void) { return ND_P; }
uint32_t wsk_force_tile(uint32_t n) { return n < (uint32_t)NF_SMALL_BELOW ? (uint32_t)NF_P_SMALL : (uint32_t)NF_P; }

template <bool IEEE>
static void launch_density(hipStream_t s, const WsDev &d, const uint32_t *start, const uint32_t *cid_srt, WsSorted srt,
                           const uint8_t *mult, bool alias, int variant, uint32_t *stats, WsMask mask, WsXYZ sxyz,
                           const WsEventPair *ev, WsSched sched)
{
    if (alias)
        WS_LAUNCH((k_density_simple<true, IEEE>), dim3(cdiv(d.n, WS_BLOCK)), dim3(WS_BLOCK), s, ev, d, start, cid_srt, srt, mult);
    else if (variant == WS_VARIANT_SIMPLE)
        WS_LAUNCH((k_density_simple<false, IEEE>), dim3(cdiv(d.n, WS_BLOCK)), dim3(WS_BLOCK), s, ev, d, start, cid_srt, srt, mult);
    else if (d.dyn && d.range_sel == WS_RANGE_EARLY)
        WS_LAUNCH((k_density_listed<IEEE, true, false>), dim3(cdiv(d.n, ND_P)), dim3(ND_P), s, ev, d, start, cid_srt, srt, sxyz, mask, stats, sched);
    else if (sched.perm && !d.dyn)
        WS_LAUNCH((k_density_listed<IEEE, false, true>), dim3(wsk_sched_grid(cdiv(d.n, ND_P))), dim3(ND_P), s, ev, d, start, cid_srt, srt, sxyz, mask, stats, sched);
    else
        WS_LAUNCH((k_density_listed<IEEE, false, false>), dim3(cdiv(d.n, ND_P)), dim3(ND_P), s, ev, d, start, cid_srt, srt, sxyz, mask, stats, sched);
}

// (d.n: the particle count of a single-GPU handle, a host-side upper bound of the range on a slab)
template <bool IEEE, bool ACCEL_ONLY, int P>
static void launch_force_listed(hipStream_t s, const WsDev &d, const uint32_t *start, const uint32_t *cid_srt, WsSorted srt, WsSoA out,
                                float4 *accel, uint32_t *cid_out, uint32_t *count, WsMask mask, const WsEventPair *ev, WsSched sched)
{
    if (d.dyn && d.range_sel == WS_RANGE_EARLY)
        WS_LAUNCH((k_force_listed<IEEE, ACCEL_ONLY, true, false, P>), dim3(cdiv(d.n, P)), dim3(P), s, ev, d, start, cid_srt, srt, out,
                  accel, cid_out, count, mask, sched);
    else if (sched.perm && !d.dyn)
        WS_LAUNCH((k_force_listed<IEEE, ACCEL_ONLY, false, true, P>), dim3(wsk_sched_grid(cdiv(d.n, P))), dim3(P), s, ev, d, start, cid_srt, srt, out,
                  accel, cid_out, count, mask, sched);
    else
        WS_LAUNCH((k_force_listed<IEEE, ACCEL_ONLY, false, false, P>), dim3(cdiv(d.n, P)), dim3(P), s, ev, d, start, cid_srt, srt, out,
                  accel, cid_out, count, mask, sched);
}

template <bool IEEE, bool ACCEL_ONLY>
static void launch_force(hipStream_t s, const WsDev &d, const uint32_t *start, const uint32_t *cid_srt, WsSorted srt,
                         WsSoA out, float4 *accel, uint32_t *cid_out, uint32_t *count, const uint8_t *mult, bool alias,
                         int variant, WsMask mask, const WsEventPair *ev, WsSched sched)
{
    if (alias)
        WS_LAUNCH((k_force_simple<true, IEEE, ACCEL_ONLY>), dim3(cdiv(d.n, WS_BLOCK)), dim3(WS_BLOCK), s, ev, d, start, cid_srt,
                  srt, out, accel, cid_out, count, mult);
    else if (variant == WS_VARIANT_SIMPLE)
        WS_LAUNCH((k_force_simple<false, IEEE, ACCEL_ONLY>), dim3(cdiv(d.n, WS_BLOCK)), dim3(WS_BLOCK), s, ev, d, start, cid_srt,
                  srt, out, accel, cid_out, count, mult);
    else if (wsk_force_tile(d.n) == (uint32_t)NF_P_SMALL)
        launch_force_listed<IEEE, ACCEL_ONLY, NF_P_SMALL>(s, d, start, cid_srt, srt, out, accel, cid_out, count, mask, ev, sched);
    else
        launch_force_listed<IEEE, ACCEL_ONLY, NF_P>(s, d, start, cid_srt, srt, out, accel, cid_out, count, mask, ev, sched);
}

void wsk_density(hipStream_t s, const WsDev &d, const uint32_t *start, const uint32_t *cid_srt, WsSorted srt,
                 const uint8_t *mult, bool alias, int variant, bool ieee, uint32_t *stats, WsMask mask, WsXYZ sxyz,
                 const WsEventPair *ev, WsSched sched)
{
    if (ieee) launch_density<true>(s, d, start, cid_srt, srt, mult, alias, variant, stats, mask, sxyz, ev, sched);
    else launch_density<false>(s, d, start, cid_srt, srt, mult, alias, variant, stats, mask, sxyz, ev, sched);
}

// accel_only: nothing but accel[i] is written (see force_integrate_bin) -- the on-demand pass of the record views
void wsk_force(hipStream_t s, const WsDev &d, const uint32_t *start, const uint32_t *cid_srt, WsSorted srt, WsSoA out,
               float4 *accel, uint32_t *cid_out, uint32_t *count, const uint8_t *mult, bool alias, int variant, bool ieee,
               WsMask mask, bool accel_only, const WsEventPair *ev, WsSched sched)
{
    if (accel_only) {
        if (ieee) launch_force<true, true>(s, d, start, cid_srt, srt, out, accel, cid_out, count, mult, alias, variant, mask, ev, sched);
        else launch_force<false, true>(s, d, start, cid_srt, srt, out, accel, cid_out, count, mult, alias, variant, mask, ev, sched);
    } else {
        if (ieee) launch_force<true, false>(s, d, start, cid_srt, srt, out, accel, cid_out, count, mult, alias, variant, mask, ev, sched);
        else launch_force<false, false>(s, d, start, cid_srt, srt, out, accel, cid_out, count, mult, alias, variant, mask, ev, sched);
    }
}

// ---------------------------------------------------------------------------------
// tile schedule (WsSched, ws_internal.h): from the costs the neighbour kernels measured in the previous step to
// split / perm for this one.  16 workgroups: workgroup (kernel k, XCD x) builds XCD x's part of kernel k's perm.
// Every workgroup first derives the same eight cut points from the whole cost array (cheap: 4 B per tile from L2, and
// no workgroup has to wait for another); the cuts give every XCD the same summed cost in one contiguous range of
// tiles, within [1/2, 3/2] of an equal share of the tiles (the launch grid is sized for 3/2).  Inside its range a
// workgroup orders the tiles by eight cost classes, heaviest first, tile order kept inside a class (a stable counting
// sort made of block scans: no atomics).  A tile that has no cost yet counts as average.
// ---------------------------------------------------------------------------------
#define WS_SCHED_BLOCK 1024
#define WS_SCHED_CLASSES 8

__device__ __forceinline__ uint32_t block_excl_scan_1024(uint32_t v, uint32_t *s_w, uint32_t &total)
{
    // exclusive scan over the 1024 threads of the block; s_w: 16 words of LDS; two barriers
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t incl = wave_incl_scan(v);
    __syncthreads();  // (s_w may still be read from an earlier call)
    if (lane == 63) s_w[wave] = incl;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < WS_SCHED_BLOCK / 64; w++) {
        const uint32_t x = s_w[w];
        if (w < wave) base += x;
        tot += x;
    }
    total = tot;
    return base + incl - v;
}

__global__ void __launch_bounds__(WS_SCHED_BLOCK) k_schedule(WsSched s4, uint32_t nt4, WsSched s5, uint32_t nt5, uint32_t nclasses,
                                                             uint32_t group_particles, uint32_t tile5)
{
    __shared__ uint32_t s_w[WS_SCHED_BLOCK / 64];
    __shared__ uint32_t s_cut[9];
    __shared__ uint32_t s_max;
    const bool second = blockIdx.x >= 8u;
    const WsSched sc = second ? s5 : s4;
    const uint32_t T = second ? nt5 : nt4, x = blockIdx.x & 7u, tid = threadIdx.x;
    if (T == 0u) return;
    // tiles are classed in GROUPS of neighbours in the sorted order (a group = about one row of cells): a class is a set
    // of whole groups, so the tiles that run side by side still come from one neighbourhood
    const uint32_t G = max(group_particles / (second ? tile5 : (uint32_t)ND_P), 1u);
    // ---- the cuts: tile ranges of equal summed cost
    const uint32_t per = (T + WS_SCHED_BLOCK - 1u) / WS_SCHED_BLOCK;
    const uint32_t a = min(tid * per, T), b = min(a + per, T);
    uint32_t known = 0, sum = 0;
    for (uint32_t t = a; t < b; t++) {
        const uint32_t c = sc.cost[t];
        known += c ? 1u : 0u;
        sum += c;
    }
    uint32_t tot_known, tot_sum;
    block_excl_scan_1024(known, s_w, tot_known);
    block_excl_scan_1024(sum >> 10, s_w, tot_sum);  // (scaled: 2^20 tiles x 2^24 ticks do not fit 32 bits)
    const uint32_t fill = tot_known ? (uint32_t)min(max(((unsigned long long)tot_sum << 10) / tot_known, 1ull), 0xFFFFFFull) : 1u;  // an unmeasured tile counts as average
    // prefix of the FILLED costs (64-bit: 2^20 tiles x 2^24 ticks)
    unsigned long long mine = 0;
    for (uint32_t t = a; t < b; t++) {
        const uint32_t c = sc.cost[t];
        mine += c ? c : fill;
    }
    // 64-bit block scan as two 32-bit halves would need carries: costs are clamped to 2^24 and a thread sums at most
    // `per` <= 2^10 tiles of them, so scale the per-thread sums down to 32 bits instead (the cuts need no more precision)
    const uint32_t shift = 10u;
    const uint32_t mine_s = (uint32_t)(mine >> shift);
    uint32_t total_s;
    const uint32_t before_s = block_excl_scan_1024(mine_s, s_w, total_s);
    if (tid < 9u) s_cut[tid] = tid == 8u ? T : 0u;
    __syncthreads();
    if (total_s >= 8u) {
        // the thread whose tiles contain the k-th eighth of the total looks for the exact tile
        for (uint32_t k = 1; k < 8u; k++) {
            const unsigned long long target = (unsigned long long)total_s * k / 8u;
            if (target >= before_s && target < (unsigned long long)before_s + mine_s) {
                unsigned long long run = (unsigned long long)before_s << shift;
                const unsigned long long want = target << shift;
                uint32_t t = a;
                for (; t < b; t++) {
                    const uint32_t c = sc.cost[t];
                    run += c ? c : fill;
                    if (run > want) break;
                }
                s_cut[k] = min(t + 1u, T);
            }
        }
    } else if (tid == 0u) {
        for (uint32_t k = 1; k < 8u; k++) s_cut[k] = (uint32_t)((unsigned long long)T * k / 8u);
    }
    __syncthreads();
    if (tid == 0u) {
        // every XCD keeps between half and three halves of an equal share of the tiles (wsk_sched_grid), and the rest
        // must stay coverable by the XCDs still to come
        const uint32_t nmax = T / 8u + T / 16u + 1u, nmin = T / 16u;
        uint32_t prev = 0;
        for (uint32_t k = 1; k < 8u; k++) {
            const uint32_t left = 8u - k;  // XCDs after this cut
            uint32_t lo = prev + nmin, hi = prev + nmax;
            lo = max(lo, T > left * nmax ? T - left * nmax : 0u);
            hi = min(hi, T - min(T, left * nmin));
            uint32_t c = s_cut[k];
            c = min(max(c, lo), max(hi, lo));
            c = min(c, T);
            s_cut[k] = c;
            prev = c;
        }
        s_cut[0] = 0u;
        s_cut[8] = T;
        s_max = 0u;
    }
    __syncthreads();
    if (x == 0u && tid < 9u) sc.split[tid] = s_cut[tid];
    // ---- this XCD's part of perm: descending cost classes
    const uint32_t r0 = s_cut[x], r1 = s_cut[x + 1u], n = r1 - r0;
    if (n == 0u) return;
    // a thread owns whole groups: ceil(groups of the range / 1024) of them
    const uint32_t g0 = r0 / G, g1 = (r1 + G - 1u) / G, ng = g1 - g0;
    const uint32_t gper = (ng + WS_SCHED_BLOCK - 1u) / WS_SCHED_BLOCK;
    const uint32_t ga = g0 + min(tid * gper, ng), gb = min(ga + gper, g1);
    auto group_cost = [&](uint32_t g) {  // mean filled cost of the group's tiles that lie in this range
        const uint32_t ta = max(g * G, r0), tb = min((g + 1u) * G, r1);
        unsigned long long sumc = 0;
        for (uint32_t t = ta; t < tb; t++) {
            const uint32_t c = sc.cost[t];
            sumc += c ? c : fill;
        }
        return tb > ta ? (uint32_t)(sumc / (tb - ta)) : 0u;
    };
    uint32_t mx = 0;
    for (uint32_t g = ga; g < gb; g++) mx = max(mx, group_cost(g));
#pragma unroll
    for (int sh = 32; sh >= 1; sh >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, sh, 64));
    if ((tid & 63u) == 0u) atomicMax(&s_max, mx);
    __syncthreads();
    const uint32_t cmax = s_max + 1u;
    auto cls_of = [&](uint32_t v) {  // 0 = the heaviest part of the cost range
        return (uint32_t)(((unsigned long long)(cmax - 1u - min(v, cmax - 1u)) * nclasses) / cmax);
    };
    uint32_t cnt[WS_SCHED_CLASSES];
#pragma unroll
    for (int k = 0; k < WS_SCHED_CLASSES; k++) cnt[k] = 0u;
    for (uint32_t g = ga; g < gb; g++) {
        const uint32_t k = cls_of(group_cost(g));
        const uint32_t m = min((g + 1u) * G, r1) - max(g * G, r0);
#pragma unroll
        for (int q = 0; q < WS_SCHED_CLASSES; q++) cnt[q] += (k == (uint32_t)q) ? m : 0u;
    }
    uint32_t off[WS_SCHED_CLASSES], cbase = 0;
#pragma unroll
    for (int k = 0; k < WS_SCHED_CLASSES; k++) {
        uint32_t tot;
        off[k] = cbase + block_excl_scan_1024(cnt[k], s_w, tot);
        cbase += tot;
    }
    for (uint32_t g = ga; g < gb; g++) {
        const uint32_t k = cls_of(group_cost(g));
        const uint32_t ta = max(g * G, r0), tb = min((g + 1u) * G, r1);
        uint32_t dst = 0;
#pragma unroll
        for (int q = 0; q < WS_SCHED_CLASSES; q++)
            if (k == (uint32_t)q) {
                dst = off[q];
                off[q] += tb - ta;
            }
        for (uint32_t t = ta; t < tb; t++) sc.perm[r0 + dst + (t - ta)] = t;
    }
}

void wsk_schedule(hipStream_t s, WsSched s4, uint32_t ntiles4, WsSched s5, WsSched s5_costs, uint32_t ntiles5, uint32_t nclasses,
                  uint32_t group_particles, uint32_t tile5)
{
    nclasses = std::min<uint32_t>(std::max<uint32_t>(nclasses, 1u), WS_SCHED_CLASSES);
    s5.cost = s5_costs.cost;  // (the kernel reads costs and writes split / perm)
    hipLaunchKernelGGL(k_schedule, dim3(16), dim3(WS_SCHED_BLOCK), 0, s, s4, ntiles4, s5, ntiles5, nclasses, group_particles, tile5);
}

// ---------------------------------------------------------------------------------
// readback in original-id order (update(), src/fluid_compute.rs:478-485)
// ---------------------------------------------------------------------------------
__global__ void __launch_bounds__(WS_BLOCK) k_gather_positions(WsSoA cur, float *__restrict__ out, uint32_t n)
{
    const uint32_t i = blockIdx.x * WS_BLOCK + threadIdx.x;
    if (i >= n) return;
    const float4 p = cur.pos(i);
    const size_t id = __float_as_uint(p.w);
    out[3 * id] = p.x;
    out[3 * id + 1] = p.y;
    out[3 * id + 2] = p.z;
}

void wsk_gather_positions(hipStream_t s, WsSoA cur, float *out_xyz, uint32_t n)
{
    hipLaunchKernelGGL(k_gather_positions, dim3(cdiv(n, WS_BLOCK)), dim3(WS_BLOCK), 0, s, cur, out_xyz, n);
}

// `velocity.length()` per particle, the quantity of the reference's (commented-out) speed colouring,
// src/fluid_compute.rs:489-502; same left-to-right sum as the WGSL length() restatement
__global__ void __launch_bounds__(WS_BLOCK) k_gather_speeds(WsSoA cur, float *__restrict__ out, uint32_t n)
{
    const uint32_t i = blockIdx.x * WS_BLOCK + threadIdx.x;
    if (i >= n) return;
    const float4 v = cur.vel(i);
    out[__float_as_uint(cur.pos(i).w)] = sqrtf(v.x * v.x + v.y * v.y + v.z * v.z);
}

void wsk_gather_speeds(hipStream_t s, WsSoA cur, float *out, uint32_t n)
{
    hipLaunchKernelGGL(k_gather_speeds, dim3(cdiv(n, WS_BLOCK)), dim3(WS_BLOCK), 0, s, cur, out, n);
}

__global__ void __launch_bounds__(WS_BLOCK) k_gather_particles(WsDev d, WsSoA cur, WsSorted srt,
                                                               const float4 *__restrict__ accel, int have_step,
                                                               ws_particle80 *__restrict__ out, uint32_t n)
{
    const uint32_t i = blockIdx.x * WS_BLOCK + threadIdx.x;
    if (i >= n) return;
    const float4 p = cur.pos(i), v = cur.vel(i), q = cur.pred[i];
    const size_t id = __float_as_uint(p.w);
    float4 dp = make_float4(0.f, 0.f, 0.f, 0.f), a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (have_step) {
        dp.x = srt.pred(i).w;  // density / near density ride in the sorted copy's w lanes
        dp.y = srt.vel(i).w;
        dp.z = d.pressure_scalar * (dp.x - d.target_density);  // simulation.wgsl:192-193
        dp.w = d.near_pressure_scalar * dp.y;
        a = accel[i];
    }
    float4 *rec = reinterpret_cast<float4 *>(out + id);
    rec[0] = make_float4(p.x, p.y, p.z, 0.f);
    rec[1] = dp;
    rec[2] = make_float4(v.x, v.y, v.z, 0.f);
    rec[3] = make_float4(a.x, a.y, a.z, 0.f);
    rec[4] = make_float4(q.x, q.y, q.z, 0.f);
}

void wsk_gather_particles(hipStream_t s, const WsDev &d, WsSoA cur, WsSorted srt, const float4 *accel, bool have_step,
                          ws_particle80 *out, uint32_t n)
{
    hipLaunchKernelGGL(k_gather_particles, dim3(cdiv(n, WS_BLOCK)), dim3(WS_BLOCK), 0, s, d, cur, srt, accel,
                       have_step ? 1 : 0, out, n);
}

// ---------------------------------------------------------------------------------
// reference-layout sort view (diagnostic, on demand; never part of ws_step)
// ---------------------------------------------------------------------------------
__global__ void __launch_bounds__(WS_BLOCK) k_view_keys(WsDev d, WsSorted srt,
                                                        uint32_t *__restrict__ keys_by_id, uint32_t *__restrict__ count)
{
    const uint32_t i = blockIdx.x * WS_BLOCK + threadIdx.x;
    if (i >= d.n) return;
    const float4 p = srt.pred(i);
    const uint32_t id = __float_as_uint(srt.pos[i].w);
    const uint32_t key = ref_hash_key(d, p.x, p.y, p.z);
    keys_by_id[id] = key;
    atomicAdd(&count[key], 1u);
}

void wsk_view_keys(hipStream_t s, const WsDev &d, WsSorted srt, uint32_t *keys_by_id, uint32_t *count)
{
    hipLaunchKernelGGL(k_view_keys, dim3(cdiv(d.n, WS_BLOCK)), dim3(WS_BLOCK), 0, s, d, srt, keys_by_id, count);
}

// the histogram alone, for keys that are already there (slab handles: gathered by id)
__global__ void __launch_bounds__(WS_BLOCK) k_view_count(const uint32_t *__restrict__ keys, uint32_t *__restrict__ count, uint32_t n)
{
    const uint32_t i = blockIdx.x * WS_BLOCK + threadIdx.x;
    if (i < n) atomicAdd(&count[keys[i]], 1u);
}

void wsk_view_count(hipStream_t s, const uint32_t *keys, uint32_t *count, uint32_t n)
{
    hipLaunchKernelGGL(k_view_count, dim3(cdiv(n, WS_BLOCK)), dim3(WS_BLOCK), 0, s, keys, count, n);
}

// stable order inside each bucket: ascending particle id
__global__ void __launch_bounds__(WS_BLOCK) k_view_fix(const uint32_t *__restrict__ tmp, const uint32_t *__restrict__ keys,
                                                       const uint32_t *__restrict__ start, uint32_t *__restrict__ perm,
                                                       uint32_t n)
{
    const uint32_t s = blockIdx.x * WS_BLOCK + threadIdx.x;
    if (s >= n) return;
    const uint32_t id = tmp[s];
    const uint32_t k = keys[id];
    const uint32_t b = start[k], e = start[k + 1];
    uint32_t rank = 0;
    for (uint32_t t = b; t < e; t++) rank += (tmp[t] < id) ? 1u : 0u;
    perm[b + rank] = id;
}

void wsk_view_fix(hipStream_t s, const uint32_t *tmp, const uint32_t *keys, const uint32_t *start, uint32_t *perm,
                  uint32_t n)
{
    hipLaunchKernelGGL(k_view_fix, dim3(cdiv(n, WS_BLOCK)), dim3(WS_BLOCK), 0, s, tmp, keys, start, perm, n);
}

// cell_offsets[k] = first slot of key k, or INF (bitonic_sort.wgsl:48-59, simulation.wgsl:36)
__global__ void __launch_bounds__(WS_BLOCK) k_view_offsets(const uint32_t *__restrict__ start, uint32_t *__restrict__ off,
                                                           uint32_t n)
{
    const uint32_t k = blockIdx.x * WS_BLOCK + threadIdx.x;
    if (k >= n) return;
    const uint32_t b = start[k], e = start[k + 1];
    off[k] = (e > b) ? b : 999999999u;
}

void wsk_view_offsets(hipStream_t s, const uint32_t *start, uint32_t *off, uint32_t n)
{
    hipLaunchKernelGGL(k_view_offsets, dim3(cdiv(n, WS_BLOCK)), dim3(WS_BLOCK), 0, s, start, off, n);
}

// four words from the host into device memory BY A KERNEL (arguments, not a copy): what a transport then reads -- a
// kernel of RCCL's, on the same stream -- was written by the kernel in front of it, the one ordering every runtime fences.
// (A small host-to-device copy may go through the CPU's window into device memory; round 4's stand-in for librccl read
// the previous call's words out of the L2 behind it now and then.)
__global__ void k_set_words4(uint32_t *__restrict__ p, uint32_t a, uint32_t b, uint32_t c, uint32_t d)
{
    p[0] = a;
    p[1] = b;
    p[2] = c;
    p[3] = d;
}
void wsk_set_words4(hipStream_t s, uint32_t *p, uint32_t a, uint32_t b, uint32_t c, uint32_t d)
{
    hipLaunchKernelGGL(k_set_words4, dim3(1), dim3(1), 0, s, p, a, b, c, d);
}

__global__ void __launch_bounds__(WS_BLOCK) k_iota(uint32_t *__restrict__ p, uint32_t n)
{
    const uint32_t i = blockIdx.x * WS_BLOCK + threadIdx.x;
    if (i < n) p[i] = i;
}

void wsk_iota(hipStream_t s, uint32_t *p, uint32_t n)
{
    hipLaunchKernelGGL(k_iota, dim3(cdiv(n, WS_BLOCK)), dim3(WS_BLOCK), 0, s, p, n);
}

// ---------------------------------------------------------------------------------
// slab (multi-GPU) support kernels
//
// Nothing a slab step needs from the device ever goes through the host: the owned count, the boundary-layer
// ranges, the ghost counts and the migration counts live in a small device block (`dyn`, WsDyn words) and in the
// headers of fixed-capacity messages; kernels are launched over host-known UPPER BOUNDS and read the real
// numbers from there.  Every capacity overrun clamps (so nothing is written out of bounds) and sets a sticky bit
// in dyn[DY_ERR], which travels to every rank with the next step's far messages and fails ws_step on all of them.
// ---------------------------------------------------------------------------------
// Migration, part 1 for particles that no force kernel has seen yet (the first step after the upload; from then on
// the force kernel's epilogue does this for the particles it moves): take the ones binned into a ghost layer out of
// the histogram and hand them to migrate_out.  4 bytes per particle; only the leavers read their records.
__global__ void __launch_bounds__(WS_BLOCK) k_migrate_mark(WsDev d, WsSoA cur, uint32_t *__restrict__ cid_cur,
                                                           uint32_t *__restrict__ count)
{
    const uint32_t k = blockIdx.x * WS_BLOCK + threadIdx.x;
    if (k >= d.dyn[DY_N]) return;
    const uint32_t i = d.base + k;
    const uint32_t rowy = (uint32_t)(d.dim[1] * d.dim[2]), c = cid_cur[i];
    if (c >= rowy && c < (uint32_t)(d.dim[0] - 1) * rowy) return;
    atomicSub(&count[c], 1u);
    cid_cur[i] = WS_DEAD;
    const float4 p = cur.pos(i), vc = cur.vel(i), q = cur.pred[i];
    const float4 v = make_float4(vc.x, vc.y, vc.z, 0.f);
    cur.vel(i).w = __uint_as_float(WS_DEAD);
    // (loads hand every particle to the slab its PREDICTED position lies in, so nothing leaves here unless the host's own
    // assignment at ws_slab_create disagrees with the device's -- at rest.  A migration record does not carry the
    // predicted position: one that the receiver could not reproduce must not travel silently.)
    if (q.x != p.x + v.x * WS_LOOKAHEAD || q.y != p.y + v.y * WS_LOOKAHEAD || q.z != p.z + v.z * WS_LOOKAHEAD)
        atomicOr(&d.mig->dyn[DY_ERR], WS_DYN_ERR_PRED);
    migrate_out(d, *d.mig, i, p, v, q);
}

void wsk_migrate_mark(hipStream_t s, const WsDev &d, WsSoA cur, uint32_t *cid_cur, uint32_t *count)
{
    if (d.n) hipLaunchKernelGGL(k_migrate_mark, dim3(cdiv(d.n, WS_BLOCK)), dim3(WS_BLOCK), 0, s, d, cur, cid_cur, count);
}

// The headers of the `world` far messages of the next exchange (one per destination, `far_next` records each): no
// records yet, and the same status words in all of them.  (halo_now: the previous step's boundary-layer population --
// this step's halo has not been packed yet.)  Word 6 is k_far_seal's.
__device__ __forceinline__ void far_headers(uint32_t *__restrict__ far_send, uint32_t world, uint32_t far_next, uint32_t err,
                                            uint32_t n_owned, uint32_t stamp, uint32_t wanted_mig, uint32_t halo_now)
{
    const size_t words = WS_HDR_WORDS + (size_t)far_next * WS_MIG_REC_WORDS;
    for (uint32_t r = 0; r < world; r++) {
        uint32_t *hdr = far_send + r * words;
        hdr[0] = 0;
        hdr[1] = err;
        hdr[2] = n_owned;
        hdr[3] = stamp;
        hdr[4] = wanted_mig;
        hdr[5] = halo_now;
        hdr[6] = 0;
        hdr[7] = 0;
    }
}

// Right before the far messages travel: the largest of this rank's `world` record counts into word 6 of every header
// (the receivers see only the count of the message addressed to them; the sizes of the next steps' messages must come
// out the same on every rank, so they are derived from a number every rank receives).  far_cur = records per message.
__global__ void k_far_seal(uint32_t world, uint32_t far_cur, uint32_t *__restrict__ far_send, uint32_t *__restrict__ dyn)
{
    const size_t words = WS_HDR_WORDS + (size_t)far_cur * WS_MIG_REC_WORDS;
    uint32_t most = 0;
    for (uint32_t r = 0; r < world; r++) most = max(most, far_send[r * words]);
    for (uint32_t r = 0; r < world; r++) far_send[r * words + 6] = most;
    dyn[DY_PEAK_FAR] = max(dyn[DY_PEAK_FAR], most);
}

void wsk_far_seal(hipStream_t s, uint32_t world, uint32_t far_cur, uint32_t *far_send, uint32_t *dyn)
{
    hipLaunchKernelGGL(k_far_seal, dim3(1), dim3(1), 0, s, world, far_cur, far_send, dyn);
}

// Exact message sizes (WS_FLAG_EXACT_MESSAGES): what this rank's next messages carry, for the all-gather every rank sizes
// its transfers from -- {records in message a, in message b, word 6 of c (the largest far message, after k_far_seal),
// the owned count}.  A count may exceed the capacity (the writers count on); the host clamps.
__global__ void k_sizes(const uint32_t *__restrict__ a, const uint32_t *__restrict__ b, const uint32_t *__restrict__ c,
                        const uint32_t *__restrict__ dyn, uint32_t *__restrict__ out)
{
    out[0] = a ? a[0] : 0u;
    out[1] = b ? b[0] : 0u;
    out[2] = c ? c[6] : 0u;
    out[3] = dyn[DY_N];
}

void wsk_sizes(hipStream_t s, const uint32_t *a, const uint32_t *b, const uint32_t *c, const uint32_t *dyn, uint32_t *out)
{
    hipLaunchKernelGGL(k_sizes, dim3(1), dim3(1), 0, s, a, b, c, dyn, out);
}

// The `world` far messages, written at the stride of the full capacity, copied to the stride of the `far_n` records
// that travel (header + records each): the all-to-all takes equal, contiguous segments.
__global__ void __launch_bounds__(WS_BLOCK) k_far_pack(uint32_t far_cap, uint32_t far_n, const uint32_t *__restrict__ src,
                                                       uint32_t *__restrict__ dst)
{
    const size_t words_src = WS_HDR_WORDS + (size_t)far_cap * WS_MIG_REC_WORDS, words_dst = WS_HDR_WORDS + (size_t)far_n * WS_MIG_REC_WORDS;
    const uint32_t t = blockIdx.x * WS_BLOCK + threadIdx.x, r = blockIdx.y;
    if (t < words_dst / 4u) reinterpret_cast<uint4 *>(dst + r * words_dst)[t] = reinterpret_cast<const uint4 *>(src + r * words_src)[t];
}

void wsk_far_pack(hipStream_t s, uint32_t world, uint32_t far_cap, uint32_t far_n, const uint32_t *src, uint32_t *dst)
{
    const uint32_t quads = (WS_HDR_WORDS + far_n * WS_MIG_REC_WORDS) / 4u;
    hipLaunchKernelGGL(k_far_pack, dim3(cdiv(quads, WS_BLOCK), world), dim3(WS_BLOCK), 0, s, far_cap, far_n, src, dst);
}

// Migration, part 2 (ONE workgroup; a step moves a few thousand particles at most): count the arrivals, fix the new
// owned count, then close the holes.  The owned range shrinks / grows from n_old to n_new.  Targets = holes below
// n_new (+ the new slots when growing); sources = arrivals (tagged with their message in the top two bits) +
// surviving particles above n_new.  Both lists come out equally long.  Also copies the received far messages' headers
// into `status` (one 4-word row per rank: the host reads it two steps later).
#define WS_FILL_THREADS 1024
__global__ void __launch_bounds__(WS_FILL_THREADS) k_migrate_fill(WsDev d, uint32_t world, uint32_t me, uint32_t cap,
                                                                 uint32_t *__restrict__ dyn, const uint32_t *__restrict__ hole,
                                                                 const uint32_t *__restrict__ recvL,
                                                                 const uint32_t *__restrict__ recvR, uint32_t mig_cap,
                                                                 const uint32_t *__restrict__ far_all, uint32_t far_cap,
                                                                 uint32_t *__restrict__ tgt, uint32_t *__restrict__ src,
                                                                 WsSoA cur, uint32_t *__restrict__ cid_cur,
                                                                 uint32_t *__restrict__ count,
                                                                 uint32_t *__restrict__ status_ring, uint32_t status_slots,
                                                                 uint32_t hole_cap, uint32_t *__restrict__ sendL,
                                                                 uint32_t *__restrict__ sendR, uint32_t *__restrict__ far_send,
                                                                 uint32_t far_next)
{
    __shared__ uint32_t s_far, s_ntgt, s_nsrc, s_nnew, s_nold, s_arr;
    const uint32_t tid = threadIdx.x;
    const uint32_t step = dyn[DY_STEP];  // (nobody writes it before the last barrier below)
    uint32_t *status = status_ring + (size_t)(step % status_slots) * world * WS_HDR_WORDS;
    const uint32_t far_words = WS_HDR_WORDS + far_cap * WS_MIG_REC_WORDS;  // words per sending rank in the received far buffer
    const bool left = me > 0, right = me + 1 < world;
    if (tid == 0) {
        s_far = 0;
        s_ntgt = 0;
        s_nsrc = 0;
    }
    __syncthreads();
    // far arrivals: every record of the far messages the other ranks addressed to this one
    if (tid < world && tid != me) atomicAdd(&s_far, min(far_all[(size_t)tid * far_words], far_cap));
    if (tid < world) {
        const uint32_t *msg = far_all + (size_t)tid * far_words;
        for (uint32_t w = 0; w < WS_HDR_WORDS; w++) status[tid * WS_HDR_WORDS + w] = msg[w];
        // Every message carries the step it belongs to.  A transport that keeps its calls in issue order can never
        // deliver another step's message; if one arrives anyway (two streams driving one communicator out of order, a
        // rank one step ahead) say so instead of simulating on with it.
        if (msg[3] != step) atomicOr(&dyn[DY_ERR], WS_DYN_ERR_STAMP);
    }
    if (tid == 0 && ((left && recvL[3] != step) || (right && recvR[3] != step))) atomicOr(&dyn[DY_ERR], WS_DYN_ERR_STAMP);
    __syncthreads();
    const uint32_t nL = left ? min(recvL[0], mig_cap) : 0u, nR = right ? min(recvR[0], mig_cap) : 0u;
    const uint32_t leave = min(dyn[DY_NHOLE], hole_cap);
    if (tid == 0 && dyn[DY_NHOLE] > hole_cap) atomicOr(&dyn[DY_ERR], WS_DYN_ERR_MIGRATION);
    if (tid == 0) {
        const uint32_t n_old = dyn[DY_N];
        uint32_t arrivals = nL + nR + s_far;
        uint32_t n_new = n_old - leave + arrivals;
        if (n_new > cap) {  // cannot hold them: drop the excess arrivals, flag the step
            atomicOr(&dyn[DY_ERR], WS_DYN_ERR_CAPACITY);
            arrivals -= n_new - cap;
            n_new = cap;
        }
        s_nold = n_old;
        s_nnew = n_new;
        s_arr = arrivals;
    }
    __syncthreads();
    const uint32_t n_old = s_nold, n_new = s_nnew, base = d.base;
    // targets
    for (uint32_t t = tid; t < leave; t += WS_FILL_THREADS) {
        const uint32_t i = hole[t];
        if (i < base + n_new) tgt[atomicAdd(&s_ntgt, 1u)] = i;
    }
    const uint32_t grow = n_new > n_old ? n_new - n_old : 0u;
    for (uint32_t t = tid; t < grow; t += WS_FILL_THREADS) tgt[atomicAdd(&s_ntgt, 1u)] = base + n_old + t;
    // sources
    const uint32_t tail = n_old > n_new ? n_old - n_new : 0u;
    for (uint32_t t = tid; t < tail; t += WS_FILL_THREADS) {
        const uint32_t i = base + n_new + t;
        if (cid_cur[i] != WS_DEAD) src[atomicAdd(&s_nsrc, 1u)] = i;
    }
    for (uint32_t t = tid; t < nL; t += WS_FILL_THREADS) src[atomicAdd(&s_nsrc, 1u)] = (1u << 30) | t;
    for (uint32_t t = tid; t < nR; t += WS_FILL_THREADS) src[atomicAdd(&s_nsrc, 1u)] = (2u << 30) | t;
    for (uint32_t t = tid; t < world * far_cap; t += WS_FILL_THREADS) {
        const uint32_t q = t / far_cap, k = t % far_cap;
        if (q == me) continue;
        const uint32_t *msg = far_all + (size_t)q * far_words;
        if (k >= min(msg[0], far_cap)) continue;
        src[atomicAdd(&s_nsrc, 1u)] = (3u << 30) | t;
    }
    __syncthreads();
    // apply: source k fills target k (any pairing will do).  After a capacity overrun there are more sources than
    // targets and the excess is dropped.
    const uint32_t pairs = min(s_ntgt, s_nsrc);
    for (uint32_t t = tid; t < pairs; t += WS_FILL_THREADS) {
        const uint32_t to = tgt[t], from = src[t];
        const uint32_t tag = from >> 30, idx = from & 0x3FFFFFFFu;
        if (tag) {
            const uint32_t *msg = tag == 1u ? recvL : tag == 2u ? recvR : far_all + (size_t)(idx / far_cap) * far_words;
            const float4 *rec = reinterpret_cast<const float4 *>(msg + WS_HDR_WORDS) + 2 * (size_t)(tag == 3u ? idx % far_cap : idx);
            const float4 p = rec[0], v = rec[1];
            const float4 q = make_float4(p.x + v.x * WS_LOOKAHEAD, p.y + v.y * WS_LOOKAHEAD, p.z + v.z * WS_LOOKAHEAD, 0.f);
            const uint32_t c = grid_cell(d, q.x, q.y, q.z);
            cur.pos(to) = p;
            cur.vel(to) = make_float4(v.x, v.y, v.z, __uint_as_float(c));
            cur.pred[to] = q;
            cid_cur[to] = c;
            cur.rank[to] = atomicAdd(&count[c], 1u);  // the next rank of its cell (k_place)
        } else {
            cur.pos(to) = cur.pos(idx);
            cur.vel(to) = cur.vel(idx);
            cur.pred[to] = cur.pred[idx];
            cid_cur[to] = cid_cur[idx];
            cur.rank[to] = cur.rank[idx];
        }
    }
    __syncthreads();
    if (tid == 0) {
        dyn[DY_N] = n_new;
        dyn[DY_STEP] = step + 1u;
        dyn[DY_NHOLE] = 0;
        dyn[DY_ARRIVED] += s_arr;
        dyn[DY_LEFT] += leave;
        const uint32_t wanted = max(sendL[0], sendR[0]);  // (the counts of the messages just exchanged, clamped or not)
        dyn[DY_PEAK_MIG] = max(dyn[DY_PEAK_MIG], wanted);
        // the outgoing migration messages of the NEXT step: counts back to zero (migrate_out counts in place), and
        // the status words every rank will read from the header of the far message addressed to it: sticky error
        // bits, the owned count and the step they describe
        sendL[0] = 0;
        sendR[0] = 0;
        sendL[3] = step + 1u;
        sendR[3] = step + 1u;
        far_headers(far_send, world, far_next, dyn[DY_ERR], n_new, step + 1u, wanted, dyn[DY_HALO_NOW]);
        dyn[DY_HALO_NOW] = 0;
    }
}

// ---- the same in five launches, for steps that move many particles --------------------------------------------
// k_migrate_fill is ONE workgroup: right for the few thousand migrants per step it was written for, 2 ms per step (up to
// 9) once a collapsed cloud sloshes through four slabs with 10^5 .. 10^6 migrants per slab and step (round 4,
// profiles/r04/slab/).  The host knows the sizes of the step's messages; above a threshold it launches this sequence
// instead: count far arrivals -> plan (one thread) -> build the target / source lists with wave-aggregated global
// cursors -> apply -> finish (one thread).  Any pairing of sources and targets gives the same result (the sort is
// canonical), so the two forms are interchangeable.
__device__ __forceinline__ uint32_t wave_append(uint32_t *__restrict__ cursor, bool want)
{
    const unsigned long long m = __ballot(want);
    if (!want) return 0xFFFFFFFFu;
    const int lane = threadIdx.x & 63, leader = __ffsll((long long)m) - 1;
    uint32_t base = 0;
    if (lane == leader) base = atomicAdd(cursor, (uint32_t)__popcll(m));
    base = __shfl(base, leader, 64);
    return base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
}

__global__ void __launch_bounds__(WS_BLOCK) k_fill_count(uint32_t world, uint32_t me, uint32_t *__restrict__ dyn,
                                                         const uint32_t *__restrict__ recvL, const uint32_t *__restrict__ recvR,
                                                         const uint32_t *__restrict__ far_all, uint32_t far_cap,
                                                         uint32_t *__restrict__ status_ring, uint32_t status_slots)
{
    const uint32_t t = blockIdx.x * WS_BLOCK + threadIdx.x;
    const uint32_t step = dyn[DY_STEP];  // (k_fill_finish advances it)
    const uint32_t far_words = WS_HDR_WORDS + far_cap * WS_MIG_REC_WORDS;
    if (t < world) {
        uint32_t *status = status_ring + (size_t)(step % status_slots) * world * WS_HDR_WORDS;
        const uint32_t *msg = far_all + (size_t)t * far_words;
        for (uint32_t w = 0; w < WS_HDR_WORDS; w++) status[t * WS_HDR_WORDS + w] = msg[w];
        if (msg[3] != step) atomicOr(&dyn[DY_ERR], WS_DYN_ERR_STAMP);
    }
    if (t == 0 && ((me > 0 && recvL[3] != step) || (me + 1 < world && recvR[3] != step))) atomicOr(&dyn[DY_ERR], WS_DYN_ERR_STAMP);
    // far arrivals: every record of the far messages the other ranks addressed to this one
    if (t < world && t != me) atomicAdd(&dyn[DY_F_FAR], min(far_all[(size_t)t * far_words], far_cap));
}

__global__ void k_fill_plan(uint32_t world, uint32_t me, uint32_t cap, uint32_t *__restrict__ dyn, const uint32_t *__restrict__ recvL,
                            const uint32_t *__restrict__ recvR, uint32_t mig_cap, uint32_t hole_cap)
{
    const uint32_t nL = me > 0 ? min(recvL[0], mig_cap) : 0u, nR = me + 1 < world ? min(recvR[0], mig_cap) : 0u;
    const uint32_t leave = min(dyn[DY_NHOLE], hole_cap);
    if (dyn[DY_NHOLE] > hole_cap) atomicOr(&dyn[DY_ERR], WS_DYN_ERR_MIGRATION);
    const uint32_t n_old = dyn[DY_N];
    uint32_t arrivals = nL + nR + dyn[DY_F_FAR];
    uint32_t n_new = n_old - leave + arrivals;
    if (n_new > cap) {  // cannot hold them: drop the excess arrivals, flag the step
        atomicOr(&dyn[DY_ERR], WS_DYN_ERR_CAPACITY);
        arrivals -= n_new - cap;
        n_new = cap;
    }
    dyn[DY_F_NL] = nL;
    dyn[DY_F_NR] = nR;
    dyn[DY_F_LEAVE] = leave;
    dyn[DY_F_NOLD] = n_old;
    dyn[DY_F_NNEW] = n_new;
    dyn[DY_F_ARR] = arrivals;
    dyn[DY_F_NTGT] = 0;
    dyn[DY_F_NSRC] = 0;
}

// one thread per item of six segments laid end to end (their host-side upper bounds are the kernel's arguments):
// holes | new slots when growing | survivors above the new end | left arrivals | right arrivals | far records
__global__ void __launch_bounds__(WS_BLOCK) k_fill_lists(WsDev d, uint32_t world, uint32_t me, uint32_t *__restrict__ dyn,
                                                         const uint32_t *__restrict__ hole, const uint32_t *__restrict__ far_all,
                                                         uint32_t far_cap, uint32_t *__restrict__ tgt, uint32_t *__restrict__ src,
                                                         const uint32_t *__restrict__ cid_cur, uint32_t b_leave, uint32_t b_arr,
                                                         uint32_t b_mig)
{
    uint32_t t = blockIdx.x * WS_BLOCK + threadIdx.x;
    const uint32_t n_old = dyn[DY_F_NOLD], n_new = dyn[DY_F_NNEW], leave = dyn[DY_F_LEAVE], base = d.base;
    const uint32_t grow = n_new > n_old ? n_new - n_old : 0u, tail = n_old > n_new ? n_old - n_new : 0u;
    bool is_tgt = false, is_src = false;
    uint32_t value = 0;
    if (t < b_leave) {
        if (t < leave) {
            value = hole[t];
            is_tgt = value < base + n_new;
        }
    } else if ((t -= b_leave) < b_arr) {
        is_tgt = t < grow;
        value = base + n_old + t;
    } else if ((t -= b_arr) < b_leave) {
        if (t < tail) {
            value = base + n_new + t;
            is_src = cid_cur[value] != WS_DEAD;
        }
    } else if ((t -= b_leave) < b_mig) {
        is_src = t < dyn[DY_F_NL];
        value = (1u << 30) | t;
    } else if ((t -= b_mig) < b_mig) {
        is_src = t < dyn[DY_F_NR];
        value = (2u << 30) | t;
    } else if ((t -= b_mig) < world * far_cap) {
        const uint32_t far_words = WS_HDR_WORDS + far_cap * WS_MIG_REC_WORDS;
        const uint32_t q = t / far_cap, k = t % far_cap;
        const uint32_t *msg = far_all + (size_t)q * far_words;
        is_src = q != me && k < min(msg[0], far_cap);
        value = (3u << 30) | t;
    }
    const uint32_t at = wave_append(&dyn[DY_F_NTGT], is_tgt);
    if (is_tgt) tgt[at] = value;
    const uint32_t as = wave_append(&dyn[DY_F_NSRC], is_src);
    if (is_src) src[as] = value;
}

__global__ void __launch_bounds__(WS_BLOCK) k_fill_apply(WsDev d, const uint32_t *__restrict__ dyn, const uint32_t *__restrict__ recvL,
                                                         const uint32_t *__restrict__ recvR, const uint32_t *__restrict__ far_all,
                                                         uint32_t far_cap, const uint32_t *__restrict__ tgt,
                                                         const uint32_t *__restrict__ src, WsSoA cur, uint32_t *__restrict__ cid_cur,
                                                         uint32_t *__restrict__ count)
{
    const uint32_t t = blockIdx.x * WS_BLOCK + threadIdx.x;
    if (t >= min(dyn[DY_F_NTGT], dyn[DY_F_NSRC])) return;  // (after a capacity overrun there are more sources: the excess is dropped)
    const uint32_t far_words = WS_HDR_WORDS + far_cap * WS_MIG_REC_WORDS;
    const uint32_t to = tgt[t], from = src[t];
    const uint32_t tag = from >> 30, idx = from & 0x3FFFFFFFu;
    if (tag) {
        const uint32_t *msg = tag == 1u ? recvL : tag == 2u ? recvR : far_all + (size_t)(idx / far_cap) * far_words;
        const float4 *rec = reinterpret_cast<const float4 *>(msg + WS_HDR_WORDS) + 2 * (size_t)(tag == 3u ? idx % far_cap : idx);
        const float4 p = rec[0], v = rec[1];
        const float4 q = make_float4(p.x + v.x * WS_LOOKAHEAD, p.y + v.y * WS_LOOKAHEAD, p.z + v.z * WS_LOOKAHEAD, 0.f);
        const uint32_t c = grid_cell(d, q.x, q.y, q.z);
        cur.pos(to) = p;
        cur.vel(to) = make_float4(v.x, v.y, v.z, __uint_as_float(c));
        cur.pred[to] = q;
        cid_cur[to] = c;
        cur.rank[to] = atomicAdd(&count[c], 1u);  // the next rank of its cell (k_place)
    } else {
        cur.pos(to) = cur.pos(idx);
        cur.vel(to) = cur.vel(idx);
        cur.pred[to] = cur.pred[idx];
        cid_cur[to] = cid_cur[idx];
        cur.rank[to] = cur.rank[idx];
    }
}

__global__ void k_fill_finish(uint32_t world, uint32_t *__restrict__ dyn, uint32_t *__restrict__ sendL, uint32_t *__restrict__ sendR,
                              uint32_t *__restrict__ far_send, uint32_t far_next)
{
    const uint32_t step = dyn[DY_STEP];
    dyn[DY_N] = dyn[DY_F_NNEW];
    dyn[DY_STEP] = step + 1u;
    dyn[DY_NHOLE] = 0;
    dyn[DY_ARRIVED] += dyn[DY_F_ARR];
    dyn[DY_LEFT] += dyn[DY_F_LEAVE];
    dyn[DY_F_FAR] = 0;
    const uint32_t wanted = max(sendL[0], sendR[0]);
    dyn[DY_PEAK_MIG] = max(dyn[DY_PEAK_MIG], wanted);
    sendL[0] = 0;
    sendR[0] = 0;
    sendL[3] = step + 1u;
    sendR[3] = step + 1u;
    far_headers(far_send, world, far_next, dyn[DY_ERR], dyn[DY_F_NNEW], step + 1u, wanted, dyn[DY_HALO_NOW]);
    dyn[DY_HALO_NOW] = 0;
}

void wsk_migrate_fill(hipStream_t s, const WsDev &d, uint32_t world, uint32_t me, uint32_t cap, uint32_t *dyn,
                      const uint32_t *hole, const uint32_t *recvL, const uint32_t *recvR, uint32_t mig_cap,
                      const uint32_t *far_all, uint32_t far_cap, uint32_t *tgt, uint32_t *src, WsSoA cur, uint32_t *cid_cur,
                      uint32_t *count, uint32_t *status_ring, uint32_t status_slots, uint32_t hole_cap, uint32_t *sendL,
                      uint32_t *sendR, uint32_t *far_send, uint32_t far_next, uint64_t leave_bound)
{
    // (mig_cap / far_cap: the records the RECEIVED messages carry -- the host's current limits, not the buffers' capacities;
    // far_next: the records each far message of the NEXT exchange will carry, the stride the new headers are written at;
    // leave_bound: how many particles can have left -- what the messages this rank SENT carry; 0 = as many as it received)
    const uint64_t volume = 2ull * mig_cap + (uint64_t)world * far_cap;
    if (volume <= 49152ull) {
        hipLaunchKernelGGL(k_migrate_fill, dim3(1), dim3(WS_FILL_THREADS), 0, s, d, world, me, cap, dyn, hole, recvL, recvR,
                           mig_cap, far_all, far_cap, tgt, src, cur, cid_cur, count, status_ring, status_slots, hole_cap, sendL,
                           sendR, far_send, far_next);
        return;
    }
    const uint32_t b_leave = (uint32_t)std::min<uint64_t>(hole_cap, leave_bound ? leave_bound : volume);
    const uint32_t b_arr = (uint32_t)std::min<uint64_t>(volume, 0x3FFFFFFFull);
    hipLaunchKernelGGL(k_fill_count, dim3(cdiv(world, WS_BLOCK)), dim3(WS_BLOCK), 0, s, world, me, dyn, recvL, recvR, far_all, far_cap,
                       status_ring, status_slots);
    hipLaunchKernelGGL(k_fill_plan, dim3(1), dim3(1), 0, s, world, me, cap, dyn, recvL, recvR, mig_cap, hole_cap);
    const uint64_t items = 2ull * b_leave + b_arr + 2ull * mig_cap + (uint64_t)world * far_cap;
    hipLaunchKernelGGL(k_fill_lists, dim3((uint32_t)((items + WS_BLOCK - 1) / WS_BLOCK)), dim3(WS_BLOCK), 0, s, d, world, me, dyn, hole,
                       far_all, far_cap, tgt, src, cid_cur, b_leave, b_arr, mig_cap);
    hipLaunchKernelGGL(k_fill_apply, dim3(cdiv((uint32_t)std::min<uint64_t>((uint64_t)b_leave + b_arr, 0x7FFFFFFFull), WS_BLOCK)),
                       dim3(WS_BLOCK), 0, s, d, dyn, recvL, recvR, far_all, far_cap, tgt, src, cur, cid_cur, count);
    hipLaunchKernelGGL(k_fill_finish, dim3(1), dim3(1), 0, s, world, dyn, sendL, sendR, far_send, far_next);
}

// Halo messages.  A: [header | cell-start slice of the boundary layer (rowy + 1 words, padded to 4) | its 32-byte
// {pred, vel} records]; B: the same particles' (density, near density), 8 bytes each.  The boundary layers are
// contiguous ranges of the sorted order: layer 1 goes left, layer nxl - 2 goes right.
__device__ __forceinline__ uint32_t halo_slice_words(uint32_t rowy) { return (rowy + 1u + 3u) & ~3u; }

__global__ void __launch_bounds__(WS_BLOCK) k_halo_pack(WsDev d, const uint32_t *__restrict__ start, WsSorted srt,
                                                        uint32_t *__restrict__ dyn, uint32_t rowy, uint32_t halo_cap,
                                                        uint32_t *__restrict__ sendL, uint32_t *__restrict__ sendR,
                                                        int densities)
{
    const uint32_t t = blockIdx.x * WS_BLOCK + threadIdx.x;
    const uint32_t side = blockIdx.y;  // 0: the left-going layer (1), 1: the right-going layer (nxl - 2)
    // no neighbour on that side, no message: the layer at the container's wall is not a boundary layer (round 4: it was
    // packed and COUNTED all the same -- particles pile up against the wall, and an end slab failed with "a boundary
    // layer holds more particles than a halo message" for a message nobody sends)
    if (side ? !d.has_right : !d.has_left) return;
    uint32_t *msg = side ? sendR : sendL;
    const uint32_t first = side ? d.lidx[2] : d.lidx[0];  // index of the layer's first cell start
    const uint32_t s0 = start[first], cnt_all = start[first + rowy] - s0;
    const uint32_t cnt = min(cnt_all, halo_cap);
    if (densities) {
        if (t < cnt) reinterpret_cast<float2 *>(msg)[t] = make_float2(srt.pred(s0 + t).w, srt.vel(s0 + t).w);
        return;
    }
    if (t == 0) {
        if (cnt_all > halo_cap) atomicOr(&dyn[DY_ERR], WS_DYN_ERR_HALO);
        atomicMax(&dyn[DY_PEAK_HALO], cnt_all);
        atomicMax(&dyn[DY_HALO_NOW], cnt_all);
        msg[0] = cnt;
        msg[1] = dyn[DY_ERR];
        msg[2] = dyn[DY_N];
        msg[3] = dyn[DY_STEP];  // this step's migration has run on both ends: the same number on both
    }
    if (t <= rowy) msg[WS_HDR_WORDS + t] = start[first + t];
    if (t < cnt) {
        float4 *rec = reinterpret_cast<float4 *>(msg + WS_HDR_WORDS + halo_slice_words(rowy)) + 2 * (size_t)t;
        rec[0] = srt.pred(s0 + t);
        rec[1] = srt.vel(s0 + t);
    }
}

void wsk_halo_pack(hipStream_t s, const WsDev &d, const uint32_t *start, WsSorted srt, uint32_t *dyn, uint32_t rowy,
                   uint32_t halo_cap, uint32_t *sendL, uint32_t *sendR, bool densities)
{
    const uint32_t work = densities ? halo_cap : max(halo_cap, rowy + 1u);
    hipLaunchKernelGGL(k_halo_pack, dim3(cdiv(work, WS_BLOCK), 2), dim3(WS_BLOCK), 0, s, d, start, srt, dyn, rowy, halo_cap,
                       sendL, sendR, densities ? 1 : 0);
}

// After halo A: the ghost records into [base - gL, base) and [base + n, base + n + gR), their planar copy for K4,
// and the cell starts of the two ghost layers (the neighbours' start slices rebased onto this slab's ghost slots)
// plus the guard entries in front of / behind the table.
//   layer 0      <- left neighbour's last owned layer      layer nxl-1  <- right neighbour's first owned layer
// After halo B (densities != 0): only the w lanes of the ghost records.
__global__ void __launch_bounds__(WS_BLOCK) k_halo_unpack(WsDev d, uint32_t *__restrict__ start, WsSorted srt, WsXYZ sxyz,
                                                          uint32_t *__restrict__ dyn, uint32_t rowy, uint32_t nxl,
                                                          uint32_t ghost_cap, const uint32_t *__restrict__ recvL,
                                                          const uint32_t *__restrict__ recvR, int densities)
{
    const uint32_t t = blockIdx.x * WS_BLOCK + threadIdx.x;
    const uint32_t side = blockIdx.y;
    const bool have = side ? d.has_right != 0u : d.has_left != 0u;
    const uint32_t *msg = side ? recvR : recvL;
    const uint32_t n = dyn[DY_N], base = d.base, guard = (uint32_t)d.guard;
    if (densities) {
        const uint32_t g = side ? dyn[DY_GR] : dyn[DY_GL];
        if (t < g) {
            const uint32_t slot = side ? base + n + t : base - g + t;
            const float2 rho = reinterpret_cast<const float2 *>(msg)[t];
            srt.pred(slot).w = rho.x;
            srt.vel(slot).w = rho.y;
        }
        return;
    }
    uint32_t g = have ? msg[0] : 0u;
    if (g > ghost_cap) {
        if (t == 0) atomicOr(&dyn[DY_ERR], WS_DYN_ERR_GHOSTS);
        g = ghost_cap;
    }
    if (t == 0 && have && msg[3] != dyn[DY_STEP]) atomicOr(&dyn[DY_ERR], WS_DYN_ERR_STAMP);  // another step's halo (see k_migrate_fill)
    if (t == 0) dyn[side ? DY_GR : DY_GL] = g;
    const uint32_t *slice = msg + WS_HDR_WORDS;
    const uint32_t first_slot = side ? base + n : base - g;
    if (t < g) {
        const float4 *rec = reinterpret_cast<const float4 *>(msg + WS_HDR_WORDS + halo_slice_words(rowy)) + 2 * (size_t)t;
        const float4 q = rec[0];
        const uint32_t slot = first_slot + t;
        srt.pred(slot) = q;
        srt.vel(slot) = rec[1];
        sxyz.x[slot] = q.x;
        sxyz.y[slot] = q.y;
        sxyz.z[slot] = q.z;
    }
    if (side == 0) {
        // guard entries + layer 0: everything in front of the owned range
        if (t < guard + rowy) {
            uint32_t v = base - g;
            if (t >= guard) v = g ? base - g + min(slice[t - guard] - slice[0], g) : base;
            start[t] = v;
        }
    } else {
        // layer nxl-1 + end sentinel + guard entries: everything behind the owned range
        if (t < rowy + guard + 2u) {
            uint32_t v = base + n + g;
            if (t < rowy) v = g ? base + n + min(slice[t] - slice[0], g) : base + n;
            start[guard + (nxl - 1u) * rowy + t] = v;
        }
    }
}

void wsk_halo_unpack(hipStream_t s, const WsDev &d, uint32_t *start, WsSorted srt, WsXYZ sxyz, uint32_t *dyn, uint32_t rowy,
                     uint32_t nxl, uint32_t ghost_cap, const uint32_t *recvL, const uint32_t *recvR, bool densities)
{
    const uint32_t work = densities ? ghost_cap : max(ghost_cap, rowy + (uint32_t)d.guard + 2u);
    hipLaunchKernelGGL(k_halo_unpack, dim3(cdiv(work, WS_BLOCK), 2), dim3(WS_BLOCK), 0, s, d, start, srt, sxyz, dyn, rowy, nxl,
                       ghost_cap, recvL, recvR, densities ? 1 : 0);
}

// Slab readback: owned particles (state of the last step, sorted order) with their ids.
__global__ void __launch_bounds__(WS_BLOCK) k_gather_slab(WsDev d, WsSoA cur, WsSorted srt, const float4 *__restrict__ accel,
                                                          int have_step, ws_particle80 *__restrict__ out,
                                                          uint32_t *__restrict__ ids)
{
    const uint32_t k = blockIdx.x * WS_BLOCK + threadIdx.x;
    if (k >= ws_n(d)) return;
    const uint32_t i = d.base + k;
    const float4 p = cur.pos(i), v = cur.vel(i), q = cur.pred[i];
    float4 dp = make_float4(0.f, 0.f, 0.f, 0.f), a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (have_step) {
        dp.x = srt.pred(i).w;
        dp.y = srt.vel(i).w;
        dp.z = d.pressure_scalar * (dp.x - d.target_density);
        dp.w = d.near_pressure_scalar * dp.y;
        a = accel[i];
    }
    float4 *rec = reinterpret_cast<float4 *>(out + k);
    rec[0] = make_float4(p.x, p.y, p.z, 0.f);
    rec[1] = dp;
    rec[2] = make_float4(v.x, v.y, v.z, 0.f);
    rec[3] = make_float4(a.x, a.y, a.z, 0.f);
    rec[4] = make_float4(q.x, q.y, q.z, 0.f);
    ids[k] = __float_as_uint(p.w);
}

void wsk_gather_slab(hipStream_t s, const WsDev &d, WsSoA cur, WsSorted srt, const float4 *accel, bool have_step,
                     ws_particle80 *out, uint32_t *ids)
{
    if (d.n == 0) return;
    hipLaunchKernelGGL(k_gather_slab, dim3(cdiv(d.n, WS_BLOCK)), dim3(WS_BLOCK), 0, s, d, cur, srt, accel,
                       have_step ? 1 : 0, out, ids);
}

// upload of a slab's initial particles: ids are given explicitly
__global__ void __launch_bounds__(WS_BLOCK) k_upload_positions_ids(const float *__restrict__ xyz,
                                                                   const uint32_t *__restrict__ ids, WsSoA cur, uint32_t n)
{
    const uint32_t i = blockIdx.x * WS_BLOCK + threadIdx.x;
    if (i >= n) return;
    const float x = xyz[3 * (size_t)i], y = xyz[3 * (size_t)i + 1], z = xyz[3 * (size_t)i + 2];
    cur.pos(i) = make_float4(x, y, z, __uint_as_float(ids[i]));
    cur.vel(i) = make_float4(0.f, 0.f, 0.f, 0.f);
    cur.pred[i] = make_float4(x, y, z, 0.f);
}

void wsk_upload_positions_ids(hipStream_t s, const float *xyz_dev, const uint32_t *ids_dev, WsSoA cur, uint32_t n)
{
    if (!n) return;
    hipLaunchKernelGGL(k_upload_positions_ids, dim3(cdiv(n, WS_BLOCK)), dim3(WS_BLOCK), 0, s, xyz_dev, ids_dev, cur, n);
}

// ---------------------------------------------------------------------------------
// slab handles: the id-ordered GLOBAL views and loads of the C ABI (ws_read_positions / ws_read_speeds /
// ws_read_particles / ws_reset / ws_write_particles / a re-grid on ws_set_params), which the reference's host uses every
// frame (update(), src/fluid_compute.rs:478-485) and on Space (despawn_liquid, :505-525).  Reads: every slab packs
// {id, payload} records of the particles it owns, the records are all-gathered, and every rank scatters them by id.
// Loads: every rank is handed the same global array and keeps the particles whose x layer falls into its cuts.
// ---------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t slab_owner(const WsDev &d, const uint32_t *__restrict__ cuts, uint32_t world, float x)
{
    const uint32_t gxg = (uint32_t)grid_layer_x(d, x);
    uint32_t dest = 0;
    while (dest + 1 < world && gxg >= cuts[dest + 1]) dest++;
    return dest;
}

// payload of one gathered record (after its id word)
enum { WS_PACK_POS = 0, WS_PACK_SPEED = 1, WS_PACK_RECORD = 2, WS_PACK_STATE = 3, WS_PACK_KEY = 4 };
__host__ __device__ constexpr uint32_t ws_pack_words(int kind)
{
    return kind == WS_PACK_POS ? 3u : kind == WS_PACK_SPEED ? 1u : kind == WS_PACK_RECORD ? 20u : kind == WS_PACK_STATE ? 9u : 1u;
}
uint32_t wsk_pack_words(int kind) { return ws_pack_words(kind); }

template <int KIND>
__global__ void __launch_bounds__(WS_BLOCK) k_slab_pack(WsDev d, WsSoA cur, WsSorted srt, const float4 *__restrict__ accel,
                                                        int have_step, uint32_t *__restrict__ out)
{
    const uint32_t k = blockIdx.x * WS_BLOCK + threadIdx.x;
    if (k >= d.n) return;
    const uint32_t i = d.base + k;
    constexpr uint32_t W = ws_pack_words(KIND) + 1u;
    uint32_t *rec = out + (size_t)k * W;
    float *f = reinterpret_cast<float *>(rec + 1);
    if constexpr (KIND == WS_PACK_KEY) {
        // the reference's particle_cell_indicies entry: hash of the cell of the predicted position the last step started from
        const float4 q = srt.pred(i);
        rec[0] = __float_as_uint(srt.pos[i].w);
        rec[1] = ref_hash_key(d, q.x, q.y, q.z);
        return;
    }
    const float4 p = cur.pos(i);
    rec[0] = __float_as_uint(p.w);
    if constexpr (KIND == WS_PACK_POS) {
        f[0] = p.x; f[1] = p.y; f[2] = p.z;
    } else if constexpr (KIND == WS_PACK_SPEED) {
        const float4 v = cur.vel(i);
        f[0] = sqrtf(v.x * v.x + v.y * v.y + v.z * v.z);
    } else if constexpr (KIND == WS_PACK_STATE) {
        const float4 v = cur.vel(i), q = cur.pred[i];
        f[0] = p.x; f[1] = p.y; f[2] = p.z;
        f[3] = v.x; f[4] = v.y; f[5] = v.z;
        f[6] = q.x; f[7] = q.y; f[8] = q.z;
    } else {  // the 80-byte record (k_gather_slab's fields)
        const float4 v = cur.vel(i), q = cur.pred[i];
        float4 dp = make_float4(0.f, 0.f, 0.f, 0.f), a = make_float4(0.f, 0.f, 0.f, 0.f);
        if (have_step) {
            dp.x = srt.pred(i).w;
            dp.y = srt.vel(i).w;
            dp.z = d.pressure_scalar * (dp.x - d.target_density);
            dp.w = d.near_pressure_scalar * dp.y;
            a = accel[i];
        }
        f[0] = p.x; f[1] = p.y; f[2] = p.z; f[3] = 0.f;
        f[4] = dp.x; f[5] = dp.y; f[6] = dp.z; f[7] = dp.w;
        f[8] = v.x; f[9] = v.y; f[10] = v.z; f[11] = 0.f;
        f[12] = a.x; f[13] = a.y; f[14] = a.z; f[15] = 0.f;
        f[16] = q.x; f[17] = q.y; f[18] = q.z; f[19] = 0.f;
    }
}

void wsk_slab_pack(hipStream_t s, const WsDev &d, int kind, WsSoA cur, WsSorted srt, const float4 *accel, bool have_step,
                   uint32_t *out)
{
    if (!d.n) return;
    const dim3 g(cdiv(d.n, WS_BLOCK)), b(WS_BLOCK);
    const int hs = have_step ? 1 : 0;
    switch (kind) {
        case WS_PACK_POS: hipLaunchKernelGGL(k_slab_pack<WS_PACK_POS>, g, b, 0, s, d, cur, srt, accel, hs, out); break;
        case WS_PACK_SPEED: hipLaunchKernelGGL(k_slab_pack<WS_PACK_SPEED>, g, b, 0, s, d, cur, srt, accel, hs, out); break;
        case WS_PACK_RECORD: hipLaunchKernelGGL(k_slab_pack<WS_PACK_RECORD>, g, b, 0, s, d, cur, srt, accel, hs, out); break;
        case WS_PACK_STATE: hipLaunchKernelGGL(k_slab_pack<WS_PACK_STATE>, g, b, 0, s, d, cur, srt, accel, hs, out); break;
        default: hipLaunchKernelGGL(k_slab_pack<WS_PACK_KEY>, g, b, 0, s, d, cur, srt, accel, hs, out); break;
    }
}

// every rank's records (rank r: cnt[4 r] of them at all + r * stride_words) scattered by id: out[id * pw + j]
__global__ void __launch_bounds__(WS_BLOCK) k_slab_unpack_by_id(const uint32_t *__restrict__ all, const uint32_t *__restrict__ cnt,
                                                                uint32_t max_n, size_t stride_words, uint32_t pw,
                                                                uint32_t n_global, uint32_t *__restrict__ out)
{
    const uint32_t k = blockIdx.x * WS_BLOCK + threadIdx.x, r = blockIdx.y;
    if (k >= min(cnt[4 * r], max_n)) return;
    const uint32_t *rec = all + (size_t)r * stride_words + (size_t)k * (pw + 1u);
    const uint32_t id = rec[0];
    if (id >= n_global) return;
    for (uint32_t j = 0; j < pw; j++) out[(size_t)id * pw + j] = rec[1 + j];
}

void wsk_slab_unpack_by_id(hipStream_t s, const uint32_t *all, const uint32_t *cnt, uint32_t world, uint32_t max_n,
                           size_t stride_words, uint32_t pw, uint32_t n_global, uint32_t *out)
{
    if (!max_n) return;
    hipLaunchKernelGGL(k_slab_unpack_by_id, dim3(cdiv(max_n, WS_BLOCK), world), dim3(WS_BLOCK), 0, s, all, cnt, max_n,
                       stride_words, pw, n_global, out);
}

// x-layer histogram of ALL particles (gathered {id, pos, vel, pred} records of every rank): the input of a re-cut
// (ws_slab_rebalance).  Every rank computes the same histogram from the same gathered data, hence the same cuts.
__global__ void __launch_bounds__(WS_BLOCK) k_slab_layer_hist(WsDev d, const uint32_t *__restrict__ all, const uint32_t *__restrict__ cnt,
                                                              uint32_t max_n, size_t stride_words, uint32_t *__restrict__ hist)
{
    const uint32_t t = blockIdx.x * WS_BLOCK + threadIdx.x, r = blockIdx.y;
    if (t >= min(cnt[4 * r], max_n)) return;
    const float *f = reinterpret_cast<const float *>(all + (size_t)r * stride_words + (size_t)t * 10u + 1u);
    atomicAdd(&hist[grid_layer_x(d, f[6])], 1u);  // by PREDICTED x, what the step bins by
}

void wsk_slab_layer_hist(hipStream_t s, const WsDev &d, const uint32_t *all, const uint32_t *cnt, uint32_t world, uint32_t max_n,
                         size_t stride_words, uint32_t *hist)
{
    if (!max_n) return;
    hipLaunchKernelGGL(k_slab_layer_hist, dim3(cdiv(max_n, WS_BLOCK), world), dim3(WS_BLOCK), 0, s, d, all, cnt, max_n, stride_words,
                       hist);
}

// Loads.  One lane per record of a chunk of a GLOBAL array; the lanes whose particle this slab owns append it to the
// owned range (one atomic per wave; the order is free: the sort is canonical).  dyn[DY_N] is the fill counter.
//   SRC 0: positions at rest (float xyz; id = id0 + t)            -- ws_reset / FluidParticle::make_vec_from_positions
//   SRC 1: 80-byte records (id = id0 + t)                          -- ws_write_particles
//   SRC 2: gathered {id, pos, vel, pred} records of every rank     -- a re-grid (blockIdx.y = source rank)
template <int SRC>
__global__ void __launch_bounds__(WS_BLOCK) k_slab_select(WsDev d, const uint32_t *__restrict__ cuts, uint32_t world,
                                                          uint32_t me, const void *__restrict__ chunk, uint32_t id0, uint32_t m,
                                                          const uint32_t *__restrict__ cnt, size_t stride_words, WsSoA cur,
                                                          uint32_t cap, uint32_t *__restrict__ dyn)
{
    const uint32_t t = blockIdx.x * WS_BLOCK + threadIdx.x;
    bool mine = false;
    float4 p = make_float4(0.f, 0.f, 0.f, 0.f), v = p, q = p;
    uint32_t id = 0;
    if constexpr (SRC == 2) {
        const uint32_t r = blockIdx.y;
        if (t < min(cnt[4 * r], m)) {
            const uint32_t *rec = static_cast<const uint32_t *>(chunk) + (size_t)r * stride_words + (size_t)t * 10u;
            const float *f = reinterpret_cast<const float *>(rec + 1);
            id = rec[0];
            p = make_float4(f[0], f[1], f[2], 0.f);
            v = make_float4(f[3], f[4], f[5], 0.f);
            q = make_float4(f[6], f[7], f[8], 0.f);
            mine = true;
        }
    } else if (t < m) {
        id = id0 + t;
        if constexpr (SRC == 0) {
            const float *f = static_cast<const float *>(chunk) + 3 * (size_t)t;
            p = q = make_float4(f[0], f[1], f[2], 0.f);
        } else {
            const float4 *rec = reinterpret_cast<const float4 *>(static_cast<const ws_particle80 *>(chunk) + t);
            p = rec[0]; v = rec[2]; q = rec[4];
        }
        mine = true;
    }
    // ownership follows the PREDICTED position: it is what the step bins by (simulation.wgsl:139)
    mine = mine && slab_owner(d, cuts, world, q.x) == me;
    const unsigned long long sel = __ballot(mine);
    if (!sel) return;
    const int lane = threadIdx.x & 63, leader = __ffsll((long long)sel) - 1;
    uint32_t first = 0;
    if (lane == leader) first = atomicAdd(&dyn[DY_N], (uint32_t)__popcll(sel));
    first = __shfl(first, leader, 64);
    if (!mine) return;
    const uint32_t slot = first + (uint32_t)__popcll(sel & ((1ull << lane) - 1ull));
    if (slot >= cap) {  // counted, not stored: the host sees DY_N > cap and fails the load
        return;
    }
    const uint32_t i = d.base + slot;
    cur.pos(i) = make_float4(p.x, p.y, p.z, __uint_as_float(id));
    cur.vel(i) = make_float4(v.x, v.y, v.z, 0.f);
    cur.pred[i] = make_float4(q.x, q.y, q.z, 0.f);
}

void wsk_slab_select(hipStream_t s, const WsDev &d, int src, const uint32_t *cuts, uint32_t world, uint32_t me,
                     const void *chunk, uint32_t id0, uint32_t m, const uint32_t *cnt, size_t stride_words, WsSoA cur,
                     uint32_t cap, uint32_t *dyn)
{
    if (!m) return;
    const dim3 b(WS_BLOCK);
    if (src == 0)
        hipLaunchKernelGGL(k_slab_select<0>, dim3(cdiv(m, WS_BLOCK)), b, 0, s, d, cuts, world, me, chunk, id0, m, cnt, stride_words, cur, cap, dyn);
    else if (src == 1)
        hipLaunchKernelGGL(k_slab_select<1>, dim3(cdiv(m, WS_BLOCK)), b, 0, s, d, cuts, world, me, chunk, id0, m, cnt, stride_words, cur, cap, dyn);
    else
        hipLaunchKernelGGL(k_slab_select<2>, dim3(cdiv(m, WS_BLOCK), world), b, 0, s, d, cuts, world, me, chunk, id0, m, cnt, stride_words, cur, cap, dyn);
}

// ---------------------------------------------------------------------------------
// The reference-order validation kernels (a literal HIP restatement of the six WGSL entry points, used only to
// cross-check the CPU restatement bit for bit) are NOT part of the product: they live in tests/refcheck/ and are
// compiled only into the test-only build (tests/libwsfluid_refcheck.so, -DWS_WITH_REFCHECK -Itests/refcheck).
// ---------------------------------------------------------------------------------
#ifdef WS_WITH_REFCHECK
#include "ws_refcheck.inc"
#endif
