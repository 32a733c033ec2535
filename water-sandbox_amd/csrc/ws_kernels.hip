// ws_kernels.hip -- gfx950 (CDNA4) kernels of the SPH fluid step.
//
// Written for MI355X only: 64-lane wavefronts, SoA particle arrays in HBM, dense
// z-fastest cell grid so that the three z-neighbour cells of any (dx,dy) column are
// ONE contiguous particle run (9 runs per particle instead of 27 bucket walks).
// No MFMA: the step is gather/scan/sort work bounded by HBM/L2/LDS and VALU.
//
// Arithmetic contract: every float expression below is written in the evaluation
// order of the reference WGSL and this file is compiled with -ffp-contract=off, so
// each operation is one IEEE binary32 operation (hipcc's default f32 divide and sqrt
// are correctly rounded).  What differs from the reference is only the ORDER in which
// neighbours are visited (dense-grid order instead of hashed-bucket order).
#include "ws_internal.h"

#pragma clang fp contract(off)

#define WS_BLOCK 256
#define WS_SCAN_ITEMS 8
#define WS_SCAN_TILE (WS_BLOCK * WS_SCAN_ITEMS)

static inline uint32_t cdiv(uint32_t a, uint32_t b) { return (a + b - 1) / b; }

// ---------------------------------------------------------------------------------
// cell helpers
// ---------------------------------------------------------------------------------

// get_cell, assets/simulation.wgsl:121-123: vec3<i32>(floor(position / h)) with an
// IEEE divide (not a reciprocal multiply), then clamped into the padded dense grid.
// Clamping is monotone per axis, so two particles whose true cells are adjacent stay
// in adjacent-or-equal grid cells: the 27-cell search over grid cells visits a superset
// of the reference's candidates and the exact distance test decides, as it does there.
__device__ __forceinline__ uint32_t grid_cell(const WsDev &d, float x, float y, float z)
{
    const float fx = floorf(x / d.h) - (float)d.org[0];
    const float fy = floorf(y / d.h) - (float)d.org[1];
    const float fz = floorf(z / d.h) - (float)d.org[2];
    // fmaxf/fminf also squash NaN to the low border
    const int gx = (int)fminf(fmaxf(fx, 0.0f), (float)(d.dim[0] - 1));
    const int gy = (int)fminf(fmaxf(fy, 0.0f), (float)(d.dim[1] - 1));
    const int gz = (int)fminf(fmaxf(fz, 0.0f), (float)(d.dim[2] - 1));
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
// uploads
// ---------------------------------------------------------------------------------

// FluidParticle::make_vec_from_positions, src/fluid_compute.rs:118-130.
__global__ void __launch_bounds__(WS_BLOCK) k_upload_positions(const float *__restrict__ xyz, WsSoA cur,
                                                               uint32_t n)
{
    const uint32_t i = blockIdx.x * WS_BLOCK + threadIdx.x;
    if (i >= n) return;
    const float x = xyz[3 * (size_t)i], y = xyz[3 * (size_t)i + 1], z = xyz[3 * (size_t)i + 2];
    cur.pos[i] = make_float4(x, y, z, __uint_as_float(i));
    cur.vel[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    cur.pred[i] = make_float4(x, y, z, 0.f);
}

__global__ void __launch_bounds__(WS_BLOCK) k_upload_particles(const ws_particle80 *__restrict__ in, WsSoA cur,
                                                               uint32_t n)
{
    const uint32_t i = blockIdx.x * WS_BLOCK + threadIdx.x;
    if (i >= n) return;
    const float4 *rec = reinterpret_cast<const float4 *>(in + i);
    const float4 p = rec[0], v = rec[2], q = rec[4];
    cur.pos[i] = make_float4(p.x, p.y, p.z, __uint_as_float(i));
    cur.vel[i] = make_float4(v.x, v.y, v.z, 0.f);
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
__global__ void __launch_bounds__(WS_BLOCK) k_bin(WsDev d, const float4 *__restrict__ pred,
                                                  uint32_t *__restrict__ cid, uint32_t *__restrict__ count)
{
    const uint32_t i = blockIdx.x * WS_BLOCK + threadIdx.x;
    if (i >= d.n) return;
    const float4 p = pred[i];
    const uint32_t c = grid_cell(d, p.x, p.y, p.z);
    cid[i] = c;
    atomicAdd(&count[c], 1u);
}

void wsk_bin(hipStream_t s, const WsDev &d, const float4 *pred, uint32_t *cid, uint32_t *count)
{
    hipLaunchKernelGGL(k_bin, dim3(cdiv(d.n, WS_BLOCK)), dim3(WS_BLOCK), 0, s, d, pred, cid, count);
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

// exclusive prefix of `v` over the 256-thread block; *total = block sum
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t *total)
{
    __shared__ uint32_t wsum[WS_BLOCK / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t incl = wave_incl_scan(v);
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < WS_BLOCK / 64; w++) {
        const uint32_t s = wsum[w];
        if (w < wave) base += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return base + incl - v;
}

__device__ __forceinline__ void load_tile(const uint32_t *__restrict__ src, uint32_t nitems, uint32_t v[WS_SCAN_ITEMS])
{
    const uint32_t base = blockIdx.x * WS_SCAN_TILE + threadIdx.x * WS_SCAN_ITEMS;
    if (base + WS_SCAN_ITEMS <= nitems) {
        const uint4 a = *reinterpret_cast<const uint4 *>(src + base);
        const uint4 b = *reinterpret_cast<const uint4 *>(src + base + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
        v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
#pragma unroll
        for (int k = 0; k < WS_SCAN_ITEMS; k++) v[k] = (base + k < nitems) ? src[base + k] : 0u;
    }
}

__global__ void __launch_bounds__(WS_BLOCK) k_scan_reduce(const uint32_t *__restrict__ count, uint32_t nitems,
                                                          uint32_t *__restrict__ bsum)
{
    uint32_t v[WS_SCAN_ITEMS];
    load_tile(count, nitems, v);
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < WS_SCAN_ITEMS; k++) s += v[k];
    uint32_t total;
    block_excl_scan(s, &total);
    if (threadIdx.x == 0) bsum[blockIdx.x] = total;
}

// single block: in-place exclusive scan of the block sums
__global__ void __launch_bounds__(WS_BLOCK) k_scan_top(uint32_t *__restrict__ bsum, uint32_t nblocks)
{
    uint32_t carry = 0;
    for (uint32_t base = 0; base < nblocks; base += WS_BLOCK) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = (i < nblocks) ? bsum[i] : 0u;
        uint32_t total;
        const uint32_t ex = block_excl_scan(v, &total);
        if (i < nblocks) bsum[i] = carry + ex;
        carry += total;
    }
}

template <bool ZERO>
__global__ void __launch_bounds__(WS_BLOCK) k_scan_apply(uint32_t *__restrict__ count, uint32_t nitems,
                                                         const uint32_t *__restrict__ bsum,
                                                         uint32_t *__restrict__ start, uint32_t *__restrict__ cursor)
{
    uint32_t v[WS_SCAN_ITEMS];
    load_tile(count, nitems, v);
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < WS_SCAN_ITEMS; k++) s += v[k];
    uint32_t total;
    uint32_t run = block_excl_scan(s, &total) + bsum[blockIdx.x];
    const uint32_t base = blockIdx.x * WS_SCAN_TILE + threadIdx.x * WS_SCAN_ITEMS;
#pragma unroll
    for (int k = 0; k < WS_SCAN_ITEMS; k++) {
        if (base + k < nitems) {
            start[base + k] = run;
            cursor[base + k] = run;
            if (ZERO) count[base + k] = 0u;
        }
        run += v[k];
    }
}

uint32_t wsk_scan_blocks(uint32_t nitems) { return cdiv(nitems, WS_SCAN_TILE); }

// start_body points at the first real entry (after the guard); entry [nitems] and the
// guards are constant and written once by the host.
void wsk_scan(hipStream_t s, uint32_t *count, uint32_t *start_body, uint32_t *cursor, uint32_t *bsum,
              uint32_t nitems, uint32_t nblocks, bool zero_count)
{
    hipLaunchKernelGGL(k_scan_reduce, dim3(nblocks), dim3(WS_BLOCK), 0, s, count, nitems, bsum);
    hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(WS_BLOCK), 0, s, bsum, nblocks);
    if (zero_count)
        hipLaunchKernelGGL(k_scan_apply<true>, dim3(nblocks), dim3(WS_BLOCK), 0, s, count, nitems, bsum, start_body,
                           cursor);
    else
        hipLaunchKernelGGL(k_scan_apply<false>, dim3(nblocks), dim3(WS_BLOCK), 0, s, count, nitems, bsum, start_body,
                           cursor);
}

// ---------------------------------------------------------------------------------
// K2': counting sort by cell.  Slot assignment inside a cell uses a returning atomic
// (arrival order), k_reorder then ranks the members of each cell by their previous
// index, so the final order is the STABLE sort of the previous order: deterministic.
// ---------------------------------------------------------------------------------
__global__ void __launch_bounds__(WS_BLOCK) k_scatter(const uint32_t *__restrict__ keys, uint32_t *__restrict__ cursor,
                                                      uint32_t *__restrict__ slot_tmp, uint32_t n)
{
    const uint32_t i = blockIdx.x * WS_BLOCK + threadIdx.x;
    if (i >= n) return;
    const uint32_t slot = atomicAdd(&cursor[keys[i]], 1u);
    slot_tmp[slot] = i;
}

void wsk_scatter(hipStream_t s, const uint32_t *keys, uint32_t *cursor, uint32_t *slot_tmp, uint32_t n)
{
    hipLaunchKernelGGL(k_scatter, dim3(cdiv(n, WS_BLOCK)), dim3(WS_BLOCK), 0, s, keys, cursor, slot_tmp, n);
}

__global__ void __launch_bounds__(WS_BLOCK) k_reorder(WsDev d, const uint32_t *__restrict__ slot_tmp,
                                                      const uint32_t *__restrict__ cid_cur,
                                                      const uint32_t *__restrict__ start, WsSoA cur, WsSoA srt,
                                                      uint32_t *__restrict__ cid_srt)
{
    const uint32_t s = blockIdx.x * WS_BLOCK + threadIdx.x;
    if (s >= d.n) return;
    const uint32_t i = slot_tmp[s];
    const uint32_t c = cid_cur[i];
    const uint32_t b = start[d.guard + c], e = start[d.guard + c + 1];
    uint32_t rank = 0;
    for (uint32_t t = b; t < e; t++) rank += (slot_tmp[t] < i) ? 1u : 0u;
    const uint32_t dst = b + rank;
    srt.pos[dst] = cur.pos[i];
    srt.vel[dst] = cur.vel[i];
    srt.pred[dst] = cur.pred[i];
    cid_srt[dst] = c;
}

void wsk_reorder(hipStream_t s, const WsDev &d, const uint32_t *slot_tmp, const uint32_t *cid_cur,
                 const uint32_t *start, WsSoA cur, WsSoA srt, uint32_t *cid_srt)
{
    hipLaunchKernelGGL(k_reorder, dim3(cdiv(d.n, WS_BLOCK)), dim3(WS_BLOCK), 0, s, d, slot_tmp, cid_cur, start, cur,
                       srt, cid_srt);
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
// two stencil cells alias (all the benchmark sizes), more for tiny N.
__device__ __forceinline__ uint32_t alias_mult(const WsDev &d, const uint8_t *__restrict__ mult, float4 a, float4 b)
{
    const int dx = (int)(floorf(b.x / d.h) - floorf(a.x / d.h));
    const int dy = (int)(floorf(b.y / d.h) - floorf(a.y / d.h));
    const int dz = (int)(floorf(b.z / d.h) - floorf(a.z / d.h));
    if (dx < -1 || dx > 1 || dy < -1 || dy > 1 || dz < -1 || dz > 1) return 0u;
    return mult[(dx + 1) * 9 + (dy + 1) * 3 + (dz + 1)];
}

// ---------------------------------------------------------------------------------
// K4 update_density, assets/simulation.wgsl:143-195
// ---------------------------------------------------------------------------------
template <bool ALIAS>
__global__ void __launch_bounds__(WS_BLOCK) k_density(WsDev d, const uint32_t *__restrict__ start,
                                                      const uint32_t *__restrict__ cid_srt,
                                                      const float4 *__restrict__ pred, float2 *__restrict__ dens,
                                                      const uint8_t *__restrict__ mult)
{
    const uint32_t i = blockIdx.x * WS_BLOCK + threadIdx.x;
    if (i >= d.n) return;
    const float4 o = pred[i];
    const int c = (int)cid_srt[i];
    const int rowz = d.dim[2], rowy = d.dim[1] * d.dim[2];
    float density = 0.f, near_density = 0.f;
    for (int dx = -1; dx <= 1; dx++) {
        for (int dy = -1; dy <= 1; dy++) {
            const int cc = d.guard + c + dx * rowy + dy * rowz;
            const uint32_t b = start[cc - 1], e = start[cc + 2];
            for (uint32_t j = b; j < e; j++) {
                const float4 q = pred[j];
                const float ex = q.x - o.x, ey = q.y - o.y, ez = q.z - o.z;
                const float d2 = ex * ex + ey * ey + ez * ez;
                if (d2 > d.d2_accept) continue;
                const float dst = sqrtf(d2);
                const float w = sk_density(d, dst), wn = sk_near(d, dst);
                if (ALIAS) {
                    const uint32_t m = alias_mult(d, mult, o, q);
                    for (uint32_t r = 0; r < m; r++) {
                        density += w;
                        near_density += wn;
                    }
                } else {
                    density += w;
                    near_density += wn;
                }
            }
        }
    }
    density = density + 0.00001f;  // DENSITY_PADDING, simulation.wgsl:4,187-188
    near_density = near_density + 0.00001f;
    dens[i] = make_float2(density, near_density);
}

void wsk_density(hipStream_t s, const WsDev &d, const uint32_t *start, const uint32_t *cid_srt,
                 const float4 *pred, float2 *dens, const uint8_t *mult, bool alias)
{
    if (alias)
        hipLaunchKernelGGL(k_density<true>, dim3(cdiv(d.n, WS_BLOCK)), dim3(WS_BLOCK), 0, s, d, start, cid_srt, pred,
                           dens, mult);
    else
        hipLaunchKernelGGL(k_density<false>, dim3(cdiv(d.n, WS_BLOCK)), dim3(WS_BLOCK), 0, s, d, start, cid_srt, pred,
                           dens, mult);
}

// ---------------------------------------------------------------------------------
// K5 update_pressure_force (simulation.wgsl:197-269) + K6 integrate (:271-310) +
// next step's K1 cell binning, fused: one pass over the sorted particles.
// Reads the sorted copy, writes the ping-pong copy in the same order.
// ---------------------------------------------------------------------------------
template <bool ALIAS>
__global__ void __launch_bounds__(WS_BLOCK) k_force(WsDev d, const uint32_t *__restrict__ start,
                                                    const uint32_t *__restrict__ cid_srt, WsSoA srt,
                                                    const float2 *__restrict__ dens, WsSoA out,
                                                    float4 *__restrict__ accel, uint32_t *__restrict__ cid_out,
                                                    uint32_t *__restrict__ count, const uint8_t *__restrict__ mult)
{
    const uint32_t i = blockIdx.x * WS_BLOCK + threadIdx.x;
    if (i >= d.n) return;
    const float4 o = srt.pred[i];
    const float4 vel = srt.vel[i];
    const float2 rho = dens[i];
    // pressure from density, simulation.wgsl:192-193 (recomputed: same two IEEE ops)
    const float pressure = d.pressure_scalar * (rho.x - d.target_density);
    const float near_pressure = d.near_pressure_scalar * rho.y;
    const int c = (int)cid_srt[i];
    const int rowz = d.dim[2], rowy = d.dim[1] * d.dim[2];

    float pfx = 0.f, pfy = 0.f, pfz = 0.f, vfx = 0.f, vfy = 0.f, vfz = 0.f;
    for (int dx = -1; dx <= 1; dx++) {
        for (int dy = -1; dy <= 1; dy++) {
            const int cc = d.guard + c + dx * rowy + dy * rowz;
            const uint32_t b = start[cc - 1], e = start[cc + 2];
            for (uint32_t j = b; j < e; j++) {
                if (j == i) continue;  // `particle_index == neighbour_index`, :232
                const float4 q = srt.pred[j];
                float ex = q.x - o.x, ey = q.y - o.y, ez = q.z - o.z;
                const float d2 = ex * ex + ey * ey + ez * ez;
                if (d2 > d.d2_accept) continue;
                const float dst = sqrtf(d2);
                if (dst > 0.f) {
                    ex = ex / dst;
                    ey = ey / dst;
                    ez = ez / dst;
                } else {
                    ex = 0.f;
                    ey = 1.f;
                    ez = 0.f;
                }
                const float2 nrho = dens[j];
                const float4 nvel = srt.vel[j];
                const float npress = d.pressure_scalar * (nrho.x - d.target_density);
                const float nnear = d.near_pressure_scalar * nrho.y;
                const float slope = sk_der(d, dst);
                const float shared = (pressure + npress) / 2.f;
                const float slope_near = sk_der_near(d, dst);
                const float shared_near = (near_pressure + nnear) / 2.f;
                const float ax = ex * shared * slope / nrho.x, ay = ey * shared * slope / nrho.x,
                            az = ez * shared * slope / nrho.x;
                const float bx = ex * shared_near * slope_near / nrho.y, by = ey * shared_near * slope_near / nrho.y,
                            bz = ez * shared_near * slope_near / nrho.y;
                const float visc = sk_visc(d, dst);
                const float wx = (nvel.x - vel.x) * visc, wy = (nvel.y - vel.y) * visc, wz = (nvel.z - vel.z) * visc;
                uint32_t m = 1;
                if (ALIAS) m = alias_mult(d, mult, o, q);
                for (uint32_t r = 0; r < m; r++) {
                    pfx += ax; pfy += ay; pfz += az;
                    pfx += bx; pfy += by; pfz += bz;
                    vfx += wx; vfy += wy; vfz += wz;
                }
            }
        }
    }
    // simulation.wgsl:265-268
    const float accx = pfx / rho.x + vfx * d.viscosity;
    const float accy = pfy / rho.x + vfy * d.viscosity;
    const float accz = pfz / rho.x + vfz * d.viscosity;
    accel[i] = make_float4(accx, accy, accz, 0.f);

    // K6 integrate, simulation.wgsl:279-309
    const float4 p0 = srt.pos[i];
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
    const float LOOKAHEAD = 0.02f;  // 1. / 50., simulation.wgsl:3
    const float qx = px + vx * LOOKAHEAD, qy = py + vy * LOOKAHEAD, qz = pz + vz * LOOKAHEAD;
    out.pos[i] = make_float4(px, py, pz, p0.w);
    out.vel[i] = make_float4(vx, vy, vz, 0.f);
    out.pred[i] = make_float4(qx, qy, qz, 0.f);

    // next step's hash_particles (simulation.wgsl:130-141) on the dense grid
    const uint32_t nc = grid_cell(d, qx, qy, qz);
    cid_out[i] = nc;
    atomicAdd(&count[nc], 1u);
}

void wsk_force(hipStream_t s, const WsDev &d, const uint32_t *start, const uint32_t *cid_srt, WsSoA srt,
               const float2 *dens, WsSoA out, float4 *accel, uint32_t *cid_out, uint32_t *count,
               const uint8_t *mult, bool alias)
{
    if (alias)
        hipLaunchKernelGGL(k_force<true>, dim3(cdiv(d.n, WS_BLOCK)), dim3(WS_BLOCK), 0, s, d, start, cid_srt, srt, dens,
                           out, accel, cid_out, count, mult);
    else
        hipLaunchKernelGGL(k_force<false>, dim3(cdiv(d.n, WS_BLOCK)), dim3(WS_BLOCK), 0, s, d, start, cid_srt, srt,
                           dens, out, accel, cid_out, count, mult);
}

// ---------------------------------------------------------------------------------
// readback in original-id order (update(), src/fluid_compute.rs:478-485)
// ---------------------------------------------------------------------------------
__global__ void __launch_bounds__(WS_BLOCK) k_gather_positions(const float4 *__restrict__ pos, float *__restrict__ out,
                                                               uint32_t n)
{
    const uint32_t i = blockIdx.x * WS_BLOCK + threadIdx.x;
    if (i >= n) return;
    const float4 p = pos[i];
    const size_t id = __float_as_uint(p.w);
    out[3 * id] = p.x;
    out[3 * id + 1] = p.y;
    out[3 * id + 2] = p.z;
}

void wsk_gather_positions(hipStream_t s, const float4 *pos, float *out_xyz, uint32_t n)
{
    hipLaunchKernelGGL(k_gather_positions, dim3(cdiv(n, WS_BLOCK)), dim3(WS_BLOCK), 0, s, pos, out_xyz, n);
}

__global__ void __launch_bounds__(WS_BLOCK) k_gather_particles(WsDev d, WsSoA cur, const float2 *__restrict__ dens,
                                                               const float4 *__restrict__ accel, int have_step,
                                                               ws_particle80 *__restrict__ out, uint32_t n)
{
    const uint32_t i = blockIdx.x * WS_BLOCK + threadIdx.x;
    if (i >= n) return;
    const float4 p = cur.pos[i], v = cur.vel[i], q = cur.pred[i];
    const size_t id = __float_as_uint(p.w);
    float4 dp = make_float4(0.f, 0.f, 0.f, 0.f), a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (have_step) {
        const float2 r = dens[i];
        dp.x = r.x;
        dp.y = r.y;
        dp.z = d.pressure_scalar * (r.x - d.target_density);  // simulation.wgsl:192-193
        dp.w = d.near_pressure_scalar * r.y;
        a = accel[i];
    }
    float4 *rec = reinterpret_cast<float4 *>(out + id);
    rec[0] = make_float4(p.x, p.y, p.z, 0.f);
    rec[1] = dp;
    rec[2] = make_float4(v.x, v.y, v.z, 0.f);
    rec[3] = make_float4(a.x, a.y, a.z, 0.f);
    rec[4] = make_float4(q.x, q.y, q.z, 0.f);
}

void wsk_gather_particles(hipStream_t s, const WsDev &d, WsSoA cur, const float2 *dens, const float4 *accel,
                          bool have_step, ws_particle80 *out, uint32_t n)
{
    hipLaunchKernelGGL(k_gather_particles, dim3(cdiv(n, WS_BLOCK)), dim3(WS_BLOCK), 0, s, d, cur, dens, accel,
                       have_step ? 1 : 0, out, n);
}

// ---------------------------------------------------------------------------------
// reference-layout sort view (diagnostic, on demand; never part of ws_step)
// ---------------------------------------------------------------------------------
__global__ void __launch_bounds__(WS_BLOCK) k_view_keys(WsDev d, const float4 *__restrict__ pred,
                                                        const float4 *__restrict__ pos_with_id,
                                                        uint32_t *__restrict__ keys_by_id, uint32_t *__restrict__ count)
{
    const uint32_t i = blockIdx.x * WS_BLOCK + threadIdx.x;
    if (i >= d.n) return;
    const float4 p = pred[i];
    const uint32_t id = __float_as_uint(pos_with_id[i].w);
    const uint32_t key = ref_hash_key(d, p.x, p.y, p.z);
    keys_by_id[id] = key;
    atomicAdd(&count[key], 1u);
}

void wsk_view_keys(hipStream_t s, const WsDev &d, const float4 *pred, const float4 *pos_with_id,
                   uint32_t *keys_by_id, uint32_t *count)
{
    hipLaunchKernelGGL(k_view_keys, dim3(cdiv(d.n, WS_BLOCK)), dim3(WS_BLOCK), 0, s, d, pred, pos_with_id, keys_by_id,
                       count);
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

__global__ void __launch_bounds__(WS_BLOCK) k_iota(uint32_t *__restrict__ p, uint32_t n)
{
    const uint32_t i = blockIdx.x * WS_BLOCK + threadIdx.x;
    if (i < n) p[i] = i;
}

void wsk_iota(hipStream_t s, uint32_t *p, uint32_t n)
{
    hipLaunchKernelGGL(k_iota, dim3(cdiv(n, WS_BLOCK)), dim3(WS_BLOCK), 0, s, p, n);
}
