// Developer (round 4): what does a per-lane (divergent) gather cost on gfx950, by load width and by how lanes share
// cache lines?  K5 fetches a neighbour's 32-byte record as two 16-B loads per lane; its texture path is the busiest
// unit in both states.  This prices that pattern and its alternatives from tables that sit in the L1
// (16 KB shared), the L2 (2 MB) or memory (1 GB), as lane-loads per second over the whole chip.
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/gather_probe tools/gather_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

enum { P_B128 = 0, P_B128_PAIR, P_B64, P_B32, P_B128_COALESCED, P_2XB128_SAME_REC, P_B128_QUAD, P_8XB32_PLANAR, P_B128_PAIR32, P_B128_PAIR16, P_COUNT };
static const char *kNames[P_COUNT] = {"b128 per-lane records", "b128 lane pairs share a 32-B record", "b64 per-lane", "b32 per-lane",
                                      "b128 coalesced (lane i -> base + 16 i)", "2 x b128, both halves of a per-lane 32-B record (K5)",
                                      "b128, quads of lanes share a 64-B segment", "8 x b32 from 8 planes, per-lane index", "b128, lanes l / l+32 share a 32-B record",
                                      "b128, lanes l / l+16 share a 32-B record"};
// loads per lane per iteration of each pattern
static const int kLoads[P_COUNT] = {1, 1, 1, 1, 1, 2, 1, 8, 1, 1};

template <int P>
__global__ void __launch_bounds__(256) k_gather(const uint32_t *__restrict__ table, uint32_t mask16, uint32_t wg_span16, int iters,
                                                uint32_t *__restrict__ sink)
{
    // the table is addressed in 16-byte units; a workgroup keeps to [blockIdx * wg_span16, +mask16] (wg_span16 = 0: all share)
    const uint32_t base16 = blockIdx.x * wg_span16;
    // U independent index streams per lane: U iterations' loads are in flight together (one load per wave at a time would
    // measure the latency, not the texture path)
    constexpr int U = 4;
    uint32_t state[U];
#pragma unroll
    for (int u = 0; u < U; u++) state[u] = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u + 977u * u;
    uint32_t acc = 0;
    const uint32_t lane = threadIdx.x & 63u;
    for (int it = 0; it < iters; it += U) {
        uint32_t r[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            state[u] = state[u] * 1664525u + 1013904223u;
            r[u] = state[u] >> 7;
            if (P == P_B128_PAIR) r[u] = __shfl(r[u], (int)(lane & ~1u)) ^ (lane & 1u);           // pair: same 32-B record, the two halves
            else if (P == P_B128_QUAD) r[u] = (__shfl(r[u], (int)(lane & ~3u)) & ~3u) | (lane & 3u);  // quad: one 64-B segment
            else if (P == P_B128_PAIR32) r[u] = (__shfl(r[u], (int)(lane & 31u)) & ~1u) | (lane >> 5);
            else if (P == P_B128_PAIR16) r[u] = (__shfl(r[u], (int)(lane & ~16u)) & ~1u) | ((lane >> 4) & 1u);
            else if (P == P_B128_COALESCED) r[u] = (__shfl(r[u], 0) & ~63u) + lane;
        }
        if (P == P_2XB128_SAME_REC) {
            uint4 a[U], b[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const uint32_t i16 = base16 + ((r[u] & mask16) & ~1u);
                a[u] = reinterpret_cast<const uint4 *>(table)[i16];
                b[u] = reinterpret_cast<const uint4 *>(table)[i16 + 1];
            }
#pragma unroll
            for (int u = 0; u < U; u++) acc ^= a[u].x ^ a[u].y ^ a[u].z ^ a[u].w ^ b[u].x ^ b[u].y ^ b[u].z ^ b[u].w;  // every component: or the load is narrowed
        } else if (P == P_B64) {
            uint2 a[U];
#pragma unroll
            for (int u = 0; u < U; u++) a[u] = reinterpret_cast<const uint2 *>(table)[2u * (base16 + (r[u] & mask16))];
#pragma unroll
            for (int u = 0; u < U; u++) acc ^= a[u].x ^ a[u].y;
        } else if (P == P_B32) {
            uint32_t a[U];
#pragma unroll
            for (int u = 0; u < U; u++) a[u] = table[4u * (base16 + (r[u] & mask16))];
#pragma unroll
            for (int u = 0; u < U; u++) acc ^= a[u];
        } else if (P == P_8XB32_PLANAR) {
            // eight planes of (mask16 + 1) / 2 words each inside the same span: the record's eight floats kept planar
            const uint32_t words = (mask16 + 1u) >> 1;
            uint32_t a[U][8];
#pragma unroll
            for (int u = 0; u < U; u++)
#pragma unroll
                for (int q = 0; q < 8; q++) a[u][q] = table[4u * base16 + q * words + ((r[u] & mask16) >> 1)];
#pragma unroll
            for (int u = 0; u < U; u++)
#pragma unroll
                for (int q = 0; q < 8; q++) acc ^= a[u][q];
        } else {
            uint4 a[U];
#pragma unroll
            for (int u = 0; u < U; u++) a[u] = reinterpret_cast<const uint4 *>(table)[base16 + (r[u] & mask16)];
#pragma unroll
            for (int u = 0; u < U; u++) acc ^= a[u].x ^ a[u].y ^ a[u].z ^ a[u].w;
        }
    }
    if (acc == 0x9E3779B9u) sink[blockIdx.x] = acc;
}

__global__ void k_clock(uint64_t *out)
{
    const uint64_t c0 = clock64(), w0 = wall_clock64();
    uint32_t x = threadIdx.x;
    for (int i = 0; i < 200000; i++) x = x * 1664525u + 1013904223u;
    const uint64_t c1 = clock64(), w1 = wall_clock64();
    if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = w1 - w0; out[2] = x; }
}

template <int P>
static void run(const char *where, const uint32_t *table, uint32_t mask16, uint32_t wg_span16, uint32_t *sink)
{
    const int blocks = 256 * 8, iters = 2000;  // 8 workgroups of 4 waves per CU: 8 waves per SIMD
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL((k_gather<P>), dim3(blocks), dim3(256), 0, 0, table, mask16, wg_span16, 50, sink);
    CK(hipEventRecord(a, 0));
    hipLaunchKernelGGL((k_gather<P>), dim3(blocks), dim3(256), 0, 0, table, mask16, wg_span16, iters, sink);
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    const double instr = (double)blocks * 4 * iters * kLoads[P];  // wave-level load instructions
    const double per_cu_clk = instr / 256.0 / (ms * 1e-3 * 2.4e9);  // load instructions per CU per 2.4 GHz clock
    printf("{\"table\": \"%s\", \"pattern\": \"%s\", \"ms\": %.3f, \"G_lane_loads_per_s\": %.1f, \"clocks_per_wave_load_per_cu\": %.1f}\n", where,
           kNames[P], ms, instr * 64 / ms * 1e-6, 1.0 / per_cu_clk);
    fflush(stdout);
}

int main()
{
    const size_t bytes = (size_t)1 << 30;
    uint32_t *table, *sink;
    CK(hipMalloc(&table, bytes));
    CK(hipMemset(table, 1, bytes));
    CK(hipMalloc(&sink, 1 << 20));
    {
        uint64_t *c, hc[3];
        CK(hipMalloc(&c, 24));
        hipLaunchKernelGGL(k_clock, dim3(1), dim3(64), 0, 0, c);
        CK(hipMemcpy(hc, c, 24, hipMemcpyDeviceToHost));
        printf("{\"shader_clocks\": %llu, \"wall_clock_ticks_100MHz\": %llu, \"shader_GHz_if_clock64_counts_shader_clocks\": %.3f}\n",
               (unsigned long long)hc[0], (unsigned long long)hc[1], (double)hc[0] / ((double)hc[1] / 100e6) * 1e-9);
    }
    struct { const char *where; uint32_t mask16, span16; } tables[] = {
        {"L1 (one 16 KB table shared by all)", (16u << 10) / 16 - 1, 0},
        {"L2 (16 KB per workgroup: 128 KB per CU)", (16u << 10) / 16 - 1, (16u << 10) / 16},
        {"L2 (2 MB shared)", (2u << 20) / 16 - 1, 0},
        {"memory (1 GB shared)", (uint32_t)(bytes / 16 - 1), 0},
    };
    for (auto &t : tables) {
        run<P_B128_COALESCED>(t.where, table, t.mask16, t.span16, sink);
        run<P_B128>(t.where, table, t.mask16, t.span16, sink);
        run<P_2XB128_SAME_REC>(t.where, table, t.mask16, t.span16, sink);
        run<P_B128_PAIR>(t.where, table, t.mask16, t.span16, sink);
        run<P_B128_QUAD>(t.where, table, t.mask16, t.span16, sink);
        run<P_B128_PAIR32>(t.where, table, t.mask16, t.span16, sink);
        run<P_B128_PAIR16>(t.where, table, t.mask16, t.span16, sink);
        run<P_B64>(t.where, table, t.mask16, t.span16, sink);
        run<P_B32>(t.where, table, t.mask16, t.span16, sink);
        run<P_8XB32_PLANAR>(t.where, table, t.mask16, t.span16, sink);
    }
    return 0;
}
