// Developer (round 4): kernels of KNOWN read byte counts, to calibrate the L2 / fabric read counters of gfx950 against
// the access patterns this library uses (tools/traffic_calib.sh runs it under rocprofv3 --pmc).
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/traffic_calib tools/traffic_calib.hip
// Every kernel reads `n` elements from a 1 GiB source (four times the Infinity Cache, so nothing is served on-die) and
// writes 4 bytes per THREAD BLOCK (nothing to speak of).  Patterns:
//   stream16 / stream8 / stream4   coalesced loads of 16 / 8 / 4 bytes per lane
//   gather16                       16-B loads through a random permutation (every 16-B record read exactly once)
//   gather32                       two 16-B loads of one 32-B record through a random permutation (K5's neighbour gather)
//   gather16_local                 16-B loads through a permutation that shuffles within windows of 64 records only
//                                  (the sort's reorder pass: its sources are nearby)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#include <numeric>
#include <random>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <class T>
__global__ void k_stream(const T *__restrict__ src, uint32_t *__restrict__ sink, size_t n)
{
    uint32_t acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const T v = src[i];
        acc ^= reinterpret_cast<const uint32_t *>(&v)[0];
    }
    if (acc == 0x12345678u) sink[blockIdx.x] = acc;
}
__global__ void k_gather16(const uint4 *__restrict__ src, const uint32_t *__restrict__ perm, uint32_t *__restrict__ sink, size_t n)
{
    uint32_t acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc ^= src[perm[i]].x;
    if (acc == 0x12345678u) sink[blockIdx.x] = acc;
}
__global__ void k_gather32(const uint4 *__restrict__ src, const uint32_t *__restrict__ perm, uint32_t *__restrict__ sink, size_t n)
{
    uint32_t acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t j = perm[i];
        acc ^= src[2 * j].x ^ src[2 * j + 1].y;
    }
    if (acc == 0x12345678u) sink[blockIdx.x] = acc;
}

int main()
{
    const size_t bytes = (size_t)1 << 30;
    void *src;
    uint32_t *sink, *perm, *perm_local, *perm32;
    CK(hipMalloc(&src, bytes));
    CK(hipMemset(src, 1, bytes));
    CK(hipMalloc(&sink, 1 << 20));
    const size_t n16 = bytes / 16, n32 = bytes / 32;
    std::vector<uint32_t> p(n16);
    std::iota(p.begin(), p.end(), 0u);
    std::mt19937_64 rng(7);
    std::vector<uint32_t> pl = p;
    for (size_t w = 0; w + 64 <= n16; w += 64) std::shuffle(pl.begin() + w, pl.begin() + w + 64, rng);
    std::shuffle(p.begin(), p.end(), rng);
    std::vector<uint32_t> p32(n32);
    std::iota(p32.begin(), p32.end(), 0u);
    std::shuffle(p32.begin(), p32.end(), rng);
    CK(hipMalloc(&perm, n16 * 4)); CK(hipMemcpy(perm, p.data(), n16 * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&perm_local, n16 * 4)); CK(hipMemcpy(perm_local, pl.data(), n16 * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&perm32, n32 * 4)); CK(hipMemcpy(perm32, p32.data(), n32 * 4, hipMemcpyHostToDevice));
    const dim3 g(256 * 16), b(256);
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(k_stream<uint4>, g, b, 0, 0, (const uint4 *)src, sink, bytes / 16);
        hipLaunchKernelGGL(k_stream<uint2>, g, b, 0, 0, (const uint2 *)src, sink, bytes / 8);
        hipLaunchKernelGGL(k_stream<uint32_t>, g, b, 0, 0, (const uint32_t *)src, sink, bytes / 4);
        hipLaunchKernelGGL(k_gather16, g, b, 0, 0, (const uint4 *)src, perm, sink, n16);
        hipLaunchKernelGGL(k_gather16, g, b, 0, 0, (const uint4 *)src, perm_local, sink, n16);
        hipLaunchKernelGGL(k_gather32, g, b, 0, 0, (const uint4 *)src, perm32, sink, n32);
        CK(hipDeviceSynchronize());
    }
    // expected read bytes per launch, in launch order (the permutation itself is a coalesced 4-B stream)
    printf("{\"expected_read_bytes\": {\"stream16\": %zu, \"stream8\": %zu, \"stream4\": %zu, \"gather16\": %zu, \"gather16_local\": %zu, \"gather32\": %zu}}\n",
           bytes, bytes, bytes, bytes + n16 * 4, bytes + n16 * 4, bytes + n32 * 4);
    return 0;
}
