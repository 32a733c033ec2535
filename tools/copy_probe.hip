// Developer probe (round 4): how does a 50 MB device->host copy share an MI355X with a compute kernel?
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/copy_probe tools/copy_probe.hip && tools/bin/copy_probe
// Host buffer kinds: hipHostMalloc (default / non-coherent), malloc + hipHostRegister.  Copy kinds: hipMemcpyAsync
// on a second stream, and a copy KERNEL of a few workgroups storing straight into the mapped host buffer.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void k_busy(float *out, int iters)
{
    float a = threadIdx.x * 1e-3f, b = 1.0001f;
    for (int i = 0; i < iters; i++) a = a * b + 0.5f;
    if (a == 123.f) out[0] = a;
}

__global__ void k_stream(const float4 *in, float4 *out, size_t n)  // HBM-bound compute stand-in
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i];
}

__global__ void k_copy_to_host(const float4 *__restrict__ src, float4 *__restrict__ dst, size_t n16)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x)
        dst[i] = src[i];
}

static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main()
{
    const size_t bytes = (size_t)4194304 * 12;
    float *dev, *scratch;
    CK(hipMalloc(&dev, bytes));
    CK(hipMalloc(&scratch, 4096));
    float4 *big_in, *big_out;
    const size_t big = (size_t)1 << 30;
    CK(hipMalloc(&big_in, big));
    CK(hipMalloc(&big_out, big));
    CK(hipMemset(dev, 1, bytes));
    hipStream_t sa, sb;
    CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    struct Host { const char *name; void *p; };
    std::vector<Host> hosts;
    void *p;
    CK(hipHostMalloc(&p, bytes, hipHostMallocDefault));
    hosts.push_back({"hipHostMalloc(default)", p});
    CK(hipHostMalloc(&p, bytes, hipHostMallocNonCoherent));
    hosts.push_back({"hipHostMalloc(noncoherent)", p});
    p = aligned_alloc(4096, bytes);
    CK(hipHostRegister(p, bytes, hipHostRegisterDefault));
    hosts.push_back({"malloc+hipHostRegister", p});

    auto busy_valu = [&]() { hipLaunchKernelGGL(k_busy, dim3(256 * 8), dim3(256), 0, sa, scratch, 60000); };
    auto busy_hbm = [&]() { hipLaunchKernelGGL(k_stream, dim3(256 * 16), dim3(256), 0, sa, big_in, big_out, big / 16); };
    auto time_it = [&](auto &&f, int reps) {
        f();
        CK(hipDeviceSynchronize());
        const double t0 = now();
        for (int i = 0; i < reps; i++) f();
        CK(hipDeviceSynchronize());
        return (now() - t0) / reps;
    };
    const double tv = time_it(busy_valu, 5), th = time_it(busy_hbm, 5);
    printf("compute stand-ins alone: VALU-bound %.3f ms, HBM-bound %.3f ms (2 GiB moved: %.0f GB/s)\n", tv, th, 2.0 * big / th / 1e6);
    for (auto &h : hosts) {
        auto copy = [&]() { CK(hipMemcpyAsync(h.p, dev, bytes, hipMemcpyDeviceToHost, sb)); };
        const double tc = time_it(copy, 10);
        const double tbv = time_it([&]() { busy_valu(); copy(); }, 5);
        const double tbh = time_it([&]() { busy_hbm(); copy(); }, 5);
        printf("%-28s hipMemcpyAsync alone %.3f ms (%.1f GB/s); with VALU kernel %.3f (sum %.3f, max %.3f); with HBM kernel %.3f (sum %.3f, max %.3f)\n",
               h.name, tc, bytes / tc / 1e6, tbv, tv + tc, tv > tc ? tv : tc, tbh, th + tc, th > tc ? th : tc);
        void *dp = nullptr;
        CK(hipHostGetDevicePointer(&dp, h.p, 0));
        for (int wgs : {4, 8, 16, 32, 64, 256}) {
            auto kcopy = [&]() { hipLaunchKernelGGL(k_copy_to_host, dim3(wgs), dim3(256), 0, sb, (const float4 *)dev, (float4 *)dp, bytes / 16); };
            const double tk = time_it(kcopy, 10);
            const double tkv = time_it([&]() { busy_valu(); kcopy(); }, 5);
            const double tkh = time_it([&]() { busy_hbm(); kcopy(); }, 5);
            printf("    copy kernel %3d workgroups: alone %.3f ms (%.1f GB/s); with VALU kernel %.3f; with HBM kernel %.3f\n", wgs, tk,
                   bytes / tk / 1e6, tkv, tkh);
        }
    }
    return 0;
}
