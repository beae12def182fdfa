/* launch_args_probe.hip -- does the way the arguments reach a launch matter?  A kernel with 832 bytes of by-value
 * arguments (what pya_one_kernel takes) launched through hipLaunchKernelGGL, and through hipModuleLaunchKernel with one
 * prepacked argument buffer (HIP_LAUNCH_PARAM_BUFFER_POINTER); the host polls a flag in pinned memory both times.
 *   hipcc --offload-arch=gfx950 -O2 -o launch_args_probe scripts/launch_args_probe.hip && timeout 60 ./launch_args_probe */
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>
#include <immintrin.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

struct Big { uint64_t w[100]; };          /* 800 bytes */
__global__ void k_big(Big a, uint32_t *flag, uint32_t seq, uint64_t *sink) {
    uint64_t acc = 0;
    for (int i = 0; i < 100; i++) acc += a.w[i];
    if (acc == 0x1234567812345678ull) sink[0] = acc;
    __threadfence_system();
    if (threadIdx.x == 0) __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
static double med(std::vector<double> &v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

int main() {
    unsigned char *host, *dev;
    CHECK(hipHostMalloc((void **)&host, 4096, hipHostMallocMapped | hipHostMallocCoherent));
    CHECK(hipHostGetDevicePointer((void **)&dev, host, 0));
    std::memset(host, 0, 4096);
    uint64_t *sink;
    CHECK(hipMalloc(&sink, 64));
    hipStream_t st;
    CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    volatile uint32_t *flag = (volatile uint32_t *)host;
    Big a;
    for (int i = 0; i < 100; i++) a.w[i] = i;
    const int N = 3000;
    {
        std::vector<double> tl, tt;
        for (int i = 1; i <= N; i++) {
            const auto t0 = std::chrono::steady_clock::now();
            hipLaunchKernelGGL(k_big, dim3(1), dim3(64), 0, st, a, (uint32_t *)dev, (uint32_t)i, sink);
            const auto t1 = std::chrono::steady_clock::now();
            while (*flag != (uint32_t)i) _mm_pause();
            const auto t2 = std::chrono::steady_clock::now();
            tl.push_back(std::chrono::duration<double, std::micro>(t1 - t0).count());
            tt.push_back(std::chrono::duration<double, std::micro>(t2 - t0).count());
        }
        std::printf("hipLaunchKernelGGL, 832 B of arguments:          launch call %.2f us, until the flag %.2f us\n", med(tl), med(tt));
    }
    {
        hipFunction_t fn;
        CHECK(hipGetFuncBySymbol(&fn, (const void *)k_big));
        struct __attribute__((packed, aligned(8))) Args { Big a; uint32_t *flag; uint32_t seq; uint32_t pad; uint64_t *sink; } args;
        args.a = a; args.flag = (uint32_t *)dev; args.pad = 0; args.sink = sink;
        size_t sz = sizeof args;
        void *extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
        std::vector<double> tl, tt;
        for (int i = N + 1; i <= 2 * N; i++) {
            const auto t0 = std::chrono::steady_clock::now();
            args.seq = (uint32_t)i;
            CHECK(hipModuleLaunchKernel(fn, 1, 1, 1, 64, 1, 1, 0, st, nullptr, extra));
            const auto t1 = std::chrono::steady_clock::now();
            while (*flag != (uint32_t)i) _mm_pause();
            const auto t2 = std::chrono::steady_clock::now();
            tl.push_back(std::chrono::duration<double, std::micro>(t1 - t0).count());
            tt.push_back(std::chrono::duration<double, std::micro>(t2 - t0).count());
        }
        std::printf("hipModuleLaunchKernel, one prepacked buffer:     launch call %.2f us, until the flag %.2f us\n", med(tl), med(tt));
    }
    CHECK(hipStreamSynchronize(st));
    return 0;
}
