/* launch_probe.hip -- what one PSM per call costs before any scoring happens: (a) a kernel launch whose last store the
 * host polls in pinned memory (how pya_score_one works), (b) the same with 5 KB read from pinned memory first, (c) a
 * resident kernel that polls a mailbox in pinned memory (with an idle timeout, so that it can never outlive its host
 * by more than 50 ms) and answers, with and without the 5 KB.
 *   hipcc --offload-arch=gfx950 -O2 -o launch_probe scripts/launch_probe.hip && timeout 60 ./launch_probe */
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>
#include <immintrin.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_flag(uint32_t *flag, uint32_t seq, const double *payload, int n64, double *sink) {
    double acc = 0.;
    for (int i = 0; i < n64; i++) acc += payload[i * 64 + threadIdx.x];
    if (n64 && acc == 12345.678) sink[0] = acc;
    __threadfence_system();
    if (threadIdx.x == 0) __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ void k_mailbox(uint32_t *req, uint32_t *ack, const double *payload, int n64, double *sink, uint64_t idle_ticks) {
    uint32_t last = 0;
    uint64_t t_idle = __builtin_amdgcn_s_memrealtime();
    for (;;) {
        uint32_t r = 0;
        if (threadIdx.x == 0) r = __hip_atomic_load(req, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
        r = (uint32_t)__builtin_amdgcn_readfirstlane((int)r);
        if (r == last) {
            if (__builtin_amdgcn_s_memrealtime() - t_idle > idle_ticks) return;      /* nobody talks to us any more */
            __builtin_amdgcn_s_sleep(2);
            continue;
        }
        if (r == 0xffffffffu) return;
        last = r;
        double acc = 0.;
        for (int i = 0; i < n64; i++) acc += __builtin_nontemporal_load(payload + i * 64 + threadIdx.x);
        if (n64 && acc == 12345.678) sink[0] = acc;
        __threadfence_system();
        if (threadIdx.x == 0) __hip_atomic_store(ack, r, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        t_idle = __builtin_amdgcn_s_memrealtime();
    }
}

static double med(std::vector<double> &v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

int main() {
    unsigned char *host, *dev;
    CHECK(hipHostMalloc((void **)&host, 1 << 16, hipHostMallocMapped | hipHostMallocCoherent));
    CHECK(hipHostGetDevicePointer((void **)&dev, host, 0));
    std::memset(host, 0, 1 << 16);
    double *sink;
    CHECK(hipMalloc(&sink, 64));
    hipStream_t st;
    CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    volatile uint32_t *flag = (volatile uint32_t *)host;
    volatile uint32_t *req = (volatile uint32_t *)(host + 64), *ack = (volatile uint32_t *)(host + 128);
    const double *pay = (const double *)(dev + 1024);
    const int N = 3000;
    for (int n64 : {0, 10}) {
        std::vector<double> tl, tt;
        for (int i = 1; i <= N; i++) {
            const auto t0 = std::chrono::steady_clock::now();
            hipLaunchKernelGGL(k_flag, dim3(1), dim3(64), 0, st, (uint32_t *)dev, (uint32_t)i, pay, n64, sink);
            const auto t1 = std::chrono::steady_clock::now();
            while (*flag != (uint32_t)i) _mm_pause();
            const auto t2 = std::chrono::steady_clock::now();
            tl.push_back(std::chrono::duration<double, std::micro>(t1 - t0).count());
            tt.push_back(std::chrono::duration<double, std::micro>(t2 - t0).count());
        }
        CHECK(hipStreamSynchronize(st));
        std::printf("launch + poll, %4d B from pinned memory: launch call %.1f us, until the flag %.1f us (medians of %d)\n", n64 * 512, med(tl), med(tt), N);
    }
    for (int n64 : {0, 10}) {
        *req = 0;
        *ack = 0;
        hipLaunchKernelGGL(k_mailbox, dim3(1), dim3(64), 0, st, (uint32_t *)(dev + 64), (uint32_t *)(dev + 128), pay, n64, sink,
                           (uint64_t)5000000 /* 50 ms of the 100 MHz clock */);
        std::vector<double> tt;
        for (int i = 1; i <= N; i++) {
            const auto t0 = std::chrono::steady_clock::now();
            *req = (uint32_t)i;
            while (*ack != (uint32_t)i) {
                _mm_pause();
                if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(1)) { std::printf("mailbox: no answer\n"); *req = 0xffffffffu; return 2; }
            }
            tt.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
        }
        *req = 0xffffffffu;
        CHECK(hipStreamSynchronize(st));
        std::printf("resident kernel, %4d B from pinned memory: request -> answer %.1f us (median of %d)\n", n64 * 512, med(tt), N);
    }
    return 0;
}
