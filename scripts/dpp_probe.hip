// Does gfx950 implement the whole-wave DPP shifts (wave_shr:1 / wave_shl:1) as the GFX9 ISA describes them?
// lane i of wave_shr:1 must receive lane i-1 (lane 0 keeps `old`), lane i of wave_shl:1 lane i+1 (lane 63 keeps `old`).
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/dpp_probe scripts/dpp_probe.hip && /tmp/dpp_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int *out) {
    const int x = (int)threadIdx.x * 3 + 7;
    out[threadIdx.x] = __builtin_amdgcn_update_dpp(-1, x, 0x138, 0xf, 0xf, false);
    out[64 + threadIdx.x] = __builtin_amdgcn_update_dpp(-2, x, 0x130, 0xf, 0xf, false);
}
int main() {
    int *d, h[128];
    hipMalloc(&d, sizeof h);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 64; i++) {
        const int want_prev = i == 0 ? -1 : (i - 1) * 3 + 7, want_next = i == 63 ? -2 : (i + 1) * 3 + 7;
        if (h[i] != want_prev || h[64 + i] != want_next) bad++;
    }
    printf("dpp wave shifts: %s (%d mismatches)\n", bad ? "BROKEN" : "ok", bad);
    return bad != 0;
}
