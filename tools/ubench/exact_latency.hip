// The exact render's two kernels on 1 .. 2 048 identical synthetic cars: what a car costs alone on a CU (latency of the whole
// dependent work) against what it costs with every CU busy (the memory system shared); then the prefilter's phases on one car
// (the header's PXR_STAMP, compiled in here only).  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define PXR_STAMPS 1
#include "../../racing_dreamer_amd/csrc/racecar_patch_exact.h"
#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(_e), __LINE__); return 1; } } while (0)
int main() {
    const int h = 400, w = 400, pitch = (w + 31) / 32, N = 2048;
    std::vector<uint32_t> drv((size_t)h * pitch, 0u);
    for (int y = 100; y < 300; ++y) for (int x = 150; x < 260; ++x) drv[(size_t)y * pitch + (x >> 5)] |= 1u << (x & 31);
    std::vector<float> hx(N, 10.0f), hy(N, 10.0f), hth(N, 0.3f); std::vector<uint8_t> hf(N, 0);
    uint32_t *d_drv; float *d_x, *d_y, *d_th; uint8_t *d_f, *d_patch; double *d_s; int32_t *d_kk;
    CK(hipMalloc(&d_drv, drv.size() * 4)); CK(hipMemcpy(d_drv, drv.data(), drv.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_x, 4 * N)); CK(hipMalloc(&d_y, 4 * N)); CK(hipMalloc(&d_th, 4 * N)); CK(hipMalloc(&d_f, N)); CK(hipMalloc(&d_patch, 4096 * (size_t)N));
    CK(hipMemcpy(d_x, hx.data(), 4 * N, hipMemcpyHostToDevice)); CK(hipMemcpy(d_y, hy.data(), 4 * N, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_th, hth.data(), 4 * N, hipMemcpyHostToDevice)); CK(hipMemcpy(d_f, hf.data(), N, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_s, (size_t)RC_EXACT_CAR_DOUBLES * 8 * N)); CK(hipMemset(d_s, 0, (size_t)RC_EXACT_CAR_DOUBLES * 8 * N));
    std::vector<int32_t> kk(RC_EXACT_TABLE_INTS, 0);
    for (int xx = 0; xx < 64; ++xx) { for (int k = 0; k < 7; ++k) kk[xx * 15 + k] = (1 << 22) / 7; kk[64 * 15 + 2 * xx] = xx * 3; kk[64 * 15 + 2 * xx + 1] = 7; }
    CK(hipMalloc(&d_kk, kk.size() * 4)); CK(hipMemcpy(d_kk, kk.data(), kk.size() * 4, hipMemcpyHostToDevice));
    RcExactParams p{};
    p.drv_words = d_drv; p.pitch = pitch; p.h = h; p.w = w; p.x = d_x; p.y = d_y; p.theta = d_th; p.fresh = d_f;
    p.fh = 400; p.r_top = 399; p.c0 = 0; p.ox = 0.0; p.oy = 0.0; p.res = 0.05; p.scratch = d_s; p.patch = d_patch; p.kk = d_kk; p.car0 = 0; p.n_cars = N;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int grids[] = {1, 64, 256, 512, 2048};
    for (int g : grids) {
        float best[2] = {1e9f, 1e9f};
        for (int rep = 0; rep < 5; ++rep) {
            float ms;
            CK(hipEventRecord(a)); hipLaunchKernelGGL(rc_patch_exact_prefilter_kernel, dim3(g), dim3(256), 0, 0, p); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            CK(hipEventElapsedTime(&ms, a, b)); best[0] = ms < best[0] ? ms : best[0];
            CK(hipEventRecord(a)); hipLaunchKernelGGL(rc_patch_exact_sample_kernel<false>, dim3(g), dim3(PX_ST), 0, 0, p); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            CK(hipEventElapsedTime(&ms, a, b)); best[1] = ms < best[1] ? ms : best[1];
        }
        printf("%5d cars: prefilter %8.1f us   sample %8.1f us\n", g, best[0] * 1e3, best[1] * 1e3);
    }
    // phases of the prefilter, one car alone: clock stamps of lane 0 of each wave (in the car's patch bytes)
    hipLaunchKernelGGL(rc_patch_exact_prefilter_kernel, dim3(1), dim3(256), 0, 0, p); CK(hipDeviceSynchronize());
    std::vector<long long> st(32);
    CK(hipMemcpy(st.data(), d_patch, 32 * 8, hipMemcpyDeviceToHost));
    const char *names[] = {"gather bits, fill", "axis 0", "barrier", "transposition + gain", "-", "axis 1"};
    for (int w = 0; w < 4; ++w) {
        printf("wave %d:", w);
        for (int k = 0; k < 6; ++k) printf("  %s %lld", names[k], st[8 * w + k + 1] - st[8 * w + k]);
        printf("  (clock64 ticks)\n");
    }
    CK(hipGetLastError());
    return 0;
}
