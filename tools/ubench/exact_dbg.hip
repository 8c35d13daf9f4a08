// debug harness of racecar_patch_exact.h: one car on a synthetic drivable bitmap; prints what each stage produced
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../racing_dreamer_amd/csrc/racecar_patch_exact.h"
#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(_e), __LINE__); return 1; } } while (0)
int main() {
    const int h = 400, w = 400, pitch = (w + 31) / 32;
    std::vector<uint32_t> drv((size_t)h * pitch, 0u);
    for (int y = 100; y < 300; ++y) for (int x = 150; x < 260; ++x) drv[(size_t)y * pitch + (x >> 5)] |= 1u << (x & 31);
    float hx = 10.0f, hy = 10.0f, hth = 0.3f; uint8_t hf = 0;
    uint32_t *d_drv; float *d_x, *d_y, *d_th; uint8_t *d_f, *d_patch; double *d_s; int32_t *d_kk;
    CK(hipMalloc(&d_drv, drv.size() * 4)); CK(hipMemcpy(d_drv, drv.data(), drv.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_x, 4)); CK(hipMalloc(&d_y, 4)); CK(hipMalloc(&d_th, 4)); CK(hipMalloc(&d_f, 1)); CK(hipMalloc(&d_patch, 4096));
    CK(hipMemcpy(d_x, &hx, 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_y, &hy, 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_th, &hth, 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_f, &hf, 1, hipMemcpyHostToDevice)); CK(hipMemset(d_patch, 7, 4096));
    CK(hipMalloc(&d_s, (size_t)RC_EXACT_CAR_DOUBLES * 8)); CK(hipMemset(d_s, 0, (size_t)RC_EXACT_CAR_DOUBLES * 8));
    std::vector<int32_t> kk(RC_EXACT_TABLE_INTS, 0);
    for (int xx = 0; xx < 64; ++xx) { for (int k = 0; k < 7; ++k) kk[xx * 15 + k] = (1 << 22) / 7; kk[64 * 15 + 2 * xx] = xx * 3; kk[64 * 15 + 2 * xx + 1] = 7; }
    CK(hipMalloc(&d_kk, kk.size() * 4)); CK(hipMemcpy(d_kk, kk.data(), kk.size() * 4, hipMemcpyHostToDevice));
    RcExactParams p{};
    p.drv_words = d_drv; p.pitch = pitch; p.h = h; p.w = w; p.x = d_x; p.y = d_y; p.theta = d_th; p.fresh = d_f;
    p.fh = 400; p.r_top = 399; p.c0 = 0; p.ox = 0.0; p.oy = 0.0; p.res = 0.05; p.scratch = d_s; p.patch = d_patch; p.kk = d_kk; p.car0 = 0; p.n_cars = 1;
    hipLaunchKernelGGL(rc_patch_exact_prefilter_kernel, dim3(1), dim3(256), 0, 0, p);
    CK(hipGetLastError()); CK(hipDeviceSynchronize());
    std::vector<double> s(RC_EXACT_CAR_DOUBLES);
    CK(hipMemcpy(s.data(), d_s, s.size() * 8, hipMemcpyDeviceToHost));
    double mx = 0; for (int i = 0; i < 220 * 220; ++i) mx = fmax(mx, fabs(s[i]));
    printf("prefilter: max |coefficient| %g  coefficient[110][110] %.17g\n", mx, s[110 * 220 + 110]);
    hipLaunchKernelGGL(rc_patch_exact_sample_kernel<false>, dim3(1), dim3(PX_ST), 0, 0, p);
    CK(hipGetLastError()); CK(hipDeviceSynchronize());
    std::vector<uint8_t> patch(4096);
    CK(hipMemcpy(patch.data(), d_patch, 4096, hipMemcpyDeviceToHost));
    int ones = 0, sevens = 0; for (uint8_t v : patch) { ones += v == 1; sevens += v == 7; }
    printf("patch: ones %d untouched %d first bytes %d %d %d\n", ones, sevens, patch[0], patch[2048 + 32], patch[4095]);
    return 0;
}
