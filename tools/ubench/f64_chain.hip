// Dependent binary64 chain v = f + z * v (one multiply, one add, as the exact render's prefilter recursions) with 1, 2, 4 waves per
// SIMD and 1, 2, 4 independent chains per wave: cycles per step of a chain.  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CHAINS>
__global__ void chain_kernel(double *out, int steps, double z, double f) {
    __shared__ double hog[12000];                                      // 96 KB: one workgroup per CU
    double v[CHAINS];
    for (int c = 0; c < CHAINS; ++c) v[c] = (double)threadIdx.x + c;
    for (int i = 0; i < steps; ++i) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) {
            double m = v[c] * z;
            asm volatile("" : "+v"(m));
            v[c] = f + m;
            asm volatile("" : "+v"(v[c]));
        }
    }
    double s = 0;
    for (int c = 0; c < CHAINS; ++c) s += v[c];
    if (s == 12345.678) hog[threadIdx.x] = s;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + hog[0] * 0.0;
}
template <int CHAINS>
void run(int waves_per_simd, double *out) {
    const int steps = 200000;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    chain_kernel<CHAINS><<<256, 256 * waves_per_simd>>>(out, 1000, -0.26, 0.5);
    hipEventRecord(a);
    chain_kernel<CHAINS><<<256, 256 * waves_per_simd>>>(out, steps, -0.26, 0.5);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("waves per SIMD %d, chains per wave %d: %.1f cycles per step of a chain at 2.4 GHz (%.1f per SIMD per step)\n", waves_per_simd, CHAINS,
           ms * 1e-3 * 2.4e9 / steps, ms * 1e-3 * 2.4e9 / steps / (CHAINS * waves_per_simd));
}
int main() {
    double *out; hipMalloc(&out, 256 * 1024 * 8);
    for (int w = 1; w <= 4; w *= 2) { run<1>(w, out); run<2>(w, out); run<4>(w, out); }
    return 0;
}
