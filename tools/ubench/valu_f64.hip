// Micro-benchmark (analysis only): issue cost on gfx950 of the binary64 vector instructions the exact render is made of
// (cycles per wave64 instruction per SIMD, 8 waves resident per SIMD, every wave a dependent chain - so with 8 waves in turn
// a SIMD issues back to back unless the instruction's own latency exceeds 8 issue slots).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_f64.hip -o tools/ubench/valu_f64 && tools/ubench/valu_f64
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)
template <int OP>
__global__ __launch_bounds__(256) void k(double *out, double a, double b, int iters) {
    double v = a + threadIdx.x * 1e-3, w = b;
    for (int i = 0; i < iters; ++i) {
        if (OP == 0) { REP64(asm volatile("v_add_f64 %0, %0, %1" : "+v"(v) : "v"(w));) }
        if (OP == 1) { REP64(asm volatile("v_mul_f64 %0, %0, %1" : "+v"(v) : "v"(w));) }
        if (OP == 2) { REP64(asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(v) : "v"(w));) }
        if (OP == 3) { REP64(asm volatile("v_add_f32 %0, %0, %1" : "+v"(*(float *)&v) : "v"(*(float *)&w));) }
        if (OP == 4) { REP64(asm volatile("v_floor_f64 %0, %0" : "+v"(v));) }
        if (OP == 5) { REP64(asm volatile("v_rcp_f64 %0, %0" : "+v"(v));) }
        if (OP == 6) { REP64(asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(*(int *)&w) : "v"(v));) }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = v + w;
}
template <int OP>
void run(const char *name) {
    double *out; hipMalloc(&out, 256 * 8 * 256 * 8);
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(256 * 8), dim3(256), 0, 0, out, 1.0, 1.0000001, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(256 * 8), dim3(256), 0, 0, out, 1.0, 1.0000001, iters);      // 8 workgroups of 4 waves per CU = 8 waves per SIMD
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double insts_per_simd = 8.0 * iters * 64;                   // 8 waves x iters x 64 instructions
    printf("%-14s %.2f cycles per wave64 instruction (at 2.4 GHz; %.3f ms)\n", name, ms * 1e-3 * 2.4e9 / insts_per_simd, ms);
    hipFree(out);
}
int main() {
    run<3>("v_add_f32"); run<0>("v_add_f64"); run<1>("v_mul_f64"); run<2>("v_fma_f64"); run<4>("v_floor_f64"); run<5>("v_rcp_f64"); run<6>("v_cvt_i32_f64");
    return 0;
}
