// Micro-benchmark (analysis only): do full-rate and half-rate vector instructions of a mix add their issue costs, or do
// the half-rate ones overlap with full-rate issue?  Groups of independent instructions, 8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP32(x) REP16(x) REP16(x)
#define A "v_add_f32 %0, %0, %4\n"
#define A2 "v_add_f32 %1, %1, %4\n"
#define A3 "v_mul_f32 %2, %2, %4\n"
#define H "v_cvt_flr_i32_f32 %3, %3\n"
#define H2 "v_cndmask_b32_e64 %3, %3, %4, s[20:21]\n"
#define H3 "v_lshl_add_u32 %3, %3, 1, %4\n"
template <int M>
__global__ __launch_bounds__(1024) void k(float *out, int iters) {
    float a = threadIdx.x, b = 1.0f, c = 2.0f, d = 3.0f, e = 0.5f;
    for (int i = 0; i < iters; ++i) {
        if (M == 0) asm volatile(REP32(A) : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e) : "s20", "s21");
        if (M == 1) asm volatile(REP32(H) : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e) : "s20", "s21");
        if (M == 2) asm volatile(REP32(A H) : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e) : "s20", "s21");
        if (M == 3) asm volatile(REP32(A A2 H) : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e) : "s20", "s21");
        if (M == 4) asm volatile(REP32(A A2 A3 H) : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e) : "s20", "s21");
        if (M == 5) asm volatile(REP32(A H H2) : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e) : "s20", "s21");
        if (M == 6) asm volatile(REP32(A A2 H H2) : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e) : "s20", "s21");
        if (M == 7) asm volatile(REP32(A A2 A3 H H3) : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e) : "s20", "s21");
        if (M == 8) asm volatile(REP32(A A2 A3 A A2 A3 H) : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e) : "s20", "s21");
        if (M == 9) asm volatile(REP32(H H2 H3) : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e) : "s20", "s21");
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d;
}
template <int M> void run(float *out, const char *name, double full, double half, hipEvent_t e0, hipEvent_t e1) {
    const int iters = 500, blocks = 512;
    k<M><<<blocks, 1024>>>(out, 5);
    hipEventRecord(e0);
    k<M><<<blocks, 1024>>>(out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double groups = 32.0 * iters * (blocks * 16.0 / 1024.0);
    printf("%-28s %8.3f ms -> %6.2f cycles per group per SIMD @2.4GHz (additive model: %5.2f)\n", name, ms, ms * 1e-3 * 2.4e9 / groups, full * 2.35 + half * 4.35);
}
int main() {
    float *out;
    hipMalloc(&out, 512 * 1024 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    run<0>(out, "1 full", 1, 0, e0, e1);
    run<1>(out, "1 half", 0, 1, e0, e1);
    run<2>(out, "1 full + 1 half", 1, 1, e0, e1);
    run<3>(out, "2 full + 1 half", 2, 1, e0, e1);
    run<4>(out, "3 full + 1 half", 3, 1, e0, e1);
    run<5>(out, "1 full + 2 half", 1, 2, e0, e1);
    run<6>(out, "2 full + 2 half", 2, 2, e0, e1);
    run<7>(out, "3 full + 2 half", 3, 2, e0, e1);
    run<8>(out, "6 full + 1 half", 6, 1, e0, e1);
    run<9>(out, "3 half (cvt, cndmask, lshl_add)", 0, 3, e0, e1);
    return 0;
}
