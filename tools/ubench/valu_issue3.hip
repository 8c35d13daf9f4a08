// Micro-benchmark (analysis only): v_cndmask_b32 variants on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)
template <int MODE>
__global__ __launch_bounds__(1024) void k(float *out, int iters) {
    float a = threadIdx.x * 1.0f, b = 2.0f, c = 3.0f, d = 0.5f, e = 1.5f;
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) asm volatile("v_cmp_lt_f32 vcc, %1, %2\n s_nop 4\n" REP64("v_cndmask_b32 %0, %0, %1, vcc\n") : "+v"(a) : "v"(b), "v"(c) : "vcc");
        if (MODE == 1) asm volatile("v_cmp_lt_f32 s[20:21], %1, %2\n s_nop 4\n" REP64("v_cndmask_b32_e64 %0, %0, %1, s[20:21]\n") : "+v"(a) : "v"(b), "v"(c) : "s20", "s21");
        if (MODE == 2) asm volatile("v_cmp_lt_f32 vcc, %4, %5\n s_nop 4\n" REP16("v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n") : "+v"(a), "+v"(c), "+v"(d), "+v"(e) : "v"(b), "v"(b) : "vcc");
        if (MODE == 3) asm volatile(REP16("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc\n v_add_f32 %2, %2, %1\n v_add_f32 %3, %3, %1\n") : "+v"(a) : "v"(b), "v"(c), "v"(d) : "vcc");
        if (MODE == 4) asm volatile(REP64("v_max_i32 %0, %0, %1\n") : "+v"(a) : "v"(b));
        if (MODE == 5) asm volatile(REP64("v_sub_f32 %0, %0, %1\n") : "+v"(a) : "v"(b));
        if (MODE == 6) asm volatile(REP64("v_mul_f32 %0, %0, %1\n") : "+v"(a) : "v"(b));
        if (MODE == 7) asm volatile(REP64("v_xor_b32 %0, %0, %1\n") : "+v"(a) : "v"(b));
        if (MODE == 8) asm volatile(REP64("v_lshrrev_b32 %0, %1, %0\n") : "+v"(a) : "v"(b));
        if (MODE == 9) asm volatile(REP64("v_add_u32 %0, %0, %1\n") : "+v"(a) : "v"(b));
        if (MODE == 10) asm volatile(REP64("v_fma_f32 %0, %0, %1, %2\n") : "+v"(a) : "v"(b), "v"(c));
        if (MODE == 11) asm volatile(REP64("v_mad_u32_u24 %0, %0, %1, %2\n") : "+v"(a) : "v"(b), "v"(c));
        if (MODE == 12) asm volatile(REP64("v_lshl_add_u32 %0, %0, 2, %1\n") : "+v"(a) : "v"(b));
        if (MODE == 13) asm volatile(REP64("v_bfe_u32 %0, %0, %1, 1\n") : "+v"(a) : "v"(b));
        if (MODE == 14) asm volatile(REP64("v_and_or_b32 %0, %0, %1, %2\n") : "+v"(a) : "v"(b), "v"(c));
        if (MODE == 15) asm volatile(REP64("v_max_f32 %0, %0, %1\n") : "+v"(a) : "v"(b));
        if (MODE == 16) asm volatile(REP64("v_cvt_i32_f32 %0, %0\n") : "+v"(a));
        if (MODE == 17) asm volatile(REP64("v_sub_u32 %0, %0, %1\n") : "+v"(a) : "v"(b));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + e;
}
int main() {
    float *out; hipMalloc(&out, 512 * 1024 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 500, blocks = 512;
    const char *names[] = {"cndmask e32 vcc (dep chain)", "cndmask e64 sgpr (dep chain)", "cndmask e32 vcc (4 indep chains)", "cmp+cndmask+2 add (per 4)", "v_max_i32", "v_sub_f32", "v_mul_f32", "v_xor_b32", "v_lshrrev_b32", "v_add_u32", "v_fma_f32", "v_mad_u32_u24", "v_lshl_add_u32", "v_bfe_u32", "v_and_or_b32", "v_max_f32", "v_cvt_i32_f32", "v_sub_u32"};
    const double per[] = {64, 64, 64, 16, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64};
#define RUN(M) { k<M><<<blocks, 1024>>>(out, 5); hipEventRecord(e0); k<M><<<blocks, 1024>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); \
    double n = per[M] * iters * (blocks * 16.0 / 1024.0); printf("%-40s %8.3f ms -> %6.2f cycles per unit per SIMD @2.4GHz\n", names[M], ms, ms * 1e-3 * 2.4e9 / n); }
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9) RUN(10) RUN(11) RUN(12) RUN(13) RUN(14) RUN(15) RUN(16) RUN(17)
    return 0;
}
