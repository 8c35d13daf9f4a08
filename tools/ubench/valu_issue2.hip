// Micro-benchmark (analysis only): per-instruction issue cost of selected gfx950 VALU/SALU ops, inline asm,
// 8 waves per SIMD resident, 64 independent-ish instructions per loop trip.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)

template <int MODE>
__global__ __launch_bounds__(1024) void k(float *out, int iters) {
    float a = threadIdx.x * 1.0f, b = 2.0f, c = 3.0f, d = 0.5f;
    int ia = threadIdx.x, ib = 77, ic = 5;
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) asm volatile(REP64("v_add_f32 %0, %0, %1\n") : "+v"(a) : "v"(b));
        if (MODE == 1) asm volatile(REP64("v_cndmask_b32 %0, %0, %1, vcc\n") : "+v"(a) : "v"(b) : "vcc");
        if (MODE == 2) asm volatile(REP64("v_bfi_b32 %0, %1, %0, %2\n") : "+v"(ia) : "v"(ib), "v"(ic));
        if (MODE == 3) asm volatile(REP64("v_cmp_lt_f32 vcc, %0, %1\n") : : "v"(a), "v"(b) : "vcc");
        if (MODE == 4) asm volatile(REP16("v_cmp_lt_f32 vcc, %0, %1\n s_nop 1\n v_cndmask_b32 %0, %0, %2, vcc\n v_cndmask_b32 %2, %2, %1, vcc\n v_cndmask_b32 %3, %3, %1, vcc\n") : "+v"(a) : "v"(b), "v"(c), "v"(d) : "vcc");
        if (MODE == 5) asm volatile(REP64("v_min_f32 %0, %0, %1\n") : "+v"(a) : "v"(b));
        if (MODE == 6) asm volatile(REP64("v_mul_lo_u32 %0, %0, %1\n") : "+v"(ia) : "v"(ib));
        if (MODE == 7) asm volatile(REP64("v_mul_i32_i24 %0, %0, %1\n") : "+v"(ia) : "v"(ib));
        if (MODE == 8) asm volatile(REP64("v_cvt_f32_i32 %0, %1\n") : "+v"(a) : "v"(ib));
        if (MODE == 9) asm volatile(REP64("v_floor_f32 %0, %0\n") : "+v"(a));
        if (MODE == 10) asm volatile(REP64("v_add3_u32 %0, %0, %1, %2\n") : "+v"(ia) : "v"(ib), "v"(ic));
        if (MODE == 11) asm volatile(REP64("v_ashrrev_i32 %0, 31, %0\n") : "+v"(ia));
        if (MODE == 12) asm volatile(REP64("s_add_u32 s20, s20, 1\n") : : : "s20", "scc");
        if (MODE == 13) asm volatile(REP16("v_add_f32 %0, %0, %1\n s_add_u32 s20, s20, 1\n v_add_f32 %0, %0, %1\n s_add_u32 s20, s20, 1\n v_add_f32 %0, %0, %1\n s_add_u32 s20, s20, 1\n v_add_f32 %0, %0, %1\n s_add_u32 s20, s20, 1\n") : "+v"(a) : "v"(b) : "s20", "scc");
        if (MODE == 14) asm volatile(REP16("v_cmp_lt_f32 vcc, %0, %1\n s_and_saveexec_b64 s[20:21], vcc\n v_add_f32 %0, %0, %1\n s_or_b64 exec, exec, s[20:21]\n") : "+v"(a) : "v"(b) : "vcc", "s20", "s21", "scc");
        if (MODE == 15) asm volatile(REP64("v_rcp_f32 %0, %0\n") : "+v"(a));
        if (MODE == 16) asm volatile(REP64("v_and_b32 %0, %0, %1\n") : "+v"(ia) : "v"(ib));
        if (MODE == 17) asm volatile(REP64("v_cmp_lt_f32 s[20:21], %0, %1\n") : : "v"(a), "v"(b) : "s20", "s21");
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + ia + ib;
}

int main() {
    float *out; hipMalloc(&out, 512 * 1024 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 500, blocks = 512;
    const char *names[] = {"v_add_f32", "v_cndmask (vcc fixed)", "v_bfi_b32", "v_cmp_lt_f32 -> vcc", "cmp+nop+3 cndmask (per instr of 5)", "v_min_f32", "v_mul_lo_u32", "v_mul_i32_i24", "v_cvt_f32_i32", "v_floor_f32", "v_add3_u32", "v_ashrrev_i32", "s_add_u32", "valu+salu interleaved (per pair)", "cmp+saveexec+add+or (per 4)", "v_rcp_f32", "v_and_b32", "v_cmp -> sgpr pair"};
    const double per[] = {64, 64, 64, 64, 16, 64, 64, 64, 64, 64, 64, 64, 64, 64, 16, 64, 64, 64};
#define RUN(M) { k<M><<<blocks, 1024>>>(out, 5); hipEventRecord(e0); k<M><<<blocks, 1024>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); \
    double n = per[M] * iters * (blocks * 16.0 / 1024.0); printf("%-40s %8.3f ms -> %6.2f cycles per unit per SIMD @2.4GHz\n", names[M], ms, ms * 1e-3 * 2.4e9 / n); }
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9) RUN(10) RUN(11) RUN(12) RUN(13) RUN(14) RUN(15) RUN(16) RUN(17)
    return 0;
}
