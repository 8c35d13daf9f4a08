// Micro-benchmark (analysis only): does gfx950 issue a wave64 vector instruction faster when one 32-lane half of EXEC
// (or more) is empty?  A wave64 instruction takes two passes of 32 lanes; if the hardware skipped an empty pass, a
// divergent loop whose surviving lanes sit in one half would run at twice the rate.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)
template <int OP>
__global__ __launch_bounds__(1024) void k(float *out, int iters, unsigned long long mask) {
    float a = threadIdx.x * 1.0f, b = 2.0f, c = 3.0f;
    const int lane = threadIdx.x & 63;
    if ((mask >> lane) & 1) {
        for (int i = 0; i < iters; ++i) {
            if (OP == 0) asm volatile(REP64("v_add_f32 %0, %0, %1\n") : "+v"(a) : "v"(b));
            if (OP == 1) asm volatile(REP64("v_cvt_i32_f32 %0, %0\n") : "+v"(a));
            if (OP == 2) asm volatile(REP64("v_fma_f32 %0, %0, %1, %2\n") : "+v"(a) : "v"(b), "v"(c));
            if (OP == 3) asm volatile(REP64("v_rcp_f32 %0, %0\n") : "+v"(a));
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c;
}
int main() {
    float *out; hipMalloc(&out, 512 * 1024 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 500, blocks = 512;
    const char *ops[] = {"v_add_f32", "v_cvt_i32_f32", "v_fma_f32", "v_rcp_f32"};
    const unsigned long long masks[] = {~0ull, 0xffffffffull, 0xffffffff00000000ull, 0x5555555555555555ull, 0xffffull, 1ull,
                                        0x0000ffff0000ffffull, 0xffff0000ffffull << 8};
    const char *mnames[] = {"all 64", "lanes 0-31", "lanes 32-63", "even lanes", "lanes 0-15", "lane 0", "0-15 + 32-47", "8-23 + 40-55"};
#define RUN(OP) for (int m = 0; m < 8; ++m) { k<OP><<<blocks, 1024>>>(out, 5, masks[m]); hipEventRecord(e0); k<OP><<<blocks, 1024>>>(out, iters, masks[m]); \
    hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); \
    double n = 64.0 * iters * (blocks * 16.0 / 1024.0); printf("%-14s %-14s %8.3f ms -> %5.2f cycles per instruction per SIMD @2.4GHz\n", ops[OP], mnames[m], ms, ms * 1e-3 * 2.4e9 / n); }
    RUN(0) RUN(1) RUN(2) RUN(3)
    return 0;
}
