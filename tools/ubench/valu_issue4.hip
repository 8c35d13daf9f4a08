// Micro-benchmark (analysis only): issue cost on gfx950 (cycles per wave64 instruction per SIMD, 8 waves resident) of
// every vector instruction the grid-traversal trip uses or could use instead.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)
#define OPS(X) \
    X(0, "v_add_f32", "v_add_f32 %0, %0, %2\n") \
    X(1, "v_fma_f32", "v_fma_f32 %0, %0, %2, %3\n") \
    X(2, "v_fma_mix_f32 (f16 lo, f32, f32)", "v_fma_mix_f32 %0, %2, %3, %0 op_sel_hi:[1,0,0]\n") \
    X(3, "v_fma_mix_f32 (f16 hi, f32, f32)", "v_fma_mix_f32 %0, %2, %3, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n") \
    X(4, "v_add_u32_sdwa sext byte0", "v_add_u32_sdwa %0, sext(%2), %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD\n") \
    X(5, "v_cvt_f32_i32", "v_cvt_f32_i32 %0, %0\n") \
    X(6, "v_cvt_f32_ubyte0", "v_cvt_f32_ubyte0 %0, %0\n") \
    X(7, "v_cvt_f32_f16", "v_cvt_f32_f16 %0, %0\n") \
    X(8, "v_pk_add_f32", "v_pk_add_f32 %1, %1, %4\n") \
    X(9, "v_pk_mul_f32", "v_pk_mul_f32 %1, %1, %4\n") \
    X(10, "v_pk_fma_f32", "v_pk_fma_f32 %1, %1, %4, %4\n") \
    X(11, "v_fract_f32", "v_fract_f32 %0, %0\n") \
    X(12, "v_floor_f32", "v_floor_f32 %0, %0\n") \
    X(13, "v_cvt_flr_i32_f32", "v_cvt_flr_i32_f32 %0, %0\n") \
    X(14, "v_cmp_lt_f32 e64 -> sgpr pair", "v_cmp_lt_f32 s[20:21], %0, %2\n") \
    X(15, "v_cmp_lt_f32 e32 -> vcc", "v_cmp_lt_f32 vcc, %0, %2\n") \
    X(16, "v_cmp_eq_u32_sdwa byte0", "v_cmp_eq_u32_sdwa s[20:21], %0, %2 src0_sel:BYTE_0 src1_sel:DWORD\n") \
    X(17, "v_cndmask_b32 e64", "v_cndmask_b32 %0, %0, %2, s[22:23]\n") \
    X(18, "v_min_f32", "v_min_f32 %0, %0, %2\n") \
    X(19, "v_lshl_add_u32", "v_lshl_add_u32 %0, %0, 1, %2\n") \
    X(20, "v_mad_u32_u24", "v_mad_u32_u24 %0, %0, %2, %3\n") \
    X(21, "v_ashrrev_i32", "v_ashrrev_i32 %0, 31, %0\n") \
    X(22, "v_bfi_b32", "v_bfi_b32 %0, %0, %2, %3\n") \
    X(23, "v_mul_f32 e64 |a| |b|", "v_mul_f32 %0, |%0|, |%2|\n") \
    X(24, "v_add_f32 e64 sgpr", "v_add_f32 %0, s20, %0\n") \
    X(25, "v_mov_b32", "v_mov_b32 %0, %2\n") \
    X(26, "v_cvt_u32_f32", "v_cvt_u32_f32 %0, %0\n") \
    X(27, "v_mul_i32_i24", "v_mul_i32_i24 %0, %0, %2\n") \
    X(28, "v_fmac_f32", "v_fmac_f32 %0, %2, %3\n") \
    X(29, "v_rndne_f32", "v_rndne_f32 %0, %0\n") \
    X(30, "v_med3_u32", "v_med3_u32 %0, %0, %2, %3\n") \
    X(31, "v_and_or_b32", "v_and_or_b32 %0, %0, %2, %3\n") \
    X(32, "v_and_b32", "v_and_b32 %0, %0, %2\n") \
    X(33, "v_sub_f32 sgpr src0", "v_sub_f32 %0, s20, %0\n") \
    X(34, "v_add_u32", "v_add_u32 %0, %0, %2\n") \
    X(35, "v_or_b32_sdwa byte0", "v_or_b32_sdwa %0, %0, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD\n") \
    X(36, "v_cvt_f32_fp8 (byte0)", "v_cvt_f32_fp8 %0, %0\n") \
    X(37, "v_perm_b32", "v_perm_b32 %0, %0, %2, %3\n") \
    X(38, "v_add3_u32", "v_add3_u32 %0, %0, %2, %3\n") \
    X(39, "v_trunc_f32", "v_trunc_f32 %0, %0\n") \
    X(40, "v_mul_f32", "v_mul_f32 %0, %0, %2\n") \
    X(41, "v_lshlrev_b32", "v_lshlrev_b32 %0, 1, %0\n") \
    X(42, "v_mad_u32_u16", "v_mad_u32_u16 %0, %0, %2, %3\n") \
    X(43, "v_max_f32", "v_max_f32 %0, %0, %2\n") \
    X(44, "v_xad_u32", "v_xad_u32 %0, %0, %2, %3\n") \
    X(45, "v_add_f32 + s_nop 0 (pairs)", "v_add_f32 %0, %0, %2\n s_nop 0\n") \
    X(46, "v_readfirstlane_b32", "v_readfirstlane_b32 s20, %0\n") \
    X(47, "v_mbcnt_lo_u32_b32", "v_mbcnt_lo_u32_b32 %0, %2, %0\n") \
    X(48, "v_cmp_lt_f32 vcc, sgpr, v", "v_cmp_lt_f32 vcc, s20, %0\n") \
    X(49, "v_mad_u32_u24 v, sgpr, v", "v_mad_u32_u24 %0, %0, s20, %2\n") \
    X(50, "v_mul_f32 literal", "v_mul_f32 %0, 0x3d888889, %0\n") \
    X(51, "v_add_f32 inline 1.0", "v_add_f32 %0, 1.0, %0\n") \
    X(52, "v_lshrrev_b32 const 8", "v_lshrrev_b32 %0, 8, %0\n") \
    X(53, "v_lshlrev_b32 vgpr shift", "v_lshlrev_b32 %0, %2, %0\n") \
    X(54, "v_lshrrev_b32 vgpr shift", "v_lshrrev_b32 %0, %2, %0\n") \
    X(55, "v_and_b32 sgpr", "v_and_b32 %0, s20, %0\n") \
    X(56, "v_or_b32 literal", "v_or_b32 %0, 0x4b000000, %0\n") \
    X(57, "v_sub_u32", "v_sub_u32 %0, %0, %2\n") \
    X(58, "v_xor_b32", "v_xor_b32 %0, %0, %2\n") \
    X(59, "v_cndmask_b32 e32 vcc (cmp hoisted)", "v_cndmask_b32 %0, %0, %2, vcc\n") \
    X(60, "v_mul_f32 -a (neg modifier, e64)", "v_mul_f32 %0, -%0, %2\n") \
    X(61, "v_sub_f32", "v_sub_f32 %0, %0, %2\n") \
    X(62, "v_subrev_f32", "v_subrev_f32 %0, %0, %2\n") \
    X(63, "v_bfe_u32 const", "v_bfe_u32 %0, %0, 8, 8\n") \
    X(64, "v_mul_lo_u32", "v_mul_lo_u32 %0, %0, %2\n") \
    X(65, "v_add_f32 dpp row_shr:1", "v_add_f32_dpp %0, %0, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n") \
    X(66, "ds_bpermute_b32 + wait", "ds_bpermute_b32 %0, %2, %0\n s_waitcnt lgkmcnt(0)\n") \
    X(67, "v_exp_f32", "v_exp_f32 %0, %0\n")
#define COUNT 68
template <int OP>
__global__ __launch_bounds__(1024) void k(float *out, int iters) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    float a = threadIdx.x * 1.0f, b = 2.0f, c = 3.0f;
    f2 p = {a, b}, q = {c, a};
    for (int i = 0; i < iters; ++i) {
#define X(ID, NAME, TEXT) if (OP == ID) asm volatile(REP64(TEXT) : "+v"(a), "+v"(p) : "v"(b), "v"(c), "v"(q) : "vcc", "s20", "s21");
        OPS(X)
#undef X
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + p.x + p.y;
}
template <int OP> void run(float *out, const char *name, hipEvent_t e0, hipEvent_t e1) {
    const int iters = 300, blocks = 512;
    k<OP><<<blocks, 1024>>>(out, 5);
    hipEventRecord(e0);
    k<OP><<<blocks, 1024>>>(out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double n = 64.0 * iters * (blocks * 16.0 / 1024.0);
    printf("%-36s %8.3f ms -> %5.2f cycles per instruction per SIMD @2.4GHz\n", name, ms, ms * 1e-3 * 2.4e9 / n);
}
template <int OP> struct Runner {
    static void go(float *out, const char *const *names, hipEvent_t e0, hipEvent_t e1) {
        Runner<OP - 1>::go(out, names, e0, e1);
        run<OP>(out, names[OP], e0, e1);
    }
};
template <> struct Runner<-1> { static void go(float *, const char *const *, hipEvent_t, hipEvent_t) {} };
int main() {
    float *out;
    hipMalloc(&out, 512 * 1024 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    static const char *names[COUNT] = {
#define X(ID, NAME, TEXT) NAME,
        OPS(X)
#undef X
    };
    Runner<COUNT - 1>::go(out, names, e0, e1);
    return 0;
}
