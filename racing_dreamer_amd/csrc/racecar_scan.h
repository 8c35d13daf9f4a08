// The LiDAR scan's device code (H3): the exact grid traversal over certified-free rectangles and the one-wave-per-car scan
// built on it.  A header, because two translation units instantiate it: racecar_kernels.hip - the shipped kernels - and
// racecar_lab.hip, the lab library (superseded scan variants 0-6 that the parity tests cross-check, the instrumented
// "stamps" build), which is compiled only when tools/ or those tests ask for it and is not part of libracecar_hip.so.
// Everything here sits in an anonymous namespace: each translation unit has its own copy.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <math.h>
#include <cstdlib>
#include <type_traits>

#include "racecar_device.h"
#include "racecar_internal.h"

#define RC_N_BEAMS 1080

namespace {

using rcd::clampf;
using rcd::sincos32;

__device__ __forceinline__ int bit_at(const uint32_t *words, int pitch, int ix, int iy) {
    return (words[iy * pitch + (ix >> 5)] >> (ix & 31)) & 1u;
}

// ------------------------------------------------------------------------------------------------
// Stage a [h][pitch] bitmap into LDS with 16-byte loads (the buffers are padded to 4 words).
__device__ __forceinline__ void stage_bitmap(uint32_t *lds, const uint32_t *__restrict__ src, int nwords) {
    const int nvec = (nwords + 3) >> 2;
    const uint4 *s4 = reinterpret_cast<const uint4 *>(src);
    uint4 *d4 = reinterpret_cast<uint4 *>(lds);
    for (int i = threadIdx.x; i < nvec; i += blockDim.x) d4[i] = s4[i];
    __syncthreads();
}

// Distance [m] along the ray to another car's rectangle, +inf if none within range (H18).
__device__ __forceinline__ float ray_vs_car(float lx, float ly, float dx, float dy, float ox, float oy, float ct2,
                                            float st2) {
    const float cx = ox + RCS_BOX_CX * ct2, cy = oy + RCS_BOX_CX * st2;
    const float rx = lx - cx, ry = ly - cy;
    // Conservative early out (most rays, usually whole waves): the rectangle lies inside the circle of radius
    // 0.3133 m around its centre, so a ray whose line passes that centre by more than 0.32 m, or whose centre
    // projection is more than 0.32 m behind the sensor or beyond range, cannot produce a return in the exact
    // slab test below (margins are 1000x the fp32 rounding of these quantities).
    const float along = -(rx * dx + ry * dy);
    if (fabsf(rx * dy - ry * dx) > 0.32f || along < -0.32f || along > RCS_MAX_RANGE + 0.32f) return INFINITY;
    const float px = rx * ct2 + ry * st2;
    const float py = ry * ct2 - rx * st2;
    const float ex = dx * ct2 + dy * st2;
    const float ey = dy * ct2 - dx * st2;
    float tn = -INFINITY, tf = INFINITY;
    bool miss = false;
    if (ex != 0.0f) {
        const float inv = 1.0f / ex;
        const float t1 = (-RCS_BOX_HL - px) * inv, t2 = (RCS_BOX_HL - px) * inv;
        const float lo = t1 < t2 ? t1 : t2, hi = t1 < t2 ? t2 : t1;
        tn = lo > tn ? lo : tn;
        tf = hi < tf ? hi : tf;
    } else {
        miss |= fabsf(px) > RCS_BOX_HL;
    }
    if (ey != 0.0f) {
        const float inv = 1.0f / ey;
        const float t1 = (-RCS_BOX_HW - py) * inv, t2 = (RCS_BOX_HW - py) * inv;
        const float lo = t1 < t2 ? t1 : t2, hi = t1 < t2 ? t2 : t1;
        tn = lo > tn ? lo : tn;
        tf = hi < tf ? hi : tf;
    } else {
        miss |= fabsf(py) > RCS_BOX_HW;
    }
    const bool hit = !miss && tn <= tf && tf >= 0.0f;
    const float tt = tn > 0.0f ? tn : 0.0f;
    return (hit && tt < RCS_MAX_RANGE) ? tt : INFINITY;
}

__device__ __forceinline__ int sign_mask(float a) {            // -1 if the sign bit is set, else 0
    int r;
    asm("v_ashrrev_i32 %0, 31, %1" : "=v"(r) : "v"(a));
    return r;
}
__device__ __forceinline__ int nonzero_mask(int a) {           // -1 if a != 0 (0 <= a < 2^31), else 0
    int r;
    asm("v_sub_u32 %0, 0, %1\n\tv_ashrrev_i32 %0, 31, %0" : "=v"(r) : "v"(a));
    return r;
}
__device__ __forceinline__ int bfi(int mask, int a, int b) {   // (mask & a) | (~mask & b)
    int r;
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "v"(mask), "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float bfi(int mask, float a, float b) {
    return __int_as_float(bfi(mask, __float_as_int(a), __float_as_int(b)));
}
// cond ? a : b through the e64 form of v_cndmask (mask in an SGPR pair, ~4.3 cycles) instead of the e32 form
// reading VCC (~16 cycles) that hipcc picks when VCC happens to hold the condition.
__device__ __forceinline__ float select64(bool cond, float a, float b) {
    float r;
    const unsigned long long m = __builtin_amdgcn_ballot_w64(cond);
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(b), "v"(a), "s"(m));
    return r;
}

__device__ __forceinline__ int floor_to_int(float a) {         // (int)floorf(a) in one instruction
    int r;
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(r) : "v"(a));
    return r;
}

__device__ __forceinline__ unsigned mad_u24(unsigned a, unsigned b, unsigned c) {   // a * b + c on 24-bit operands
    unsigned r;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(b), "v"(c));
    return r;
}
__device__ __forceinline__ unsigned med3_u32(unsigned a, unsigned lo, unsigned hi) {  // clamp(a, lo, hi) in one instruction
    unsigned r;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(lo), "v"(hi));
    return r;
}
template <int BYTE>
__device__ __forceinline__ int byte_xor(unsigned word, int m) {              // ((word >> 8 BYTE) & 255) ^ m
    int r;
    if (BYTE == 0)
        asm("v_xor_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD" : "=v"(r) : "v"(word), "v"(m));
    else
        asm("v_xor_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "=v"(r) : "v"(word), "v"(m));
    return r;
}

// The two reciprocals 1/dx, 1/dy of a ray (IEEE-correct, as the spec demands; 3e38 stands in for 1/0).
// v_rcp_f32 plus one FMA Newton step is the correctly rounded reciprocal for EVERY fp32 input with
// 2^-100 <= |d| <= 2^100 on gfx950 - verified exhaustively on the device (tools/ubench/rcp_exhaustive.hip:
// 0 mismatches against IEEE division over all 3.37e9 such inputs; the raw instruction alone differs on 10.7 %).
// Anything else (zero, subnormal, huge, NaN) takes the IEEE division; a direction component is <= 1 in magnitude
// and only exactly zero or >= 1e-32 in practice, so that path runs when a beam is exactly axis-parallel.
// 6 + 4 instructions instead of 2 x 11 for the division expansion.
__device__ __forceinline__ void ray_reciprocals(float dx, float dy, float &idx, float &idy) {
    // callers pass components already forced into [-2, 2], so only the lower bound needs a test - and one test of the
    // product serves both: |dx dy| >= 2^-99 with both factors <= 2 puts each at 2^-100 or more (a product below the
    // bound merely takes the slower path, which is correct for every input)
    if (fabsf(dx * dy) >= 0x1p-99f) {
        const float rx = __builtin_amdgcn_rcpf(dx), ry = __builtin_amdgcn_rcpf(dy);
        idx = __builtin_fmaf(__builtin_fmaf(-dx, rx, 1.0f), rx, rx);
        idy = __builtin_fmaf(__builtin_fmaf(-dy, ry, 1.0f), ry, ry);
    } else {
        idx = select64(dx != 0.0f, 1.0f / dx, 3.0e38f);
        idy = select64(dy != 0.0f, 1.0f / dy, 3.0e38f);
    }
}

__device__ __forceinline__ unsigned long long cmp_nlt_f32(float a, float b) {          // lane mask of !(a < b)
    unsigned long long m;
    asm("v_cmp_nlt_f32_e64 %0, %1, %2" : "=s"(m) : "v"(a), "v"(b));
    return m;
}
typedef float v2f __attribute__((ext_vector_type(2)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned long long cmp_lt_f32(float a, float b) {             // lane mask of a < b
    unsigned long long m;
    asm("v_cmp_lt_f32_e64 %0, %1, %2" : "=s"(m) : "v"(a), "v"(b));
    return m;
}
__device__ __forceinline__ uint32_t select_mask_u(unsigned long long m, uint32_t a, uint32_t b) {   // m ? a : b
    uint32_t r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(b), "v"(a), "s"(m));
    return r;
}
__device__ __forceinline__ unsigned long long cmp_nlt_f32_s(float a, float b) {        // the same with a wave-uniform b
    unsigned long long m;
    asm("v_cmp_nlt_f32_e64 %0, %1, %2" : "=s"(m) : "v"(a), "s"(b));
    return m;
}
__device__ __forceinline__ bool cmp_lt_f32_s(float a, float b) {                        // a < b, b wave-uniform
    return a < b;
}
__device__ __forceinline__ unsigned long long cmp_ne_u32(uint32_t a, uint32_t b) {      // lane mask of a != b
    unsigned long long m;
    asm("v_cmp_ne_u32_e64 %0, %1, %2" : "=s"(m) : "v"(a), "v"(b));
    return m;
}
__device__ __forceinline__ float select_mask(unsigned long long m, float a, float b) {  // m ? a : b
    float r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(b), "v"(a), "s"(m));
    return r;
}
template <int BYTE>
__device__ __forceinline__ uint32_t add_ubyte(unsigned word, uint32_t a) {   // a + (uint8)(word >> 8 BYTE), one instruction
    uint32_t r;
    if (BYTE == 0)
        asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD" : "=v"(r) : "v"(word), "v"(a));
    else
        asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "=v"(r) : "v"(word), "v"(a));
    return r;
}
template <int BYTE>
__device__ __forceinline__ int add_sbyte(unsigned word, int a) {             // a + (int8)(word >> 8 BYTE)
    int r;
    if (BYTE == 0)
        asm("v_add_u32_sdwa %0, sext(%1), %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD" : "=v"(r) : "v"(word), "v"(a));
    else
        asm("v_add_u32_sdwa %0, sext(%1), %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "=v"(r) : "v"(word), "v"(a));
    return r;
}
__device__ __forceinline__ float max_with(float a, float lo) {               // IEEE maxNum: a NaN becomes `lo`
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(lo));
    return r;
}
__device__ __forceinline__ float min_with(float a, float hi) {
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(hi));
    return r;
}

// Table entry (uint16 per quadrant plane and cell): byte 0 = width, byte 1 = height (1..255 cells) of a free rectangle
// with the cell at its corner, extending towards the quadrant; a wall is 0x0000 and a cell of the sentinel ring 0x0100,
// so "stop" <=> byte 0 == 0 and "no return" <=> entry != 0 at the stop.
//
// THE MIRRORED FRAME.  The traversal negates every axis on which the ray heads towards -: position g~ = -g, direction
// |d|, reciprocal |1/d|, cell index i~ = floor(-p) = ~i, boundary b~ = -b.  Negation is exact and IEEE arithmetic is
// symmetric under it, so every boundary time fl(fl(b~ - g~) * |1/d|) is bit-identical to the spec's fl(fl(b - g) * (1/d))
// and the comparisons (including "ties go to y") are the same - but in that frame EVERY ray heads towards + on both
// axes: the boundary that leaves a rectangle is i~ + width, the cell behind it has that very index, and a trip holds
// no sign arithmetic.  The quadrant planes are stored mirrored to match (plane q: columns reversed when q & 1, rows
// when q & 2), so a cell's entry sits at plane(q) + 2 (i~x + (q & 1 ? w : 0)) + pitch2 (i~y + (q & 2 ? h : 0)).
//
// Cell indices as float bits.  T = bits(1.5 * 2^23 + i~) = 0x4b400000 + i~ (|i~| < 2^22): adding an integer to T is
// adding it to the float, bits(z + 1.5 * 2^23) is 0x4b400000 + rne(z), and as_float(T) - 1.5 * 2^23 is the index as a
// float, exactly.  The conversions of a trip (int -> float for the two boundaries, float -> int for the new cell)
// become one full-rate add / subtract each; on gfx950 v_cvt_*, v_floor, v_fract, SDWA forms, v_bfi, v_cndmask, compares,
// min / max, every three-operand integer op AND any op that reads an SGPR issue at half the rate of
// v_add / v_sub / v_mul / v_fma_f32, v_add / v_sub_u32, v_and / v_or / v_xor and the right shifts
// (tools/ubench/valu_issue4.hip), and this kernel is bound by exactly that issue rate.
// The start cell's entry for a ray of direction (dx, dy): quadrant from the signs, slope bin from the float bits of
// |dy| * |1/dx| (relative error 1.2e-7 against the bin edges the builder widened by 1e-6; 1/0 is stood in for by
// 3e38, which lands in the steepest bin like every slope above 2^4).  lds_line = LDS address RC_FIRST_BIAS entries
// before the wave's copy of the cell's line (all zeros when the sensor is off the grid: the ray reads 0).
__device__ __forceinline__ unsigned first_trip_entry(uint32_t lds_line, float dy, float idx, int nx, int ny) {
    float slope;
    asm("v_mul_f32_e64 %0, |%1|, |%2|" : "=v"(slope) : "v"(dy), "v"(idx));
    const unsigned bin = med3_u32(__float_as_uint(slope) >> RC_FIRST_SHIFT, RC_FIRST_BIAS, RC_FIRST_BIAS + RC_FIRST_BINS - 1);
    unsigned off, addr;
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(off) : "v"(ny), "v"(4u * RC_FIRST_BINS), "v"((unsigned)nx & (2u * RC_FIRST_BINS)));
    asm("v_lshl_add_u32 %0, %1, 1, %2" : "=v"(addr) : "v"(bin), "v"(off));
    typedef const __attribute__((address_space(3))) uint16_t *lds_u16_ptr;
    return *(lds_u16_ptr)(uintptr_t)(lds_line + addr);
}

// ---- The traversal (variants 6 and 7), in the mirrored frame with cell indices as float bits (see above) ----------
constexpr float kCellMagic = 12582912.0f;               // 1.5 * 2^23
constexpr uint32_t kCellMagicBits = 0x4b400000u;        // its bit pattern: T = kCellMagicBits + i~

// What a trip needs besides the ray.  The per-car kernel holds these in VECTOR registers (pin_vgpr): a full-rate
// vector instruction that reads a scalar register issues at half rate.
struct TravConst {
    float band_mh, res;        // band - 0.5 (see below); metres per cell
    uint32_t kx, ky, c00;      // plane address = 2 Tx + pitch2 (Ty mod 2^24) + c00 + (dx < 0 ? kx : 0) + (dy < 0 ? ky : 0), mod 2^32
};
__device__ __forceinline__ TravConst trav_const(const RcTrackDev &t) {
    const uint32_t pitch2 = 2u * (uint32_t)t.cell_pitch, P = (uint32_t)t.quad_plane_bytes;
    // plane q = 2 (dy < 0) + (dx < 0) starts at q P; a mirrored axis adds the grid's extent to its index (i~ = ~i >= -w);
    // c00 takes the magic out of 2 Tx and pitch2 (Ty mod 2^24) again
    return {t.band_mh, t.res, P + 2u * (uint32_t)t.w, 2u * P + pitch2 * (uint32_t)t.h, 0u - 2u * kCellMagicBits - pitch2 * 0x400000u};
}
__device__ __forceinline__ float pin_vgpr(float x) { float r; asm("v_mov_b32 %0, %1" : "=v"(r) : "s"(x)); return r; }
__device__ __forceinline__ uint32_t pin_vgpr(uint32_t x) { uint32_t r; asm("v_mov_b32 %0, %1" : "=v"(r) : "s"(x)); return r; }
__device__ __forceinline__ int pin_vgpr(int x) { int r; asm("v_mov_b32 %0, %1" : "=v"(r) : "s"(x)); return r; }

// The exact other-axis cell after the exit crossing at time tt (all in the mirrored frame: the ray heads towards +).
// est_T / cur_T: the estimated new cell (off by at most one) and the current one as float bits, og / oid the origin and
// |1 / d| on that axis.  The spec's traversal crosses boundary b before the exit iff t_b < tt, or t_b == tt when the
// exit is an x crossing (tie != 0: "ties go to y"); boundary times can be -0.0, so these are IEEE comparisons.
__device__ __forceinline__ uint32_t exact_other_cell_m(uint32_t est_T, uint32_t cur_T, float og, float oid, float tt, uint32_t tie) {
    const int m0 = max((int)(est_T - cur_T) - 1, 0);
    const uint32_t b0T = cur_T + 1u + (uint32_t)m0;                       // first boundary that is in doubt
    const float b0 = __uint_as_float(b0T) - kCellMagic;
    const float tb0 = (b0 - og) * oid, tb1 = ((b0 + 1.0f) - og) * oid;
    const uint32_t c0 = (tb0 < tt || (tie != 0 && tb0 == tt)) ? 1u : 0u;
    const uint32_t c1 = (tb1 < tt || (tie != 0 && tb1 == tt)) ? 1u : 0u;
    return b0T - 1u + c0 + c1;
}

// New cell after an exit: z = fma(tt, |d|, fl(g~ + band - 0.5)) on BOTH axes, T = bits(z + 1.5 * 2^23), i.e. the cell
// is rne(z) = floor(position + band) - except at an exact tie, which the band test below catches.  With u = 2^-24 and
// M = the largest coordinate on the grid (cells):
// * exit axis: the position is the boundary xe itself, and z - (xe + band - 0.5) is at most
//   (xe - g~)(e0 + e1 + e2) + (g~ + band - 0.5) e4 + z e3 with |e| <= u - the subtraction, the reciprocal, the product,
//   the rounded origin and the FMA: 5 u M = 3.0e-7 M cells.  band > 5 u M makes floor(xe + band +- 5 u M) = xe, so no
//   select between "the boundary" and "floor of the position" is needed.
// * other axis: floor(position + band) is trusted unless fract(z + 0.5) < 2 band, i.e. unless the position lies within
//   `band` of a cell boundary; then the spec's own comparisons decide (exact_other_cell_m).  The spec crosses boundary b
//   iff fl(fl(b - g) * fl(1/d)) < tt, which differs from the real-number test "b before the position at time tt" by at
//   most 3 u M; z carries 2 u M of its own (origin, FMA) and the sum z + 0.5 of the test 1 u M: a position is in doubt
//   only if z + 0.5 lands in [k, k + band + 6 u M) for an integer k, so 2 band >= band + 6 u M is needed.
// band = (M + 2) 2^-21 = 8 u M covers both (8 > 5, 8 > 6): 2.6e-4 cell on austria (548 cells wide), 1e-3 on a 2048-cell
// map.  tools/band_validation.sh (profiles/r02_d_band_validation.txt): the parity tests fail for bands <= M 2^-24 - the
// exit axis then lands in the wrong cell and rays run off - and pass from M 2^-22 on.
// The traversal proper: from start cell (ix, iy) with start entry v (FROM_PLANE: read from the ray's plane instead, 0
// when !in_grid), direction (dx, dy) (finite, never -0.0), its reciprocals and sign masks nx, ny (-1 for a negative
// component, 0 otherwise).  21 full-rate and 9 half-rate vector instructions per trip.
struct NothingBetween { __device__ __forceinline__ void operator()() const {} };
// GUARD: the bounded form of the trip loop (a wave-level trip budget, see the loop).  The loop does not need one: with the
// shipped band a trip puts the ray into the cell behind the exit boundary (exactly, see above), at least one cell further
// along the exit axis, and never back on the other one, so after at most w + h trips the ray stands in a stop cell - the
// grid is ringed by them.  The bound costs 4 % of the scan, so it is compiled into the builds that run when that proof
// does not cover the run: the validation scan of rc_load_track, any run with a validation band or RC_DBG_SCAN_BOUNDED,
// and the per-ray variant 6.
template <bool FROM_PLANE, class Between = NothingBetween, bool GUARD = true>
__device__ __forceinline__ float ray_traverse(const uint16_t *qr, const RcTrackDev &t, const TravConst &k, float gx, float gy,
                                              float dx, float dy, float idx, float idy, int nx, int ny, int ix, int iy,
                                              unsigned v, bool in_grid, int *wave_trips = nullptr, int *wave_exact = nullptr,
                                              Between between = Between(), int *overrun = nullptr) {
    const int pitch2 = t.cell_pitch * 2;
    const char *qb = reinterpret_cast<const char *>(qr);
    // mirrored origin, the origin of the position estimate, the start cell (i~ = ~i on a mirrored axis) and the part of
    // the table address that depends on the quadrant only
    const float gmx = __uint_as_float(__float_as_uint(gx) ^ ((uint32_t)nx & 0x80000000u));
    const float gmy = __uint_as_float(__float_as_uint(gy) ^ ((uint32_t)ny & 0x80000000u));
    const float hx = gmx + k.band_mh, hy = gmy + k.band_mh;
    uint32_t Tx = (uint32_t)(ix ^ nx) + kCellMagicBits, Ty = (uint32_t)(iy ^ ny) + kCellMagicBits;
    uint32_t qoff = (((uint32_t)nx & k.kx) + ((uint32_t)ny & k.ky)) + k.c00;
    asm("" : "+v"(qoff));                                                 // one value: keep it out of the loop's address math
    if (FROM_PLANE) {
        v = 0;
        if (in_grid) v = *reinterpret_cast<const uint16_t *>(qb + mad_u24(Ty, pitch2, (Tx << 1) + qoff));
    }
    const float band2 = t.band2;
    float tt = 0.0f;
    // A trip in two halves: `trip_head` ends with the REQUEST for the new cell's entry, `trip_tail` has the band test, the
    // exact path and the wait.  The entry is requested from the estimate as soon as the cell is known - the band test and
    // its branch are then off the chain entry -> trip -> request that a wave's time consists of (a lane in the band asks
    // again in the tail) - and as inline assembly with its own wait: the compiler sinks a C++ load below the branch.
    uint32_t xe = 0, ye = 0, xflag = 0;                                   // xflag: 1 = left through the x side (per lane: a lane
    float zo = 0.0f;                                                      // mask in a scalar pair cannot live across `between`)
    unsigned vnew = 0;
    auto trip_head = [&]() {
        if (wave_trips) *wave_trips += 1;                                 // (instrumented build only: this LANE's trips)
        xe = add_ubyte<0>(v, Tx); ye = add_ubyte<1>(v, Ty);               // boundaries that leave the rectangle, as float bits
        const float ax = (__uint_as_float(xe) - kCellMagic) - gmx, ay = (__uint_as_float(ye) - kCellMagic) - gmy;
        const float txe = ax * fabsf(idx), tye = ay * fabsf(idy);
        // leaves through the x side iff txe < tye (ties: y)
        const unsigned long long xm = cmp_lt_f32(txe, tye);
        tt = select_mask(xm, txe, tye);
        // (an estimate, not a spec value - the exact path covers its error: a fused multiply-add is welcome)
        const float zx = __builtin_fmaf(tt, fabsf(dx), hx), zy = __builtin_fmaf(tt, fabsf(dy), hy);
        Tx = __float_as_uint(zx + kCellMagic);
        Ty = __float_as_uint(zy + kCellMagic);
        asm volatile("global_load_ushort %0, %1, %2" : "=v"(vnew) : "v"(mad_u24(Ty, pitch2, (Tx << 1) + qoff)), "s"(qb));
        float zxo = zx, zyo = zy;
        asm volatile("" : "+v"(zxo), "+v"(zyo));                          // (keeps the band test's arithmetic behind the request)
        zo = select_mask(xm, zyo, zxo);                                   // the other axis
        xflag = select_mask_u(xm, 1u, 0u);
    };
    auto trip_tail = [&]() {
        if (cmp_lt_f32_s(__builtin_amdgcn_fractf(zo + 0.5f), band2)) {    // within `band` of a boundary: exact count
            // (first let the request land: its register must not be handed to anything else while it is under way)
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(vnew));
            if (wave_exact) *wave_exact += 1;
            const unsigned long long xm = cmp_ne_u32(xflag, 0u);
            const uint32_t tie = xflag;
            // the current cell on that axis = boundary - extent (the old Tx, Ty are not kept: no register copies per trip)
            // (the entry through an opaque copy: shared with the loop condition, `v & 255` would stay a separate
            // instruction in every trip instead of folding into the compare's byte select)
            unsigned vv = v;
            asm("" : "+v"(vv));
            const uint32_t cur = select_mask_u(xm, ye, xe) - select_mask_u(xm, vv >> 8, vv & 255u);
            const uint32_t nT = exact_other_cell_m(select_mask_u(xm, Ty, Tx), cur, select_mask(xm, gmy, gmx),
                                                   select_mask(xm, fabsf(idy), fabsf(idx)), tt, tie);
            Tx = select_mask_u(xm, Tx, nT);
            Ty = select_mask_u(xm, nT, Ty);
            vnew = *reinterpret_cast<const uint16_t *>(qb + mad_u24(Ty, pitch2, (Tx << 1) + qoff));
        }
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(vnew));
        v = vnew;
    };
    if (!std::is_same<Between, NothingBetween>::value) {
        // The first trip peeled: `between` - the caller's work that does not depend on this ray, in the per-car kernel
        // the preparation of the NEXT round - runs for ALL lanes while the first request is under way.  Pays where a SIMD
        // holds few waves (4 096 cars: 0.0257 -> 0.0235 ms); with 8 waves per SIMD the others fill that time anyway and
        // the second copy of the trip only costs (65 536 cars: 0.186 -> 0.191 ms), so the launcher picks it by batch size.
        const bool started = (v & 255u) != 0;                             // false: the sensor sits in a stop cell
        if (started) trip_head();
        between();
        if (!started) return 0.0f;
        trip_tail();
    } else if ((v & 255u) == 0) {
        return 0.0f;                                                      // the sensor sits in a stop cell
    }
    if (GUARD) {
        // THE BOUNDED FORM of the loop: a wave-level trip budget of w + h + 2, counted on the scalar unit behind the trip's
        // table request.  A lane goes on while the width byte of its entry exceeds `kz`, which is 0 - "not a stop cell" -
        // until the budget is used up and 255 from then on: every lane then counts as stopped, and the unfinished ones
        // read "no return" (their entry is not 0).  With the shipped band the budget is never used up (every trip moves
        // every unfinished ray at least one cell along its exit axis, see above); a corrupted table line or a mis-set band
        // ends in "no return" and a count in RcParams::scan_overrun instead of a hung wave.
        // Cost, A/B on one box (profiles/r03_a_ab_trip_bound.txt): 0.1789 -> 0.1864 ms at 65 536 cars (+ 4.2 %; nothing at
        // 4 096) for its three instructions per trip - the "shadow" of the request is already full of the band test - against
        // + 2.2 % for the per-lane counter of round 2; a per-pair count in a loop body of two trips and a scalar threshold
        // operand of the compare (`inverse_ballot`) both made the compiler's loop control longer than what they saved.
        // So the bound is not in the production loop: it runs (a) over every spawn pose of a track when its tables are
        // built (rc_load_track fails if any ray overruns), (b) whenever a validation band is set, (c) on request
        // (RC_DBG_SCAN_BOUNDED), and always in the per-ray variant 6.
        int budget = t.w + t.h + 2;
        uint32_t kz = 0u;
        asm volatile("v_mov_b32 %0, 0" : "=v"(kz));
        while ((v & 255u) > kz) {
            trip_head();
            int lim;
            asm volatile("s_sub_u32 %0, %0, 1\n\ts_cselect_b32 %1, 255, 0" : "+s"(budget), "=s"(lim) : : "scc");
            asm volatile("v_mov_b32 %0, %1" : "=v"(kz) : "s"(lim));
            trip_tail();
        }
        if (overrun != nullptr && budget < 0) *overrun = 1;               // (wave-uniform)
    } else {
        while ((v & 255u) != 0) { trip_head(); trip_tail(); }
    }
    // The one place where the mirrored frame is not bit-identical: a zero boundary time.  The spec's fl(b - g) is +0 and
    // its product with 1/d < 0 is -0.0, which the range then carries; here it is +0.  A ray that stops at time 0 never
    // left its origin; it crossed x at all only if it started on the far face of its column, and when it crossed both
    // axes (a corner) the spec stepped y first - so its last crossing was the x one iff it left the start column.
    // (a wave-uniform branch that is almost never taken: one compare per round; the start cell is recomputed inside it
    // rather than kept in a register across the loop)
    if (__builtin_amdgcn_ballot_w64(tt == 0.0f) != 0) {
        int mx = nx;
        asm volatile("" : "+v"(mx));
        const uint32_t sgn = (uint32_t)(Tx != (uint32_t)(ix ^ mx) + kCellMagicBits ? mx : ny) & 0x80000000u;
        tt = __uint_as_float(__float_as_uint(tt) | (tt == 0.0f ? sgn : 0u));
    }
    // stopped at a wall within range: the range; beyond 15 m, at the ring (entry 0x0100) or never stopped: no return
    return select_mask(cmp_nlt_f32_s(tt, t.tmax) | cmp_ne_u32(v, 0u), RCS_MAX_RANGE, tt * k.res);
}

// Two values of a LiDAR row as uint16: q = rne(fl(fl(v + off) * scale)), one IEEE operation per operator like the rest
// of the spec (oracle: racecar_oracle.quantise_lidar_u16).  Adding 2^23 rounds the product to an integer (ties to
// even) in the float's low mantissa bits; the product is <= 65 535 < 2^23.
__device__ __forceinline__ uint32_t quantise_pair(float a, float b, float off, float scale) {
    const float ta = (a + off) * scale + 8388608.0f, tb = (b + off) * scale + 8388608.0f;
    return (__float_as_uint(ta) & 0xffffu) | (__float_as_uint(tb) << 16);
}

constexpr unsigned kCarRowBytes = ((RC_N_BEAMS + 63) / 64) * 64 * 4;   // LDS per wave of rc_raycast_car_kernel: its car's ranges ...
constexpr unsigned kCarLdsBytes = kCarRowBytes + 2 * RC_FIRST_PLANES;  // ... and the start cell's line of the first-trip table

// STAMPS: the instrumented build (rc_debug_scan_stamps): RC_STAMP_SLOTS uint64 per wave.  Shader-clock values (s_memtime: a
// per-CU counter, comparable within a wave only) at fixed points of the wave's life - slot 0 entry, 1 car state arrived,
// 2 first-trip line staged and first round prepared, 20 rounds done, 21 flush issued; the chip-wide 100 MHz clock
// (s_memrealtime) at entry / flush in 6 / 7; the phases of a round summed over the wave's rounds in 27 (wait), 28 (prepare),
// 29 (traversal), 3 (inter-car returns, transform), 4 (store to LDS), 30 (the rest); 22 wave-level trips, 23 lanes that ever
// took the exact path, 25 / 26 trips per round (a nibble each), 24 HW_ID, 5 XCC_ID.
// OVERLAP: the next round is prepared under the first request of the current one (ray_traverse's `between`) instead of
// ahead of the traversal.
template <int A, bool STAMPS = false, bool OVERLAP = false, bool GUARD = true>
__device__ __forceinline__ void scan_car(const RcParams &p, const unsigned car, const unsigned part, const int split,
                                         const unsigned lane, const uint32_t lds_row, unsigned long long *stamps = nullptr) {
    const RcTrackDev &t = p.trk;
    auto stamp = [&](int slot) {
        if (STAMPS && stamps != nullptr) {
            const unsigned long long now = __builtin_amdgcn_s_memtime();
            if (lane == 0) stamps[slot] = now;
        }
    };
    auto stamp_value = [&](int slot, unsigned long long value) {
        if (STAMPS && stamps != nullptr && lane == 0) stamps[slot] = value;
    };
    int wave_trips = 0, wave_exact = 0, round_index = 0, trips_before = 0, total_trips = 0;
    int overrun = 0;                                                      // bounded build: a round used up its trip budget
    // phases of a round, summed over the wave's rounds in scalar registers (no stores in between): 27 wait for the
    // previous round's loads, 28 prepare the next round, 29 traversal, 3 inter-car returns and transform, 4 LDS store,
    // 30 round loop control
    unsigned long long t_wait = 0, t_prep = 0, t_trav = 0, t_rest = 0, t_post = 0, t_store = 0, t_mark = 0;
    auto phase = [&](unsigned long long &acc) {
        if (STAMPS) {
            const unsigned long long now = __builtin_amdgcn_s_memtime();
            acc += now - t_mark;
            t_mark = now;
        }
    };
    unsigned long long nib_lo = 0, nib_hi = 0;
    stamp(0);
    if (STAMPS) stamp_value(6, __builtin_amdgcn_s_memrealtime());       // the 100 MHz reference clock, one for the whole chip (s_memtime is per CU)
    // the wave's first beam pair does not depend on the car: requested before the car's state, so the two round trips
    // overlap (a wave's start-up - state, start cell, first-trip line - is serial latency that nothing else hides)
    const char *beams = reinterpret_cast<const char *>(t.beams);        // padded to 17 * 64 entries (rc_load_track)
    unsigned boff = lane * 8u + 512u * part;                             // byte offset of this lane's beam pair
    float2 bm = *reinterpret_cast<const float2 *>(beams + boff);
    // all four state words in one scalar 16-byte load
    const float4 sp = p.st.scan_pose[car];
    float ct = sp.z, st = sp.w;
    const float car_x = sp.x, car_y = sp.y;
    // One check per car instead of a clamp per ray: a heading whose (cos, sin) pair is
    // not finite, not of magnitude <= 2 or not at least 0.5 in one component (a diverged car state; sincos32 never
    // produces one from a finite angle) is replaced by heading 0 - the scan of such a car is unspecified, it only
    // has to terminate.  With a legal pair and the beam table's entries all non-zero (checked at rc_load_track) at
    // least one of the two products in dx = ct cb - st sb and in dy = st cb + ct sb is non-zero, so neither
    // component can be -0.0, which the spec would step as +.
    const bool legal = ((int)(fabsf(ct) <= 2.0f) & (int)(fabsf(st) <= 2.0f) & ((int)(fabsf(ct) >= 0.5f) | (int)(fabsf(st) >= 0.5f))) != 0;
    if (!legal) { ct = 1.0f; st = 0.0f; }
    if (STAMPS) { asm volatile("" :: "v"(ct)); stamp(1); }
    const float lx = car_x + RCS_LIDAR_X * ct;
    const float ly = car_y + RCS_LIDAR_X * st;
    const float gx = (lx - t.org_x) * t.inv_res;
    const float gy = (ly - t.org_y) * t.inv_res;
    float *out = p.out.lidar + (size_t)car * RC_N_BEAMS;
    // The start cell and its 512-byte line of the first-trip table: the same for all 1080 rays, so the wave copies it
    // into its LDS ONCE (eight bytes per lane; zeros when the sensor is off the grid: every ray then reads 0) and each
    // round's 64 entries come from there.  As a global load per round it was 17 vector-memory instructions per car,
    // each of them 16 quad requests to the L1 (which counts requests, not bytes: 84 M per launch kept it 77 % busy).
    const int ix = __builtin_amdgcn_readfirstlane((int)floorf(gx)), iy = __builtin_amdgcn_readfirstlane((int)floorf(gy));
    const uint32_t lds_first = lds_row + kCarRowBytes;
    {
        static_assert(2 * RC_FIRST_PLANES == 64 * 8, "one 8-byte piece of the line per lane");
        typedef unsigned v2u_t __attribute__((ext_vector_type(2)));
        v2u_t w = {0u, 0u};
        if ((unsigned)ix < (unsigned)t.w && (unsigned)iy < (unsigned)t.h)
            w = reinterpret_cast<const v2u_t *>(reinterpret_cast<const char *>(t.first_rect) + ((size_t)iy * t.cell_pitch + ix) * (2 * RC_FIRST_PLANES))[lane];
        typedef __attribute__((address_space(3))) v2u_t *lds_v2u_ptr;
        *(lds_v2u_ptr)(uintptr_t)(lds_first + 8u * lane) = w;
    }
    const uint32_t first_line = lds_first - 2u * RC_FIRST_BIAS;          // (the slope bins are biased: first_trip_entry)
    // wave-uniform operands of the trip, in vector registers
    const TravConst kc = trav_const(t);
    const TravConst k = {pin_vgpr(kc.band_mh), pin_vgpr(kc.res), pin_vgpr(kc.kx), pin_vgpr(kc.ky), pin_vgpr(kc.c00)};
    const int ixv = pin_vgpr(ix), iyv = pin_vgpr(iy);
    constexpr int kRounds = (RC_N_BEAMS + 63) / 64;
    // (per-round steps live in vector registers: a full-rate add that reads a scalar register issues at half rate)
    const unsigned bstep = pin_vgpr(512u * (unsigned)split), ostep = pin_vgpr(256u * (unsigned)split);
    // LDS address of this lane's slot in the wave's staged output row (lds_row = LDS address of the row: the kernel's
    // dynamic LDS is its only LDS object and starts at 0, checked by rck_set_lds_limits) and the row's end
    typedef __attribute__((address_space(3))) float *lds_f32_ptr;
    typedef const __attribute__((address_space(3))) v4u *lds_v4u_ptr;
    uint32_t oslot = lds_row + lane * 4u + 256u * part;
    const uint32_t oend = lds_row + 4u * RC_N_BEAMS;
    // Software pipeline over the rounds: while round r is traversed, round r + 1's direction, reciprocals, sign masks
    // and first-trip entry are already computed / in flight (and round r + 2's beam pair is being fetched), so no
    // round starts by waiting for its start entry.  Two register sets take turns (the loop body holds two rounds), so
    // nothing is copied between them.
    struct Ray { float dx, dy, idx, idy; int nx, ny; unsigned v; };
    // (dx, dy) = (ct cb - st sb, ct sb + st cb), one rounding per operator: four products, a subtract and an add - plain
    // full-rate instructions (the packed forms issue at half rate and need their operands swizzled into pairs)
    auto prepare = [&](float2 b, Ray &r) {
        r.dx = ct * b.x - st * b.y;
        r.dy = ct * b.y + st * b.x;
        ray_reciprocals(r.dx, r.dy, r.idx, r.idy);
        r.nx = sign_mask(r.dx); r.ny = sign_mask(r.dy);
        r.v = first_trip_entry(first_line, r.dy, r.idx, r.nx, r.ny);
    };
    // one round: prepare `nxt` for round + split, traverse `cur`, store.  false: this lane has no beam in the round
    auto stage = [&](int round, const Ray &cur, Ray &nxt) -> bool {
        if (oslot >= oend) return false;                                  // last round: 56 of 64 lanes
        phase(t_rest);
        // `cur` was requested a whole round ago and has arrived: say so BEFORE the next round's loads go out, or the
        // compiler, unable to count the conditional loads in flight, waits for all of them at the first use of cur.v
        // (vmcnt(0), other counters untouched)
        __builtin_amdgcn_s_waitcnt(0x0F70);
        phase(t_wait);
        // the next round is prepared while the first request of this one is under way (ray_traverse calls it there)
        auto prepare_next = [&]() {
            if (round + split < kRounds) {
                prepare(bm, nxt);                                         // (the padded beams of the last round included)
                boff += bstep;
                if (round + 2 * split < kRounds) bm = *reinterpret_cast<const float2 *>(beams + boff);
            }
        };
        float rng;
        if (OVERLAP) {
            rng = ray_traverse<false, decltype(prepare_next), GUARD>(t.quad_rect, t, k, gx, gy, cur.dx, cur.dy, cur.idx, cur.idy, cur.nx, cur.ny, ixv, iyv, cur.v, true,
                                      STAMPS ? &wave_trips : nullptr, STAMPS ? &wave_exact : nullptr, prepare_next, GUARD ? &overrun : nullptr);
        } else {
            prepare_next();
            if (STAMPS) asm volatile("" :: "v"(nxt.idx), "v"(nxt.idy));
            phase(t_prep);
            rng = ray_traverse<false, NothingBetween, GUARD>(t.quad_rect, t, k, gx, gy, cur.dx, cur.dy, cur.idx, cur.idy, cur.nx, cur.ny, ixv, iyv, cur.v, true,
                                      STAMPS ? &wave_trips : nullptr, STAMPS ? &wave_exact : nullptr, NothingBetween(), GUARD ? &overrun : nullptr);
        }
        if (STAMPS) asm volatile("" :: "v"(rng));
        phase(t_trav);
        if (A > 1) {
            const unsigned env = car / A;
#pragma unroll
            for (unsigned o = 0; o < (unsigned)A; ++o) {
                const unsigned oc = env * A + o;
                if (oc != car) {
                    const float tc = ray_vs_car(lx, ly, cur.dx, cur.dy, p.st.x[oc], p.st.y[oc], p.st.ct[oc], p.st.st[oc]);
                    rng = tc < rng ? tc : rng;
                }
            }
        }
        if (p.lidar_transform == 1) rng = rng / RCS_MAX_RANGE - 0.5f;                 // dreamer/tools.py:274
        else if (p.lidar_transform == 2) rng = rng * (1.0f / RCS_MAX_RANGE);          // single_agent.py:92-99
        if (STAMPS) asm volatile("" :: "v"(rng));
        phase(t_post);
        *(lds_f32_ptr)(uintptr_t)oslot = rng;                             // staged: see the flush below
        oslot += ostep;
        phase(t_store);
        if (STAMPS) {                                                     // + this round's trips, a nibble per round
            phase(t_rest);                                                // (the counting below is not charged to any phase)
            // wave-level trips of the round = the most any lane made (the counters are per lane)
            int most = 0;
            for (int n = 1; n <= 15; ++n) most = __builtin_amdgcn_ballot_w64(wave_trips - trips_before >= n) != 0 ? n : most;
            total_trips += most;
            const unsigned long long n = (unsigned long long)most;
            if (round_index < 16) nib_lo |= n << (4 * round_index); else nib_hi |= n << (4 * (round_index - 16));
            trips_before = wave_trips;
            ++round_index;
            // (read the clock AFTER the counting: the builtin alone may be scheduled ahead of it)
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_mark) : "s"(most), "s"(nib_lo), "s"(nib_hi));
        }
        return true;
    };
    Ray ra, rb;
    prepare(bm, ra);
    if (STAMPS) { asm volatile("" :: "v"(ra.v)); stamp(2); t_mark = __builtin_amdgcn_s_memtime(); }
    boff += bstep;
    if ((int)part + split < kRounds) bm = *reinterpret_cast<const float2 *>(beams + boff);
    for (int round = (int)part; round < kRounds; round += 2 * split) {
        if (!stage(round, ra, rb)) break;
        if (round + split >= kRounds) break;
        if (!stage(round + split, rb, ra)) break;
    }
    if (GUARD && overrun != 0 && p.scan_overrun != nullptr && lane == 0) atomicAdd(p.scan_overrun, 1u);
    // Flush the wave's ranges from LDS to the output row.  A store per round costs more than its 256 bytes: loads and
    // stores share one in-order counter on gfx9, so the first table load of the NEXT round also waited for the
    // store's acknowledgement from L2 (the scan ran 11 % faster with the stores removed).  Staged in LDS (its own
    // counter), the 17 rows go out back to back at the end and nothing waits for them.
    stamp(20);
    if (GUARD && p.out.lidar == nullptr) return;                          // the validation scan of rc_load_track keeps no ranges
    char *out_bytes = reinterpret_cast<char *>(out);
    // Optional second copy of the row as uint16 (rc_set_compact_slab: the half-size record of the multi-GPU gather):
    // q = rne((value + q_off) * q_scale), 0 .. 65535 over the row's value range - taken from the same LDS row, so it
    // costs the scan 2 160 more bytes of stores per car and ~30 instructions.
    char *out16 = reinterpret_cast<char *>(p.out.lidar_u16);            // wave-uniform, null when not asked for
    if (out16 != nullptr) out16 += (size_t)car * (2 * RC_N_BEAMS);
    const float q_off = p.lidar_transform == 1 ? 0.5f : 0.0f;
    const float q_scale = p.lidar_transform == 0 ? 65535.0f / RCS_MAX_RANGE : 65535.0f;
    if (split == 1) {                   // the whole row is this wave's: 270 16-byte vectors, 5 stores of 1 KB
#pragma unroll
        for (int k = 0; k < (RC_N_BEAMS / 4 + 63) / 64; ++k) {
            const unsigned o = (lane + 64u * (unsigned)k) * 16u;
            if (o < 4u * RC_N_BEAMS) {
                const v4u val = *(lds_v4u_ptr)(uintptr_t)(lds_row + o);
                __builtin_nontemporal_store(val, reinterpret_cast<v4u *>(out_bytes + o));    // streamed: leaves the tables in L2 (1 % faster)
                if (out16 != nullptr) {
                    typedef unsigned v2u __attribute__((ext_vector_type(2)));
                    const v2u q = {quantise_pair(__uint_as_float(val.x), __uint_as_float(val.y), q_off, q_scale),
                                   quantise_pair(__uint_as_float(val.z), __uint_as_float(val.w), q_off, q_scale)};
                    __builtin_nontemporal_store(q, reinterpret_cast<v2u *>(out16 + (o >> 1)));
                }
            }
        }
    } else {
        for (unsigned o = lane * 4u + 256u * part; o < 4u * RC_N_BEAMS; o += 256u * (unsigned)split) {
            const float v = *(lds_f32_ptr)(uintptr_t)(lds_row + o);
            *reinterpret_cast<float *>(out_bytes + o) = v;
            if (out16 != nullptr) *reinterpret_cast<uint16_t *>(out16 + (o >> 1)) = (uint16_t)quantise_pair(v, v, q_off, q_scale);
        }
    }
    if (STAMPS) {
        stamp(21);
        stamp_value(7, __builtin_amdgcn_s_memrealtime());
        stamp_value(22, (unsigned long long)total_trips);
        stamp_value(23, (unsigned long long)__builtin_popcountll(__builtin_amdgcn_ballot_w64(wave_exact != 0)));   // lanes that ever took it
        stamp_value(27, t_wait);
        stamp_value(28, t_prep);
        stamp_value(29, t_trav);
        stamp_value(30, t_rest);
        stamp_value(3, t_post);
        stamp_value(4, t_store);
        stamp_value(25, nib_lo);
        stamp_value(26, nib_hi);
        stamp_value(24, (unsigned long long)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | ((32 - 1) << 11)));   // HW_ID
        stamp_value(5, (unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | ((32 - 1) << 11)));   // XCC_ID
    }
}

}  // namespace
