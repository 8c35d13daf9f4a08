// obs_type lidar_occupancy_reference: the reference's OccupancyMapObs.step (dreamer/wrappers.py:396-406: to_pixel, 220 x 220
// crop, scipy.ndimage.rotate, centre 200 x 200, PIL resize to 64 x 64) restated down to the binary64 operation.
// The spec - with the library line every step follows - is oracle/patch_reference.py (render_patch_exact); the C oracle's
// oc_patch_exact_range and the two kernels here are the same arithmetic, operation by operation (IEEE binary64, no fused
// multiply-add: the library is built with -ffp-contract=off; integer arithmetic for Pillow's resize).
//
//   rc_patch_exact_prefilter_kernel   one 256-thread workgroup per car: the crop's cubic-spline coefficients, 220 x 220 binary64
//       (387 KB per car - no LDS holds it: the registers and the LDS of a CU together do), both axes and the transposition between
//       them on the chip, the result written once to the car's scratch in [column][row] order.
//   rc_patch_exact_sample_kernel      one 256-thread workgroup per car: the 200 x 200 centre window of the rotated image in tiles
//       of 25 x 20 pixels (a tile's coefficients staged in LDS, the next tile's already on their way in registers; a pixel is
//       decided by a binary32 estimate of its 16-tap sum wherever the estimate's error bound allows, by the library's binary64
//       sum where it does not - PX_BAND), then Pillow's two integer passes (200 x 200 -> 200 x 64 -> 64 x 64) from LDS, the patch
//       written once.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "racecar_internal.h"       // RcExactParams

#define PX_CROP 220
#define PX_WIN 200
#define PX_OUT 64
#define PX_KSIZE 15
#define PX_BITS 22
#define PX_T 25                               // the sample kernel renders the 200 x 200 window in 8 x 10 tiles of PX_T rows x PX_TW columns:
#ifndef PX_TW
#define PX_TW 20                              // 500 pixels = two passes of 256 threads at 98 % (square tiles of 40 / 25 / 20 / 10: EXPERIMENTS.md 000.4;
                                              // 25 x 25 is three passes at 81 %)
#define PX_TILE_N 39                          // a tile's taps lie within sqrt(24^2 + 19^2) + 9 = 39.6 coefficients per axis
#define PX_TILE_PITCH 39                      // values per staged column (odd: consecutive columns start in different LDS banks)
#endif
#define PX_Z (-0.2679491924311227)            // scipy ni_splines.c: the cubic spline's pole, sqrt(3) - 2 correctly rounded
#define PX_ZN (-5.539710763905135e-126)       // pow(PX_Z, 219)
#define PX_PI180 1.74532925199432957692e-2
#define PX_CAR_DOUBLES RC_EXACT_CAR_DOUBLES

namespace px {

__device__ __forceinline__ bool skip_car(const RcExactParams &p, int car) {
    // all-zero patch: the first observation of an episode (dreamer/wrappers.py:413), or a position that is no position
    const float x = p.x[car], y = p.y[car];
    return p.fresh[car] != 0 || !(fabsf(x) <= 1.0e5f && fabsf(y) <= 1.0e5f);
}

__device__ __forceinline__ void pixel_of(const RcExactParams &p, int car, int &pr, int &pc) {
    const double x = (double)p.x[car], y = (double)p.y[car];
    pr = (int)((double)p.fh - (y - p.oy) / p.res);          // GridMap.to_pixel: truncation towards zero
    pc = (int)((x - p.ox) / p.res);
}

__device__ __forceinline__ void sincos_deg(double x, double &cosv, double &sinv) {       // patch_reference.py, sincos_degrees
    double y = floor(x / 45.0);
    int j = (int)(y - 8.0 * floor(y / 8.0));
    if (j & 1) { y = y + 1.0; j += 1; }
    j &= 7;
    const double z = (x - y * 45.0) * PX_PI180, zz = z * z;
    const double sp = z + z * (zz * (-1.0 / 6.0 + zz * (1.0 / 120.0 + zz * (-1.0 / 5040.0 + zz * (1.0 / 362880.0 + zz * (-1.0 / 39916800.0 + zz * (
        1.0 / 6227020800.0 + zz * (-1.0 / 1307674368000.0))))))));
    const double cp = 1.0 - zz * (0.5 - zz * (1.0 / 24.0 - zz * (1.0 / 720.0 - zz * (1.0 / 40320.0 - zz * (1.0 / 3628800.0 - zz * (1.0 / 479001600.0 - zz * (
        1.0 / 87178291200.0 - zz * (1.0 / 20922789888000.0))))))));
    cosv = j == 0 ? cp : (j == 2 ? -sp : (j == 4 ? -cp : sp));
    sinv = j == 0 ? sp : (j == 2 ? cp : (j == 4 ? -sp : -cp));
}

// x / 6.0, correctly rounded, without the division: q = RN(x r) with r = RN(1 / 6) is a faithful quotient, the remainder
// x - 6 q is exact in one fused multiply-add, and RN(q + rem r) is the correctly rounded quotient (Markstein's theorem; 6 has no
// all-ones significand, and the operands here - spline weights' numerators, 2^-159 .. 8 - are far from underflow).  The compiler's
// own expansion of a binary64 division is ~12 instructions around a quarter-rate v_rcp_f64; a pixel needs six such quotients.
// `rc_selftest_div6` compares the two on the device over 2^32 operands (tests/test_gpu_api.py).
__device__ __forceinline__ double div6(double x) {
    const double r = 1.0 / 6.0;
    const double q = x * r;
    const double rem = __builtin_fma(-6.0, q, x);
    return __builtin_fma(rem, r, q);
}

__device__ __forceinline__ void weights(double cc, double (&w)[4]) {
    const double y = cc - floor(cc), z = 1.0 - y;
    w[1] = div6((y * y) * (y - 2.0) * 3.0 + 4.0);
    w[2] = div6((z - 2.0) * (z * z) * 3.0 + 4.0);
    w[0] = div6((z * z) * z);
    w[3] = ((1.0 - w[0]) - w[1]) - w[2];
}

// The same weights in binary32 from the fractional part - for the sampling pass's ESTIMATE only (see PX_BAND): absolute error
// <= 1e-6 each, <= 2e-6 for the one got by subtraction.
__device__ __forceinline__ void weights32(float y, float (&w)[4]) {
    const float z = 1.0f - y, sixth = 1.0f / 6.0f;
    w[1] = ((y * y) * (y - 2.0f) * 3.0f + 4.0f) * sixth;
    w[2] = ((z - 2.0f) * (z * z) * 3.0f + 4.0f) * sixth;
    w[0] = ((z * z) * z) * sixth;
    w[3] = ((1.0f - w[0]) - w[1]) - w[2];
}

__device__ __forceinline__ int mirror(int i) {
    i = i < 0 ? -i : i;
    return i >= PX_CROP ? 2 * PX_CROP - 2 - i : i;
}

}  // namespace px

// ---------------------------------------------------------------------------------------------------------------- prefilter
// The crop's cubic-spline coefficients (patch_reference.py, spline_coefficients): along every column, then along every row, the
// mirror-summed causal start, the causal recursion, the anticausal start and recursion - 880 dependent steps per line, each a
// binary64 multiply and a dependent add (45 cycles on a SIMD whatever the number of active lanes: tools/ubench/f64_chain.hip), in
// the library's order: nothing to reassociate.  What sets the pace is the number of lines a CU has in flight.  One workgroup per
// car, a line per LANE: 4 waves x 55 lanes = the 220 lines of an axis at once, a line's 220 values in the lane's registers
// (PXR_REG of them; the kernel runs at one wave per SIMD with all 512 registers) and LDS (the rest).  Every loop over a line is
// fully unrolled, so every register index is a constant.  Axis 0 (lane = column): the column's bits from the staged crop, filtered
// in place.  Then the array is transposed ON THE CHIP (pxr_transpose) and axis 1 (lane = row) filters in place again, its backward
// pass storing [column][row] - contiguous across lanes - straight from registers: the only memory traffic is the 387 KB result.
// Forms measured before this one (EXPERIMENTS.md 000.4): forward values through memory (HBM-bound at 4.6 TB/s, 1.56 ms per 2 048
// cars), lines in LDS ten per wave (80 lines in flight per CU: 0.97 ms), this form with the intermediate array through memory
// (HBM-bound again: 0.54 ms); now 0.47 ms - one car alone on a CU: 45 us = axis 0 14 + transposition 12 + axis 1 16 + the rest 3
// (tools/ubench/exact_latency.hip), the axes at the chains' own length (660 two-operation and 220 one-operation steps).
#define PXR_REG 161
#define PXR_LDS (PX_CROP - PXR_REG)
#define PXR_LANES 55
#define PXR_TB 16                               // the backward pass is unrolled in blocks of this many steps
#ifndef PXR_SB
#define PXR_SB 8                                // the scheduler may move code within blocks of this many steps only (else it
#endif                                          // hoists hundreds of loads and spills; 16: no spill either, same time; 4, 12: spills)
#define PXR_LINES (PX_CROP + 4)                 // + one dummy line per wave for the idle lanes 55 .. 63

// Slot i of the lane's line: register S[i] for i < PXR_REG, else the lane's LDS row sl[i - PXR_REG].  i is a constant wherever these
// are used (fully unrolled loops), so each expands to one of its branches; the two sides of PXR_GET are rvalues on purpose - as
// lvalues the conditional becomes a select between a private and an LDS pointer and the whole array stays in scratch memory.
#define PXR_GET(i) ((i) < PXR_REG ? +S[(i) < PXR_REG ? (i) : 0] : +sl[(i) < PXR_REG ? 0 : (i) - PXR_REG])
#define PXR_PUT(i, v) do { if ((i) < PXR_REG) S[(i) < PXR_REG ? (i) : 0] = (v); else sl[(i) < PXR_REG ? 0 : (i) - PXR_REG] = (v); } while (0)

// One line, in place: patch_reference.py, _filter_lines (the line already carries the gain).  `emit(i, v)` takes the results of the
// backward pass (i = 219 .. 0).
template <class Emit>
__device__ __forceinline__ void pxr_filter_line(double (&S)[PXR_REG], double *sl, Emit emit) {
    const double z = PX_Z, zn = PX_ZN;
    double c0 = PXR_GET(0) + zn * PXR_GET(PX_CROP - 1), zi = z;
#pragma unroll
    for (int i = 1; i < PX_CROP - 1; ++i) {
        c0 = c0 + zi * (PXR_GET(i) + zn * PXR_GET(PX_CROP - 1 - i));
        zi *= z;
        if (i % PXR_SB == 0) __builtin_amdgcn_sched_barrier(0);
    }
    double v = c0 / (1.0 - zn * zn), prev = 0.0;
    PXR_PUT(0, v);
#pragma unroll
    for (int i = 1; i < PX_CROP; ++i) {
        prev = v;
        v = PXR_GET(i) + z * v;
        PXR_PUT(i, v);
        if (i % PXR_SB == 0) __builtin_amdgcn_sched_barrier(0);
    }
    v = (z * prev + v) * z / (z * z - 1.0);
    emit(PX_CROP - 1, v);
    // (in blocks of PXR_TB steps: the unroller has a size limit - a body of 219 large steps gets unrolled in part only, and then the
    // register array stays in scratch memory)
#pragma unroll
    for (int m = (PX_CROP - 1) / PXR_TB; m >= 0; --m) {
#pragma unroll
        for (int k = PXR_TB - 1; k >= 0; --k) {
            const int i = PXR_TB * m + k;
            if (i <= PX_CROP - 2) {
                v = z * (v - PXR_GET(i));
                emit(i, v);
            }
            if (k == PXR_SB) __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// The array's transposition between the two axes, on the chip.  After axis 0 lane l of wave W holds column 55 W + l (row i in slot
// i); axis 1 wants row 55 W + l there (column j in slot j).  Seen as 4 x 4 blocks of 55 lanes x 55 slots, block (W, P) of wave W
// and block (P, W) of wave P swap, each transposed: both waves publish their block in LDS (xb[.][lane][slot], odd pitch), a
// barrier, each reads the other's with lane and slot exchanged.  Two 55 x 55 buffers fit beside the lines' LDS part, so one pair of
// waves (or two diagonal blocks) moves at a time: 8 steps.  W is a template parameter: every slot index is a constant.  Every value
// passes exactly once: it is scaled by the gain on the way (the second pass scales its input too).
#define PXR_XPITCH 57
#define PXR_XROWS 56                            // 55 lanes + a row the idle lanes write to
template <int W>
__device__ __forceinline__ void pxr_transpose(double (&S)[PXR_REG], double *sl, double *xb, int lrow, int lcol, double gain) {
    constexpr int sched[8][2] = {{0, 1}, {2, 3}, {0, 2}, {1, 3}, {0, 3}, {1, 2}, {0, 1}, {2, 3}};     // the last two: diagonal blocks of waves a, b
#pragma unroll
    for (int step = 0; step < 8; ++step) {
        const int a = sched[step][0], b = sched[step][1];
        const bool in = W == a || W == b, diag = step >= 6;
        const int P = diag ? W : (W == a ? b : a);                    // the wave whose block range of slots moves
        double *mine = xb + (W == a ? 0 : 1) * (PXR_XROWS * PXR_XPITCH);
        double *other = diag ? mine : xb + (W == a ? 1 : 0) * (PXR_XROWS * PXR_XPITCH);
        if (in) {
#pragma unroll
            for (int q = 0; q < PXR_LANES; ++q) mine[lrow * PXR_XPITCH + q] = PXR_GET(PXR_LANES * P + q);
        }
        __syncthreads();
        if (in) {
#pragma unroll
            for (int q = 0; q < PXR_LANES; ++q) PXR_PUT(PXR_LANES * P + q, other[q * PXR_XPITCH + lcol] * gain);
        }
        __syncthreads();
    }
}

// What a wave does once the crop's bits are staged, compiled once per wave index (pxr_transpose: its slot indices depend on the
// wave).  The four instances never meet again - behind a common continuation the register allocator has to reconcile four
// assignments of the 161 register-resident values and does it through scratch memory (measured: + 24 us per car).
#ifdef PXR_STAMPS      // (tools/ubench/exact_latency.hip: the clock at the phase boundaries of each wave, lane 0, into `stamps`)
#define PXR_STAMP(k) do { __builtin_amdgcn_sched_barrier(0); if (lane == 0) stamps[8 * W + (k)] = clock64(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define PXR_STAMP(k) do { } while (0)
#endif
template <int W>
__device__ __forceinline__ void pxr_wave(const uint32_t (*cropw)[9], double (*slds)[PXR_LDS | 1], double *xb, double *colmaj, long long *stamps, int gx0, int gxw, int lane) {
    (void)stamps;
    const double z = PX_Z, gain = (1.0 - 1.0 / z) * (1.0 - z);
    const bool act = lane < PXR_LANES;
    const int n = act ? W * PXR_LANES + lane : PX_CROP + W;      // the lane's line (idle lanes share a dummy line)
    double *sl = slds[n];
    double S[PXR_REG];
    PXR_STAMP(0);
    // ---- axis 0: the line = column n of the crop, filtered in place
    {
        const int bitpos = (gx0 + (act ? n : 0)) - gxw * 32, wq = bitpos >> 5, sh = bitpos & 31;
        uint32_t colbits[(PX_CROP + 31) / 32];                         // the column's 220 cells, 32 LDS reads in flight at a time
#pragma unroll
        for (int wi = 0; wi < (PX_CROP + 31) / 32; ++wi) {
            uint32_t w32[32];
#pragma unroll
            for (int b = 0; b < 32; ++b) w32[b] = (32 * wi + b < PX_CROP) ? cropw[32 * wi + b < PX_CROP ? 32 * wi + b : 0][wq] : 0u;
            uint32_t acc = 0;
#pragma unroll
            for (int b = 0; b < 32; ++b) acc |= ((w32[b] >> sh) & 1u) << b;
            colbits[wi] = acc;
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int i = 0; i < PX_CROP; ++i) {
            const double f = ((colbits[i >> 5] >> (i & 31)) & 1u) ? gain : 0.0;        // (1.0 * gain, 0.0 * gain)
            PXR_PUT(i, f);
        }
        __builtin_amdgcn_sched_barrier(0);
        PXR_STAMP(1);
        pxr_filter_line(S, sl, [&](int i, double v) { PXR_PUT(i, v); });
        PXR_STAMP(2);
    }
    __syncthreads();          // (every wave is done with the crop bits: the exchange buffers may be written)
    // ---- columns -> rows (scaled by the gain on the way), then axis 1: the line = row n
    {
        const int lrow = act ? lane : PXR_LANES, lcol = act ? lane : 0;
        __builtin_amdgcn_sched_barrier(0);
        PXR_STAMP(3);
        pxr_transpose<W>(S, sl, xb, lrow, lcol, gain);
        PXR_STAMP(4);
        __builtin_amdgcn_sched_barrier(0);
    }
    {
        // element (row n, column i) goes to colmaj[i * 220 + n]: contiguous across lanes.  The whole pass runs under the lane mask
        // of the 55 active lanes (one branch) - a store masked on its own costs a mask save / branch / restore at every step of the
        // chain, and idle lanes storing to a dummy place cost a cache line of write traffic per step
        double *mine = colmaj + n;
        PXR_STAMP(5);
        if (act) pxr_filter_line(S, sl, [&](int i, double v) { mine[(size_t)i * PX_CROP] = v; });
        PXR_STAMP(6);
    }
}

__global__ __launch_bounds__(256, 1) void rc_patch_exact_prefilter_kernel(RcExactParams p) {
    const int car = p.car0 + (int)blockIdx.x, t = (int)threadIdx.x;
    if (px::skip_car(p, car)) return;                                  // (uniform over the workgroup)
    double *colmaj = p.scratch + (size_t)blockIdx.x * PX_CAR_DOUBLES;      // [column][row]
    // LDS: the lines' LDS part [224][59] (odd pitch: the lanes of a step fall into different banks), then the two exchange
    // buffers of the transposition - which the staged crop bits share: they are dead once every lane has gathered its column
    __shared__ double lds[PXR_LINES * (PXR_LDS | 1) + 2 * PXR_XROWS * PXR_XPITCH];
    double (*slds)[PXR_LDS | 1] = reinterpret_cast<double (*)[PXR_LDS | 1]>(lds);
    double *xb = lds + PXR_LINES * (PXR_LDS | 1);
    uint32_t (*cropw)[9] = reinterpret_cast<uint32_t (*)[9]>(xb);
    static_assert(PX_CROP * 9 * 4 <= 2 * PXR_XROWS * PXR_XPITCH * 8, "the crop bits share the exchange buffers");
    int pr, pc;
    px::pixel_of(p, car, pr, pc);
    const int gx0 = (pc - PX_CROP / 2) - p.c0, gxw = gx0 >> 5;
    for (int q = t; q < PX_CROP * 8; q += 256) {
        const int r = q >> 3, k = q & 7, gy = p.r_top - (pr - PX_CROP / 2 + r), gw = gxw + k;
        uint32_t word = 0;
        if ((unsigned)gy < (unsigned)p.h && (unsigned)gw < (unsigned)p.pitch) {
            word = p.drv_words[(size_t)gy * p.pitch + gw];
            const int first = gw * 32;
            if (first + 32 > p.w) word &= first >= p.w ? 0u : (0xffffffffu >> (32 - (p.w - first)));
        }
        cropw[r][k] = word;
    }
    __syncthreads();
    const int lane = t & 63;
    long long *stamps = reinterpret_cast<long long *>(p.patch + (size_t)car * (PX_OUT * PX_OUT));      // (PXR_STAMPS builds only)
    switch (t >> 6) {
        case 0: pxr_wave<0>(cropw, slds, xb, colmaj, stamps, gx0, gxw, lane); break;
        case 1: pxr_wave<1>(cropw, slds, xb, colmaj, stamps, gx0, gxw, lane); break;
        case 2: pxr_wave<2>(cropw, slds, xb, colmaj, stamps, gx0, gxw, lane); break;
        default: pxr_wave<3>(cropw, slds, xb, colmaj, stamps, gx0, gxw, lane); break;
    }
}

// ------------------------------------------------------------------------------------------------- rotate, crop, resize
#define PX_ST 256                              // threads of the sample kernel (320 = a 25 x 25 tile in two passes instead of three: one car alone 7 % faster,
                                               // 2 048 cars 36 % slower - five waves per workgroup do not spread evenly over four SIMDs)
// The sampling pass decides a pixel by a binary32 ESTIMATE of the interpolated value and computes the library's binary64 sum only
// where the estimate cannot decide.  The pixel is floor(tv + 0.5), clamped; the estimate's error is bounded: a spline coefficient
// of an image with values in [0, 1] is at most 9 in magnitude (the prefilter's impulse response 1.732 x 0.268^|k| sums to 3 per
// axis; measured: 1.87), its binary32 copy is off by <= 9 x 6e-8, a binary32 weight by <= 1e-6 (eight operations on values <= 6,
// the fraction rounded to binary32) and the one got by subtraction by <= 2e-6, the exact weights of an axis sum to 1, and 20 fused
// multiply-adds on partial sums <= 9 add <= 1.1e-5: |estimate - exact| <= 9 x (5e-6 + 5e-6) + 5.4e-7 + 1.1e-5 < 1.1e-4.  PX_BAND
// leaves a factor of 9: outside the band estimate and exact sum lie between the same two integers.  (The patches of 131 072 poses
// on all 32 maps, of the 379 goldens and of every parity test equal the oracle's, which computes every pixel in binary64.)
#define PX_BAND 1.0e-3f
#define PX_RING 64                             // rows of the horizontally resized window kept in LDS (a power of two >= 15 + PX_T)
#define PX_NTILES ((PX_WIN / PX_T) * (PX_WIN / PX_TW))
#define PX_NPRE ((PX_TILE_N * PX_TILE_PITCH + PX_ST - 1) / PX_ST)      // coefficients of a tile per thread
// CHECK (rc_selftest_exact_estimate): every pixel's exact sum is computed beside its estimate; p.check counts the pixels inside the
// array, those the band sent to the exact sum, those the estimate alone would have decided differently (must be 0) and keeps the
// largest |estimate - exact| seen (binary32 bits).
template <bool CHECK>
__global__ __launch_bounds__(PX_ST) void rc_patch_exact_sample_kernel(RcExactParams p) {
    const int car = p.car0 + (int)blockIdx.x, t = (int)threadIdx.x;
    uint8_t *out = p.patch + (size_t)car * (PX_OUT * PX_OUT);
    if (px::skip_car(p, car)) {
        for (int q = t; q < PX_OUT * PX_OUT / 16; q += PX_ST) reinterpret_cast<uint4 *>(out)[q] = make_uint4(0u, 0u, 0u, 0u);
        return;
    }
    // The 200 x 200 window is rendered in tiles of PX_T x PX_TW pixels.  A tile's taps lie in a rectangle of the coefficient array
    // at most 39 x 39 large (the tile rotated, + the 4 x 4 footprint): it is staged in LDS first - coalesced column segments - and
    // the 16 taps per pixel are LDS reads.  Read straight from memory (the first form) a wave's 64 taps of one load instruction lay
    // in up to 64 different cache lines, and the texture-address unit, not the arithmetic, set the pace: 1.51 ms per 2 048 cars;
    // more workgroups per CU made it worse (their 387 KB arrays push each other out of the XCD's L2: EXPERIMENTS.md 000.4).
    // A row of tiles is resized horizontally as soon as it is complete.
    __shared__ float tile[PX_TILE_N * PX_TILE_PITCH];                  // the tile's coefficients, rounded to binary32 (the estimate's operands)
    __shared__ uint8_t win[PX_T * PX_WIN];
    __shared__ uint8_t tmp[PX_RING * PX_OUT];                          // the horizontally resized rows, a ring of PX_RING rows
    __shared__ int32_t kk[PX_OUT * PX_KSIZE];
    __shared__ int32_t bounds[PX_OUT * 2];
    __shared__ int32_t rects[PX_NTILES][4];
    for (int q = t; q < PX_OUT * PX_KSIZE; q += PX_ST) kk[q] = p.kk[q];
    for (int q = t; q < PX_OUT * 2; q += PX_ST) bounds[q] = p.kk[PX_OUT * PX_KSIZE + q];
    const double *coef = p.scratch + (size_t)blockIdx.x * PX_CAR_DOUBLES;      // [column][row]
    // the rotation of scipy.ndimage.rotate(reshape=True) on a 220 x 220 input (patch_reference.py, rotation)
    double cs, sn;
    px::sincos_deg((2.0 * 3.141592653589793 - (double)p.theta[car]) * (180.0 / 3.141592653589793), cs, sn);
    const double n = (double)PX_CROP;
    const double b0[4] = {cs * 0.0 + sn * 0.0, cs * 0.0 + sn * n, cs * n + sn * 0.0, cs * n + sn * n};
    const double b1[4] = {-sn * 0.0 + cs * 0.0, -sn * 0.0 + cs * n, -sn * n + cs * 0.0, -sn * n + cs * n};
    double lo0 = b0[0], hi0 = b0[0], lo1 = b1[0], hi1 = b1[0];
    for (int k = 1; k < 4; ++k) {
        lo0 = b0[k] < lo0 ? b0[k] : lo0; hi0 = b0[k] > hi0 ? b0[k] : hi0;
        lo1 = b1[k] < lo1 ? b1[k] : lo1; hi1 = b1[k] > hi1 ? b1[k] : hi1;
    }
    const int S0 = (int)((hi0 - lo0) + 0.5), S1 = (int)((hi1 - lo1) + 0.5);
    const double h0 = (double)(S0 - 1) / 2, h1 = (double)(S1 - 1) / 2;
    const double off0 = (double)(PX_CROP - 1) / 2 - (cs * h0 + sn * h1), off1 = (double)(PX_CROP - 1) / 2 - (-sn * h0 + cs * h1);
    const int i0 = S0 / 2 - PX_WIN / 2, j0 = S1 / 2 - PX_WIN / 2;
    auto src0 = [&](int i, int j) { return ((0.0 + (double)(i0 + i) * cs) + (double)(j0 + j) * sn) + off0; };        // the library's expression
    auto src1 = [&](int i, int j) { return ((0.0 + (double)(i0 + i) * (-sn)) + (double)(j0 + j) * cs) + off1; };
    // the rectangle of coefficients a tile's taps can touch: the map is linear, so its extremes are at the tile's corners; two cells
    // of margin beyond the 4 x 4 footprint (rounding of the corner values, mirrored taps at the array's edges)
    auto rect = [&](int ti, int tj, int &r_lo, int &c_lo, int &nr, int &nc, bool &edge) -> bool {
        const int ia = ti * PX_T, ib = ia + PX_T - 1, ja = tj * PX_TW, jb = ja + PX_TW - 1;
        const double r00 = src0(ia, ja), r01 = src0(ia, jb), r10 = src0(ib, ja), r11 = src0(ib, jb);
        const double c00 = src1(ia, ja), c01 = src1(ia, jb), c10 = src1(ib, ja), c11 = src1(ib, jb);
        const double rmin = fmin(fmin(r00, r01), fmin(r10, r11)), rmax = fmax(fmax(r00, r01), fmax(r10, r11));
        const double cmin = fmin(fmin(c00, c01), fmin(c10, c11)), cmax = fmax(fmax(c00, c01), fmax(c10, c11));
        const bool none = rmax < -0.5 || rmin > PX_CROP - 0.5 || cmax < -0.5 || cmin > PX_CROP - 0.5;      // every pixel of the tile reads the constant 0
        int r_hi = (int)floor(rmax) + 4, c_hi = (int)floor(cmax) + 4;
        r_lo = (int)floor(rmin) - 3; c_lo = (int)floor(cmin) - 3;
        edge = r_lo < 0 || c_lo < 0 || r_hi > PX_CROP - 1 || c_hi > PX_CROP - 1;      // a tap of this tile may be a mirrored one
        r_lo = r_lo < 0 ? 0 : r_lo; c_lo = c_lo < 0 ? 0 : c_lo;
        r_hi = r_hi > PX_CROP - 1 ? PX_CROP - 1 : r_hi; c_hi = c_hi > PX_CROP - 1 ? PX_CROP - 1 : c_hi;
        nr = r_hi - r_lo + 1; nc = c_hi - c_lo + 1;
        return !none && nr > 0 && nc > 0 && nr <= PX_TILE_N && nc <= PX_TILE_N;      // (always, by the bound above; else: straight from memory)
    };
    // A tile's coefficients travel memory -> registers -> LDS, and the registers are filled for tile k + 1 BEFORE the pixels of tile
    // k are computed: the memory latency (half of a tile's time when a car has the CU alone) hides behind the arithmetic.
    double pre[PX_NPRE];
    auto fetch = [&](bool staged, int r_lo, int c_lo, int nr, int nc) {
#pragma unroll
        for (int k = 0; k < PX_NPRE; ++k) {
            const int q = t + PX_ST * k, c = q / PX_TILE_PITCH, r = q - c * PX_TILE_PITCH;
            pre[k] = (staged && c < nc && r < nr) ? coef[(c_lo + c) * PX_CROP + r_lo + r] : 0.0;
        }
    };
    // every tile's rectangle once per car, a tile per thread (computed where it is used - by all 256 threads, the same values - the
    // eight corner evaluations were a sixth of a tile's arithmetic)
    for (int q = t; q < PX_NTILES; q += PX_ST) {
        int a, b, c, d;
        bool edge;
        const bool ok = rect(q / (PX_WIN / PX_TW), q % (PX_WIN / PX_TW), a, b, c, d, edge);
        rects[q][0] = a; rects[q][1] = b; rects[q][2] = ok ? c : 0; rects[q][3] = d | (edge ? 1 << 16 : 0);      // (rows 0: not staged)
    }
    __syncthreads();
    int r_lo = rects[0][0], c_lo = rects[0][1], nr = rects[0][2], nc = rects[0][3] & 0xffff;
    bool staged = nr > 0, edge = (rects[0][3] >> 16) != 0;
    fetch(staged, r_lo, c_lo, nr, nc);
    int next_out = 0;                                                  // the first output row not written yet
    unsigned long long chk_inside = 0, chk_band = 0, chk_wrong = 0;
    float chk_err = 0.0f;
    for (int ti = 0; ti < PX_WIN / PX_T; ++ti) {
        for (int tj = 0; tj < PX_WIN / PX_TW; ++tj) {
            const int ia = ti * PX_T, ja = tj * PX_TW;
            const int r_lo_t = r_lo, c_lo_t = c_lo;
            const bool staged_t = staged, edge_t = edge;
            if (staged_t) {
#pragma unroll
                for (int k = 0; k < PX_NPRE; ++k)
                    if (t + PX_ST * k < PX_TILE_N * PX_TILE_PITCH) tile[t + PX_ST * k] = (float)pre[k];      // [column][row], pitch = PX_TILE_PITCH
            }
            __syncthreads();
            {
                const int tn = ti * (PX_WIN / PX_TW) + tj + 1;
                if (tn < PX_NTILES) {
                    r_lo = rects[tn][0]; c_lo = rects[tn][1]; nr = rects[tn][2]; nc = rects[tn][3] & 0xffff;
                    staged = nr > 0; edge = (rects[tn][3] >> 16) != 0;
                    fetch(staged, r_lo, c_lo, nr, nc);
                }
            }
            for (int q = t; q < PX_T * PX_TW; q += PX_ST) {
                const int il = q / PX_TW, jl = q - il * PX_TW, i = ia + il, j = ja + jl;
                const double cc0 = src0(i, j), cc1 = src1(i, j);
                int pix = 0;
                if (!(cc0 < 0 || cc0 > PX_CROP - 1 || cc1 < 0 || cc1 > PX_CROP - 1)) {
                    const double f0 = floor(cc0), f1 = floor(cc1);
                    const int st0 = (int)f0 - 1, st1 = (int)f1 - 1;
                    bool exact = !staged_t;
                    float est = 0.0f;
                    int est_pix = -1;
                    if (staged_t) {
                        // the ESTIMATE: weights and taps in binary32 from the staged binary32 copy (a third of the exact sum's issue
                        // cycles).  The pixel is floor(tv + 0.5) clamped to 0 .. 255, so the estimate decides it unless tv + 0.5 lies
                        // within PX_BAND of an integer - then, and only then, the lane computes the exact sum below
                        float w0[4], w1[4];
                        px::weights32((float)(cc0 - f0), w0);
                        px::weights32((float)(cc1 - f1), w1);
                        float tvf = 0.0f;
                        if (!edge_t) {                                 // (most tiles: no tap beyond the array, nothing to mirror)
                            const float *t0 = tile + (st1 - c_lo_t) * PX_TILE_PITCH + (st0 - r_lo_t);
#pragma unroll
                            for (int a = 0; a < 4; ++a) {
                                float rs = 0.0f;
#pragma unroll
                                for (int b = 0; b < 4; ++b) rs = __builtin_fmaf(t0[b * PX_TILE_PITCH + a], w1[b], rs);
                                tvf = __builtin_fmaf(rs, w0[a], tvf);
                            }
                        } else {
                            int col[4];
#pragma unroll
                            for (int b = 0; b < 4; ++b) col[b] = (px::mirror(st1 + b) - c_lo_t) * PX_TILE_PITCH - r_lo_t;
#pragma unroll
                            for (int a = 0; a < 4; ++a) {
                                const int row = px::mirror(st0 + a);
                                float rs = 0.0f;
#pragma unroll
                                for (int b = 0; b < 4; ++b) rs = __builtin_fmaf(tile[col[b] + row], w1[b], rs);
                                tvf = __builtin_fmaf(rs, w0[a], tvf);
                            }
                        }
                        const float u = tvf + 0.5f, fl = floorf(u);
                        exact = u - fl < PX_BAND || (fl + 1.0f) - u < PX_BAND;
                        pix = fl < 0.0f ? 0 : (fl > 255.0f ? 255 : (int)fl);
                        if (CHECK) { est = tvf; est_pix = pix; chk_inside += 1; chk_band += exact ? 1 : 0; }
                    }
                    if (exact || CHECK) {                                       // the library's sum, operation by operation, from the binary64 array
                        double w0[4], w1[4];
                        px::weights(cc0, w0);
                        px::weights(cc1, w1);
                        int col[4];
#pragma unroll
                        for (int b = 0; b < 4; ++b) col[b] = px::mirror(st1 + b) * PX_CROP;
                        double tv = 0.0;
#pragma unroll
                        for (int a = 0; a < 4; ++a) {
                            const int row = px::mirror(st0 + a);
#pragma unroll
                            for (int b = 0; b < 4; ++b) tv = tv + (coef[col[b] + row] * w0[a]) * w1[b];
                        }
                        if (CHECK && est_pix >= 0) chk_err = fmaxf(chk_err, fabsf(est - (float)tv));
                        tv = tv > 0 ? tv + 0.5 : 0.0;
                        pix = (int)(uint8_t)(tv > 255.0 ? 255.0 : tv);
                        if (CHECK && est_pix >= 0 && !exact && est_pix != pix) chk_wrong += 1;
                    }
                }
                win[il * PX_WIN + j] = (uint8_t)pix;
            }
            __syncthreads();          // (the next tile's staging overwrites `tile`)
        }
        // Pillow's 8-bit resize, horizontal pass, this row of tiles: PX_T rows x 64 columns
        for (int q = t; q < PX_T * PX_OUT; q += PX_ST) {
            const int r = q / PX_OUT, xx = q - r * PX_OUT;
            const int x0 = bounds[2 * xx], xm = bounds[2 * xx + 1];
            int32_t acc = 1 << (PX_BITS - 1);
            for (int k = 0; k < xm; ++k) acc += (int32_t)win[r * PX_WIN + x0 + k] * kk[xx * PX_KSIZE + k];
            acc >>= PX_BITS;
            tmp[((ti * PX_T + r) & (PX_RING - 1)) * PX_OUT + xx] = (uint8_t)(acc < 0 ? 0 : (acc > 255 ? 255 : acc));
        }
        __syncthreads();
        // ... vertical pass: the output rows whose taps are all there now (an output row reaches back <= 15 rows, a row of tiles
        // adds 25: the ring's 64 rows hold what is still needed - 12.8 KB of LDS less, a fifth workgroup per CU); each byte of the
        // patch is written once
        int lim = next_out;
        while (lim < PX_OUT && bounds[2 * lim] + bounds[2 * lim + 1] <= (ti + 1) * PX_T) ++lim;       // (uniform)
        for (int q = t; q < (lim - next_out) * PX_OUT; q += PX_ST) {
            const int yy = next_out + q / PX_OUT, xx = q % PX_OUT;
            const int y0 = bounds[2 * yy], ym = bounds[2 * yy + 1];
            int32_t acc = 1 << (PX_BITS - 1);
            for (int k = 0; k < ym; ++k) acc += (int32_t)tmp[((y0 + k) & (PX_RING - 1)) * PX_OUT + xx] * kk[yy * PX_KSIZE + k];
            acc >>= PX_BITS;
            out[yy * PX_OUT + xx] = (uint8_t)(acc < 0 ? 0 : (acc > 255 ? 255 : acc));
        }
        next_out = lim;
    }
    if (CHECK) {
        if (chk_inside) atomicAdd(p.check + 0, chk_inside);
        if (chk_band) atomicAdd(p.check + 1, chk_band);
        if (chk_wrong) atomicAdd(p.check + 2, chk_wrong);
        atomicMax(p.check + 3, (unsigned long long)__float_as_uint(chk_err));       // (non-negative floats order like their bits)
    }
}

// x / 6.0 by the hardware's division against px::div6, bit for bit, over `n` operands per lane drawn by a 64-bit LCG from
// [2^-160, 8) with either sign (exponent uniform, significand uniform): the count of disagreements (rc_selftest_div6)
__global__ __launch_bounds__(256) void rc_selftest_div6_kernel(unsigned long long seed, int n, unsigned long long *mismatches) {
    unsigned long long s = seed + 0x9e3779b97f4a7c15ull * (unsigned long long)(blockIdx.x * blockDim.x + threadIdx.x + 1u);
    unsigned long long bad = 0;
    for (int k = 0; k < n; ++k) {
        s = s * 6364136223846793005ull + 1442695040888963407ull;
        const unsigned long long mant = s >> 12;
        const unsigned long long e = 1023ull - 160ull + ((s >> 3) & 0xffull) % 163ull;          // 2^-160 .. 2^2
        const unsigned long long bits = ((s & 1ull) << 63) | (e << 52) | mant;
        const double x = __longlong_as_double((long long)bits);
        const double a = x / 6.0, b = px::div6(x);
        bad += __double_as_longlong(a) != __double_as_longlong(b) ? 1ull : 0ull;
    }
    if (bad) atomicAdd(mismatches, bad);
}
