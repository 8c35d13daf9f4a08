// obs_type lidar_occupancy_reference: the reference's OccupancyMapObs.step (dreamer/wrappers.py:396-406: to_pixel, 220 x 220
// crop, scipy.ndimage.rotate, centre 200 x 200, PIL resize to 64 x 64) restated down to the binary64 operation.
// The spec - with the library line every step follows - is oracle/patch_reference.py (render_patch_exact); the C oracle's
// oc_patch_exact_range and the two kernels here are the same arithmetic, operation by operation (IEEE binary64, no fused
// multiply-add: the library is built with -ffp-contract=off; integer arithmetic for Pillow's resize).
//
//   rc_patch_exact_prefilter_kernel   one 256-thread workgroup per car: the crop's cubic-spline coefficients, 220 x 220 binary64
//       (387 KB per car: global scratch, two buffers per car of the chunk - no LDS holds it).  Axis 0 with one lane per COLUMN
//       (the column's 220 bits in LDS words; the causal start value summed from the bits in registers; forward values stored row-major
//       = coalesced across lanes; the backward pass writes its results TRANSPOSED, four rows per lane at a time), then axis 1 with
//       one lane per ROW reading the transposed array = coalesced again; the result stays in [column][row] order.
//   rc_patch_exact_sample_kernel      one 256-thread workgroup per car: the 200 x 200 centre window of the rotated image (16 taps
//       per pixel from the coefficient array through L1 / L2, weights and sums in binary64, rounded to uint8 as the library does)
//       into LDS, then Pillow's two integer passes (200 x 200 -> 200 x 64 -> 64 x 64) from LDS, the patch written once.
// A recurrence of 220 dependent steps per line cannot be sped up by a scan without changing the order of the additions; what
// hides its latency is other cars: two workgroups (eight waves) per SIMD, every load of the backward passes issued ahead.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "racecar_internal.h"       // RcExactParams

#define PX_CROP 220
#define PX_WIN 200
#define PX_OUT 64
#define PX_KSIZE 15
#define PX_BITS 22
#define PX_T 25                               // the sample kernel renders the 200 x 200 window in 8 x 8 tiles of 25 x 25 pixels (40: 11 % slower,
                                              // 20: 2 %, 10: 40 % - tools/exact_tile_sweep.sh)
#define PX_TW 20                              // ... of PX_T rows x PX_TW columns: 500 pixels = two passes of 256 threads at 98 % (25 x 25: three at 81 %)
#define PX_TILE_N 39                          // a tile's taps lie within sqrt(24^2 + 19^2) + 9 = 39.6 coefficients per axis
#define PX_TILE_PITCH 39                      // doubles per staged column (odd: consecutive columns start in different LDS banks)
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

__device__ __forceinline__ int mirror(int i) {
    i = i < 0 ? -i : i;
    return i >= PX_CROP ? 2 * PX_CROP - 2 - i : i;
}

}  // namespace px

// ---------------------------------------------------------------------------------------------------------------- prefilter
// One line of the cubic-spline prefilter, IN PLACE in LDS, by one lane (patch_reference.py, _filter_lines; the line already
// carries the gain): the mirror-summed causal start, the causal recursion, the anticausal start and recursion.  The LDS reads of a
// block of PX_BLK steps are issued together ahead of the block's dependent chain (their addresses do not depend on it).
#define PX_BLK 11                              // 220 = 20 x 11
__device__ __forceinline__ void px_filter_line_lds(double *line) {
    const double z = PX_Z, zn = PX_ZN;
    double c0 = line[0] + zn * line[PX_CROP - 1], zi = z;
    // i = 1 .. 218: two steps alone, then 18 blocks of 12
    for (int i = 1; i < 3; ++i) {
        c0 = c0 + zi * (line[i] + zn * line[PX_CROP - 1 - i]);
        zi *= z;
    }
    for (int i0 = 3; i0 < PX_CROP - 1; i0 += 12) {
        double f[12], g[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) { f[k] = line[i0 + k]; g[k] = line[PX_CROP - 1 - i0 - k]; }
#pragma unroll
        for (int k = 0; k < 12; ++k) {
            c0 = c0 + zi * (f[k] + zn * g[k]);
            zi *= z;
        }
    }
    double v = c0 / (1.0 - zn * zn), prev = 0.0;
    line[0] = v;
    // forward, i = 1 .. 219: 19 blocks of 11, then 10 steps
    for (int i0 = 1; i0 < PX_CROP; i0 += PX_BLK) {
        double f[PX_BLK];
#pragma unroll
        for (int k = 0; k < PX_BLK; ++k) f[k] = (i0 + k < PX_CROP) ? line[i0 + k] : 0.0;
#pragma unroll
        for (int k = 0; k < PX_BLK; ++k)
            if (i0 + k < PX_CROP) {
                prev = v;
                v = f[k] + z * v;
                line[i0 + k] = v;
            }
    }
    v = (z * prev + v) * z / (z * z - 1.0);
    line[PX_CROP - 1] = v;
    // backward, i = 218 .. 0: 19 blocks of 11, then 10 steps
    for (int i0 = PX_CROP - 2; i0 >= 0; i0 -= PX_BLK) {
        double f[PX_BLK];
#pragma unroll
        for (int k = 0; k < PX_BLK; ++k) f[k] = (i0 - k >= 0) ? line[i0 - k] : 0.0;
#pragma unroll
        for (int k = 0; k < PX_BLK; ++k)
            if (i0 - k >= 0) {
                v = z * (v - f[k]);
                line[i0 - k] = v;
            }
    }
}

// The crop's spline coefficients, 220 x 220 binary64, into the car's scratch in [column][row] order.  Every recursion runs in LDS:
// each of the workgroup's four waves takes eight lines at a time - all 64 lanes fill them (axis 0: from the crop's bits, staged
// once per car; axis 1: from the axis-0 result in memory, scaled by the gain), lanes 0 - 7 filter a line each in place, all 64
// lanes write the lines out - so a pass costs one read and one write of the array (axis 0: the write alone) where the first form
// of this kernel (forward values through memory) moved 3.5 MB per car at 4.6 TB/s of HBM traffic.  64 lines are in flight per CU
// (160 KB of LDS hold 92); the waves run out of step with one another, so loads, recursions and stores of different waves overlap.
#define PX_NBW 10
#define PX_PF_WAVES 4                           // waves per workgroup of the prefilter kernel (PX_PF_WAVES x PX_NBW lines in LDS at a time)
#define PX_LPITCH 221
#define PX_RPW ((PX_CROP + 63) / 64)             // passes of the wave over a line (axis 0)
#define PX_IPW (64 / PX_NBW)                    // columns of a batch one pass of the wave covers (axis 1)
#define PX_NIT ((PX_CROP + PX_IPW - 1) / PX_IPW) // passes per batch
__global__ __launch_bounds__(64 * PX_PF_WAVES, 2) void rc_patch_exact_prefilter_kernel(RcExactParams p) {
    const int car = p.car0 + (int)blockIdx.x, t = (int)threadIdx.x;
    if (px::skip_car(p, car)) return;                                  // (uniform over the workgroup)
    double *colmaj = p.scratch + (size_t)blockIdx.x * PX_CAR_DOUBLES + PX_CROP * PX_CROP;      // [c][r]
    __shared__ double lines[PX_PF_WAVES][PX_NBW][PX_LPITCH];
    __shared__ uint32_t cropw[PX_CROP][9];                             // the crop's rows as bits: bit k of the row = cell column gxw * 32 + k
    const double z = PX_Z, gain = (1.0 - 1.0 / z) * (1.0 - z);
    (void)z;
    int pr, pc;
    px::pixel_of(p, car, pr, pc);
    const int gx0 = (pc - PX_CROP / 2) - p.c0;                        // grid column of crop column 0 (may be negative)
    const int gxw = gx0 >> 5;                                          // its word (floor)
    for (int q = t; q < PX_CROP * 8; q += 64 * PX_PF_WAVES) {
        const int r = q >> 3, k = q & 7, gy = p.r_top - (pr - PX_CROP / 2 + r), gw = gxw + k;
        uint32_t word = 0;
        if ((unsigned)gy < (unsigned)p.h && (unsigned)gw < (unsigned)p.pitch) {
            word = p.drv_words[(size_t)gy * p.pitch + gw];
            const int first = gw * 32;                                 // cells at or beyond the grid's width read 0
            if (first + 32 > p.w) word &= first >= p.w ? 0u : (0xffffffffu >> (32 - (p.w - first)));
        }
        cropw[r][k] = word;
    }
    __syncthreads();
    const int wave = t >> 6, lane = t & 63;
    double (*mine)[PX_LPITCH] = lines[wave];
    // ---- axis 0: a line = a column of the crop (north-up: crop row r is grid row r_top - (pr - 110 + r))
    for (int b = wave; b * PX_NBW < PX_CROP; b += PX_PF_WAVES) {
        const int c0 = b * PX_NBW;
#pragma unroll
        for (int l = 0; l < PX_NBW; ++l) {
            const int bitpos = (gx0 + c0 + l) - gxw * 32;              // (uniform over the wave)
#pragma unroll
            for (int k = 0; k < PX_RPW; ++k) {
                const int r = lane + 64 * k;
                if (r < PX_CROP) {
                    const uint32_t bit = (c0 + l < PX_CROP) ? (cropw[r][bitpos >> 5] >> (bitpos & 31)) & 1u : 0u;
                    mine[l][r] = bit ? gain : 0.0;                     // (1.0 * gain, 0.0 * gain)
                }
            }
        }
        __builtin_amdgcn_wave_barrier();                               // (one wave: its own LDS writes, in order)
#ifndef PX_EXP_NO_CHAIN
        if (lane < PX_NBW && c0 + lane < PX_CROP) px_filter_line_lds(mine[lane]);
#endif
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int l = 0; l < PX_NBW; ++l)
#pragma unroll
            for (int k = 0; k < PX_RPW; ++k) {
                const int r = lane + 64 * k;
                if (r < PX_CROP && c0 + l < PX_CROP) colmaj[(size_t)(c0 + l) * PX_CROP + r] = mine[l][r];
            }
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();          // every column's result is visible to the whole workgroup (global memory, same CU)
    // ---- axis 1: a line = a row; element (row r, column i) lies at colmaj[i * 220 + r].  The next batch's rows are fetched into
    // registers BEFORE this batch's recursions start (they are other rows than the ones this batch writes), so the memory latency
    // of a fill hides behind the dependent chains instead of standing in front of them.
    double pre[PX_NIT];
    const int li = lane / PX_NBW, ll = lane - li * PX_NBW;            // lane = (column within a pass of PX_IPW columns, line)
    auto fetch = [&](int r0) {
        const double *src = colmaj + (size_t)li * PX_CROP + r0 + ll;
#pragma unroll
        for (int k = 0; k < PX_NIT; ++k)
            pre[k] = (li < PX_IPW && li + PX_IPW * k < PX_CROP && r0 + ll < PX_CROP) ? src[(size_t)k * (PX_IPW * PX_CROP)] : 0.0;
    };
    if (wave * PX_NBW < PX_CROP) fetch(wave * PX_NBW);
    for (int b = wave; b * PX_NBW < PX_CROP; b += PX_PF_WAVES) {
        const int r0 = b * PX_NBW;
#pragma unroll
        for (int k = 0; k < PX_NIT; ++k)
            if (li < PX_IPW && li + PX_IPW * k < PX_CROP) mine[ll][li + PX_IPW * k] = pre[k] * gain;      // the second pass scales its input too
        __builtin_amdgcn_wave_barrier();
        if ((b + PX_PF_WAVES) * PX_NBW < PX_CROP) fetch((b + PX_PF_WAVES) * PX_NBW);
#ifndef PX_EXP_NO_CHAIN
        if (lane < PX_NBW && r0 + lane < PX_CROP) px_filter_line_lds(mine[lane]);
#endif
        __builtin_amdgcn_wave_barrier();
        {
            double *dst = colmaj + (size_t)li * PX_CROP + r0 + ll;
#pragma unroll
            for (int k = 0; k < PX_NIT; ++k)
                if (li < PX_IPW && li + PX_IPW * k < PX_CROP && r0 + ll < PX_CROP) dst[(size_t)k * (PX_IPW * PX_CROP)] = mine[ll][li + PX_IPW * k];
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ------------------------------------------------------------------------------------------ prefilter, lines in registers
// The same arithmetic with a line per LANE: 4 waves x 55 lanes = the 220 lines of an axis at once, a line's 220 values in the lane's
// registers (PXR_REG of them) and LDS (the rest) - what bounds the LDS form is lines in flight per CU (80) over the 45 cycles a
// step of a recursion takes (tools/ubench/f64_chain.hip), and a CU's registers hold three times what its LDS holds.  Every loop over
// a line is fully unrolled, so every register index is a constant.  Axis 0 (lane = column): the column's bits from the staged crop,
// results of the backward pass through a 55 x 16 LDS tile per wave so that they reach memory [column][row] in 128-byte runs.
// Axis 1 (lane = row): [column][row] is contiguous across lanes, so the 220 loads of the fill go out back to back and the backward
// pass stores from registers.
#define PX_PREFILTER_REG 1
#define PXR_REG 161
#define PXR_LDS (PX_CROP - PXR_REG)
#define PXR_LANES 55
#define PXR_TB 16
#define PXR_TPITCH 17
#define PXR_SB 8                                // the scheduler may move code within blocks of this many steps only (else it
                                                // hoists hundreds of loads and spills)
#define PXR_LINES (PX_CROP + 4)                 // + one dummy line per wave for the idle lanes 55 .. 63

template <class Emit, class Block>
__device__ __forceinline__ void pxr_filter_line(double (&S)[PXR_REG], double *sl, Emit emit, Block block) {
    const double z = PX_Z, zn = PX_ZN;
#define PXR_GET(i) ((i) < PXR_REG ? +S[(i) < PXR_REG ? (i) : 0] : +sl[(i) < PXR_REG ? 0 : (i) - PXR_REG])      // (rvalues: no pointer select)
#define PXR_PUT(i, v) do { if ((i) < PXR_REG) S[(i) < PXR_REG ? (i) : 0] = (v); else sl[(i) < PXR_REG ? 0 : (i) - PXR_REG] = (v); } while (0)
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
    // (blocks of PXR_TB steps: a loop whose body held the block's work at every step would exceed the unroller's size limit, be
    // unrolled in part only, and leave the register array in scratch memory)
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
        block(PXR_TB * m);
        __builtin_amdgcn_sched_barrier(0);
    }
}

__global__ __launch_bounds__(256, 1) void rc_patch_exact_prefilter_reg_kernel(RcExactParams p) {
    const int car = p.car0 + (int)blockIdx.x, t = (int)threadIdx.x;
    if (px::skip_car(p, car)) return;                                  // (uniform over the workgroup)
    double *colmaj = p.scratch + (size_t)blockIdx.x * PX_CAR_DOUBLES + PX_CROP * PX_CROP;      // [c][r]
    __shared__ double slds[PXR_LINES][PXR_LDS | 1];                    // (odd pitch: the lanes of a step fall into different banks)
    __shared__ double tile[4][64][PXR_TPITCH];
    __shared__ uint32_t cropw[PX_CROP][9];
    const double z = PX_Z, gain = (1.0 - 1.0 / z) * (1.0 - z);
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
    const int wave = t >> 6, lane = t & 63;
    const bool act = lane < PXR_LANES;
    const int n = act ? wave * PXR_LANES + lane : PX_CROP + wave;      // the lane's line (idle lanes share a dummy line)
    double *sl = slds[n];
    double (*tl)[PXR_TPITCH] = tile[wave];
    double S[PXR_REG];
    // ---- axis 0: the line = column n of the crop
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
        const int cq = lane >> 4, k = lane & 15;
        pxr_filter_line(S, sl, [&](int i, double v) { tl[lane][i & (PXR_TB - 1)] = v; },
            [&](int i) {                                               // rows i .. i + 15 of the wave's 55 columns are complete
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int pass = 0; pass < (PXR_LANES + 3) / 4; ++pass) {
                    const int cl = 4 * pass + cq;
                    if (cl < PXR_LANES && i + k < PX_CROP) colmaj[(size_t)(wave * PXR_LANES + cl) * PX_CROP + i + k] = tl[cl][k];
                }
                __builtin_amdgcn_wave_barrier();
            });
    }
    __syncthreads();          // every column's result is visible to the whole workgroup (global memory, same CU)
    // ---- axis 1: the line = row n; element (row n, column i) lies at colmaj[i * 220 + n]
    {
        double *mine = colmaj + (act ? n : wave * PXR_LANES);          // (idle lanes read a neighbour's row - no branch around the loads,
                                                                       //  or every one of them gets a wait of its own - and store nothing)
#pragma unroll
        for (int i = PXR_REG; i < PX_CROP; ++i) PXR_PUT(i, mine[(size_t)i * PX_CROP] * gain);      // the second pass scales its input too
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < PXR_REG; ++i) S[i] = mine[(size_t)i * PX_CROP];                        // 161 loads in flight, then
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < PXR_REG; ++i) S[i] = S[i] * gain;
        __builtin_amdgcn_sched_barrier(0);
        pxr_filter_line(S, sl, [&](int i, double v) { if (act) mine[(size_t)i * PX_CROP] = v; }, [](int) {});
    }
}

// ------------------------------------------------------------------------------------------------- rotate, crop, resize
#define PX_ST 256                              // threads of the sample kernel (320 = a 25 x 25 tile in two passes instead of three: one car alone 7 % faster,
                                               // 2 048 cars 36 % slower - five waves per workgroup do not spread evenly over four SIMDs)
#define PX_NPRE ((PX_TILE_N * PX_TILE_PITCH + PX_ST - 1) / PX_ST)      // coefficients of a tile per thread
__global__ __launch_bounds__(PX_ST) void rc_patch_exact_sample_kernel(RcExactParams p) {
    const int car = p.car0 + (int)blockIdx.x, t = (int)threadIdx.x;
    uint8_t *out = p.patch + (size_t)car * (PX_OUT * PX_OUT);
    if (px::skip_car(p, car)) {
        for (int q = t; q < PX_OUT * PX_OUT / 16; q += PX_ST) reinterpret_cast<uint4 *>(out)[q] = make_uint4(0u, 0u, 0u, 0u);
        return;
    }
    // The 200 x 200 window is rendered in 8 x 8 tiles of PX_T x PX_T pixels.  A tile's taps lie in a rectangle of the coefficient
    // array at most 41 x 41 large (the tile rotated, + the 4 x 4 footprint): it is staged in LDS first - coalesced column segments -
    // and the 16 taps per pixel are LDS reads.  Read straight from memory (the first form) a wave's 64 taps of one load instruction
    // lay in up to 64 different cache lines, and the texture-address unit, not the arithmetic, set the pace: 1.51 ms per 2 048
    // cars; more workgroups per CU made it worse (their 387 KB arrays push each other out of the XCD's L2:
    // tools/exact_strip_sweep.sh).  A row of tiles is resized horizontally as soon as it is complete.
    __shared__ double tile[PX_TILE_N * PX_TILE_PITCH];
    __shared__ uint8_t win[PX_T * PX_WIN];
    __shared__ uint8_t tmp[PX_WIN * PX_OUT];
    __shared__ int32_t kk[PX_OUT * PX_KSIZE];
    __shared__ int32_t bounds[PX_OUT * 2];
    for (int q = t; q < PX_OUT * PX_KSIZE; q += PX_ST) kk[q] = p.kk[q];
    for (int q = t; q < PX_OUT * 2; q += PX_ST) bounds[q] = p.kk[PX_OUT * PX_KSIZE + q];
    const double *coef = p.scratch + (size_t)blockIdx.x * PX_CAR_DOUBLES + PX_CROP * PX_CROP;      // [column][row]
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
    auto rect = [&](int ti, int tj, int &r_lo, int &c_lo, int &nr, int &nc) -> bool {
        const int ia = ti * PX_T, ib = ia + PX_T - 1, ja = tj * PX_TW, jb = ja + PX_TW - 1;
        const double r00 = src0(ia, ja), r01 = src0(ia, jb), r10 = src0(ib, ja), r11 = src0(ib, jb);
        const double c00 = src1(ia, ja), c01 = src1(ia, jb), c10 = src1(ib, ja), c11 = src1(ib, jb);
        const double rmin = fmin(fmin(r00, r01), fmin(r10, r11)), rmax = fmax(fmax(r00, r01), fmax(r10, r11));
        const double cmin = fmin(fmin(c00, c01), fmin(c10, c11)), cmax = fmax(fmax(c00, c01), fmax(c10, c11));
        const bool none = rmax < -0.5 || rmin > PX_CROP - 0.5 || cmax < -0.5 || cmin > PX_CROP - 0.5;      // every pixel of the tile reads the constant 0
        int r_hi = (int)floor(rmax) + 4, c_hi = (int)floor(cmax) + 4;
        r_lo = (int)floor(rmin) - 3; c_lo = (int)floor(cmin) - 3;
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
    int r_lo, c_lo, nr, nc;
    bool staged = rect(0, 0, r_lo, c_lo, nr, nc);
    fetch(staged, r_lo, c_lo, nr, nc);
    for (int ti = 0; ti < PX_WIN / PX_T; ++ti) {
        for (int tj = 0; tj < PX_WIN / PX_TW; ++tj) {
            const int ia = ti * PX_T, ja = tj * PX_TW;
            const int r_lo_t = r_lo, c_lo_t = c_lo;
            const bool staged_t = staged;
            if (staged_t) {
#pragma unroll
                for (int k = 0; k < PX_NPRE; ++k)
                    if (t + PX_ST * k < PX_TILE_N * PX_TILE_PITCH) tile[t + PX_ST * k] = pre[k];      // [column][row], pitch = PX_TILE_PITCH
            }
            __syncthreads();
            {
                const int tn = ti * (PX_WIN / PX_TW) + tj + 1;
                if (tn < (PX_WIN / PX_T) * (PX_WIN / PX_TW)) {
                    staged = rect(tn / (PX_WIN / PX_TW), tn % (PX_WIN / PX_TW), r_lo, c_lo, nr, nc);
                    fetch(staged, r_lo, c_lo, nr, nc);
                }
            }
            for (int q = t; q < PX_T * PX_TW; q += PX_ST) {
                const int il = q / PX_TW, jl = q - il * PX_TW, i = ia + il, j = ja + jl;
                const double cc0 = src0(i, j), cc1 = src1(i, j);
                double tv = 0.0;
                if (!(cc0 < 0 || cc0 > PX_CROP - 1 || cc1 < 0 || cc1 > PX_CROP - 1)) {
                    double w0[4], w1[4];
                    px::weights(cc0, w0);
                    px::weights(cc1, w1);
                    const int st0 = (int)floor(cc0) - 1, st1 = (int)floor(cc1) - 1;
                    if (staged_t) {
                        int col[4];
#pragma unroll
                        for (int b = 0; b < 4; ++b) col[b] = (px::mirror(st1 + b) - c_lo_t) * PX_TILE_PITCH - r_lo_t;
#pragma unroll
                        for (int a = 0; a < 4; ++a) {
                            const int row = px::mirror(st0 + a);
#pragma unroll
                            for (int b = 0; b < 4; ++b) tv = tv + (tile[col[b] + row] * w0[a]) * w1[b];
                        }
                    } else {
                        int col[4];
#pragma unroll
                        for (int b = 0; b < 4; ++b) col[b] = px::mirror(st1 + b) * PX_CROP;
#pragma unroll
                        for (int a = 0; a < 4; ++a) {
                            const int row = px::mirror(st0 + a);
#pragma unroll
                            for (int b = 0; b < 4; ++b) tv = tv + (coef[col[b] + row] * w0[a]) * w1[b];
                        }
                    }
                }
                tv = tv > 0 ? tv + 0.5 : 0.0;
                win[il * PX_WIN + j] = (uint8_t)(tv > 255.0 ? 255.0 : tv);
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
            tmp[ti * PX_T * PX_OUT + q] = (uint8_t)(acc < 0 ? 0 : (acc > 255 ? 255 : acc));
        }
        __syncthreads();
    }
    // ... vertical pass: 64 x 64, written once
    for (int q = t; q < PX_OUT * PX_OUT; q += PX_ST) {
        const int yy = q / PX_OUT, xx = q - yy * PX_OUT;
        const int y0 = bounds[2 * yy], ym = bounds[2 * yy + 1];
        int32_t acc = 1 << (PX_BITS - 1);
        for (int k = 0; k < ym; ++k) acc += (int32_t)tmp[(y0 + k) * PX_OUT + xx] * kk[yy * PX_KSIZE + k];
        acc >>= PX_BITS;
        out[q] = (uint8_t)(acc < 0 ? 0 : (acc > 255 ? 255 : acc));
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
