// THE LAB: scan kernels that are not shipped.  libracecar_lab.so is built on request (racing_dreamer_amd.build.build_lab, which
// tools/ and the variant tests call) and loaded by libracecar_hip.so only when rc_set_raycast_variant(0..6) or
// rc_debug_scan_stamps is used; without it those two calls fail with a message that says how to build it.
//
//   rc_raycast_kernel<A, VARIANT>   the earlier forms of the scan, one lane per ray: 0 cell by cell (the spec's own loop),
//                                   1 / 2 free-block skipping with bitmap + block table in LDS, 3 / 4 the packed block table
//                                   (LDS / global), 5 per-cell distance table, 6 per-cell, per-quadrant rectangles - all
//                                   bit-identical to the shipped scan, which is what tests/test_gpu_parity.py checks
//   rc_raycast_car_stamps_kernel    the shipped scan (scan_car, racecar_scan.h) with shader-clock stamps (tools/scan_stamps.py)
//   rclab_cohabit_kernel            a co-tenant of chosen size for the scan (tools/cohabit_sweep.py: what a collective beside it costs)
//
// Numerics as in racecar_kernels.hip: one IEEE operation per written operator (-ffp-contract=off).
#include <string>

#include "racecar_scan.h"

namespace {

// Exact grid traversal (H3).  Cell boundaries are derived from the integer cell index at every
// step (t = (boundary - origin) * 1/d), so the visited cell sequence and the returned range do
// not depend on how the traversal is scheduled.  The bitmap's outermost ring is set and means
// "no return", so no per-step bounds check is needed.
__device__ __forceinline__ float cast_ray_dda(const uint32_t *bits, const RcTrackDev &t, float gx, float gy,
                                              float dx, float dy) {
    int ix = (int)floorf(gx), iy = (int)floorf(gy);
    if ((unsigned)ix >= (unsigned)t.w || (unsigned)iy >= (unsigned)t.h) return 0.0f;
    if (bit_at(bits, t.pitch, ix, iy)) return 0.0f;
    const float idx = dx != 0.0f ? 1.0f / dx : 0.0f;
    const float idy = dy != 0.0f ? 1.0f / dy : 0.0f;
    const int sx = dx > 0.0f ? 1 : -1, sy = dy > 0.0f ? 1 : -1;
    const float sxf = (float)sx, syf = (float)sy;
    float bx = (float)(ix + (dx > 0.0f ? 1 : 0));
    float by = (float)(iy + (dy > 0.0f ? 1 : 0));
    float tx = dx != 0.0f ? (bx - gx) * idx : INFINITY;
    float ty = dy != 0.0f ? (by - gy) * idy : INFINITY;
    const int wm1 = t.w - 1, hm1 = t.h - 1;
    for (;;) {
        const bool stepx = tx < ty;
        const float tt = stepx ? tx : ty;
        if (tt >= t.tmax) return RCS_MAX_RANGE;
        if (stepx) {
            ix += sx;
            bx += sxf;
            tx = (bx - gx) * idx;
        } else {
            iy += sy;
            by += syf;
            ty = (by - gy) * idy;
        }
        if (bit_at(bits, t.pitch, ix, iy)) {
            const bool ring = ix == 0 || iy == 0 || ix == wm1 || iy == hm1;
            return ring ? RCS_MAX_RANGE : tt * t.res;
        }
    }
}

// Same traversal, but whole certified-free rectangles are crossed in one iteration.
//
// `blk` holds, per (1 << shift)^2-cell block, v = min over the block's cells of the Chebyshev distance
// to the nearest stop cell (0 if the block contains one).  For v >= 1 every cell of the rectangle
// [block - (v-1), block + (v-1)] is free, so the cell-by-cell traversal would walk through it without a
// hit; we jump straight to the crossing that leaves it.  Because the reference traversal derives every
// boundary time from integer boundary coordinates (t = (b - g) * 1/d), the state after that crossing is
// a pure function of the ray: the exit axis is decided by the same comparison (tx < ty, ties -> y), and
// the number of other-axis boundaries crossed before it is the count of j with t_other(j) <= t_exit
// (y before x on ties) resp. < t_exit - found from an fp32 estimate and corrected with the exact
// comparisons, so the visited-cell sequence outside free rectangles, the hit cell and the returned range
// are bit-identical to cast_ray_dda (checked against the CPU oracle in tests/test_gpu_parity.py).
// With v == 0 the rectangle is the current cell and the iteration is exactly one traversal step.
__device__ __forceinline__ float cast_ray_skip(const uint32_t *bits, const uint8_t *blk, const RcTrackDev &t,
                                               float gx, float gy, float dx, float dy) {
    int ix = (int)floorf(gx), iy = (int)floorf(gy);
    if ((unsigned)ix >= (unsigned)t.w || (unsigned)iy >= (unsigned)t.h) return 0.0f;
    if (bit_at(bits, t.pitch, ix, iy)) return 0.0f;
    const bool hx = dx != 0.0f, hy = dy != 0.0f, px = dx > 0.0f, py = dy > 0.0f;
    const float idx = hx ? 1.0f / dx : 0.0f;
    const float idy = hy ? 1.0f / dy : 0.0f;
    const int sx = px ? 1 : -1, sy = py ? 1 : -1;
    const float sxf = (float)sx, syf = (float)sy;
    const int shift = t.blk_shift, bs = 1 << shift, bmask = ~(bs - 1);
    const int wm1 = t.w - 1, hm1 = t.h - 1;
    for (int it = 0; it < 4096; ++it) {          // a ray crosses < 430 cells; the cap only bounds a logic error
        const int v = blk[(iy >> shift) * t.blk_w + (ix >> shift)];
        const int r = v - 1;
        const int x0 = v ? (ix & bmask) - r : ix, x1 = v ? (ix & bmask) + bs + r : ix + 1;
        const int y0 = v ? (iy & bmask) - r : iy, y1 = v ? (iy & bmask) + bs + r : iy + 1;
        const int xe = px ? x1 : x0, ye = py ? y1 : y0;                 // boundaries that leave the rectangle
        const float txe = hx ? ((float)xe - gx) * idx : INFINITY;
        const float tye = hy ? ((float)ye - gy) * idy : INFINITY;
        const bool xexit = txe < tye;
        const float tt = xexit ? txe : tye;
        if (tt >= t.tmax) return RCS_MAX_RANGE;
        if (xexit) {
            int m = 0;
            if (hy && v) {
                const float b0 = (float)(iy + (py ? 1 : 0));            // first y boundary ahead
                m = ((int)floorf(gy + tt * dy) - iy) * sy;
                m = m < 0 ? 0 : m;
                for (int g = 0; g < 8 && ((b0 + (float)m * syf) - gy) * idy <= tt; ++g) ++m;
                for (int g = 0; g < 8 && m > 0 && ((b0 + (float)(m - 1) * syf) - gy) * idy > tt; ++g) --m;
            }
            iy += m * sy;
            ix = px ? x1 : x0 - 1;
        } else {
            int m = 0;
            if (hx && v) {
                const float b0 = (float)(ix + (px ? 1 : 0));
                m = ((int)floorf(gx + tt * dx) - ix) * sx;
                m = m < 0 ? 0 : m;
                for (int g = 0; g < 8 && ((b0 + (float)m * sxf) - gx) * idx < tt; ++g) ++m;
                for (int g = 0; g < 8 && m > 0 && ((b0 + (float)(m - 1) * sxf) - gx) * idx >= tt; ++g) --m;
            }
            ix += m * sx;
            iy = py ? y1 : y0 - 1;
        }
        if (bit_at(bits, t.pitch, ix, iy)) {
            const bool ring = ix == 0 || iy == 0 || ix == wm1 || iy == hm1;
            return ring ? RCS_MAX_RANGE : tt * t.res;
        }
    }
    return RCS_MAX_RANGE;
}
// Variant 2: free-rectangle skipping tuned to the gfx950 VALU issue costs measured on MI355X
// (tools/ubench/valu_issue{2,3}.hip; cycles per wave64 instruction per SIMD, 8 waves resident):
//   ~2.4  v_add/sub/mul/fma_f32, v_add/sub_u32, v_and/or/xor, shifts
//   ~4.3  v_cmp, v_cvt, v_floor, v_min/max, v_bfi, v_bfe, v_add3, v_mad_u32_u24, v_mul_i32_i24, e64 v_cndmask
//   ~16   v_cndmask_b32 e32 reading VCC (what hipcc emits for most `?:`), ~8 v_rcp_f32; SALU ~4.2, overlapping
// The kernel is VALU-issue bound, so the loop is written to minimise instructions:
//  * one body for both kinds of iteration: with block value v == 0 the "rectangle" is the current cell
//    and the iteration is exactly one traversal step;
//  * selections use sign masks and v_bfi (kept opaque to LLVM by one-instruction asm, otherwise they are
//    folded back into compare + select);
//  * the other-axis cell after the exit crossing is floor(g + t * d) whenever that position is at least
//    1e-3 cell away from a cell boundary: the fp32 boundary times the reference traversal compares
//    deviate from the real ones by < 2e-4 cell, so there the count of crossed boundaries is unambiguous
//    under either tie rule; within 1e-3 of a boundary (corner grazing, <1 % of iterations) the exact
//    comparisons of the reference are evaluated;
//  * a certified block (v >= 1) needs no occupancy test, so most iterations read LDS once.
// A direction component that is exactly zero gets the reciprocal 3e38 (finite) and a positive step: its
// boundary times are huge but never NaN/inf and the cell sequence is unchanged.  Bit-identical to
// cast_ray_dda (tests/test_gpu_parity.py::test_raycast_variants_from_arbitrary_poses).
// The exact other-axis cell after the exit crossing at time tt, for the rare iterations in which the fp32
// position estimate lies within 1e-3 cell of a boundary.  `on_est` is that estimate (off by at most one cell),
// oi the current cell, opi 1 / 0 for a positive / negative direction on that axis, og / oid the ray origin and
// reciprocal direction on it.  The reference traversal crosses the other-axis boundary b before the exit iff
// t_b < tt, or t_b == tt when the exit is an x crossing (mx = -1: "ties go to y"); boundary times can be -0.0
// (sensor exactly on a boundary, negative direction), so these are IEEE comparisons, not sign-bit tests.
__device__ __forceinline__ int exact_other_cell(int on_est, int oi, int opi, float og, float oid, float tt, int mx) {
    const int os = 2 * opi - 1;
    const float osf = (float)os;
    const int m0 = max(__mul24(on_est - oi, os) - 1, 0);
    const float b0 = (float)(oi + opi + __mul24(m0, os));
    const float tb0 = (b0 - og) * oid, tb1 = ((b0 + osf) - og) * oid;
    const int c0 = (tb0 < tt || (mx != 0 && tb0 == tt)) ? 1 : 0;
    const int c1 = (tb1 < tt || (mx != 0 && tb1 == tt)) ? 1 : 0;
    return oi + __mul24(m0 + c0 + c1, os);
}

__device__ __forceinline__ float cast_ray_fast(const uint32_t *bits, const uint8_t *blk, const RcTrackDev &t,
                                               float gx, float gy, float dx, float dy) {
    int ix = (int)floorf(gx), iy = (int)floorf(gy);
    bool alive = (unsigned)ix < (unsigned)t.w && (unsigned)iy < (unsigned)t.h;
    if (alive) alive = bit_at(bits, t.pitch, ix, iy) == 0;
    const bool started = alive;                                           // false: the sensor sits in a stop cell
    const float idx = select64(dx != 0.0f, 1.0f / dx, 3.0e38f);
    const float idy = select64(dy != 0.0f, 1.0f / dy, 3.0e38f);
    const int pxi = dx >= 0.0f ? 1 : 0, pyi = dy >= 0.0f ? 1 : 0;      // 1: the boundary ahead is the upper one
    int nx = pxi - 1, ny = pyi - 1;                                       // -1 for a negative direction
    asm("" : "+v"(nx));                                                   // see cast_ray_packed
    asm("" : "+v"(ny));
    const int shift = t.blk_shift, bs = 1 << shift, bmask = ~(bs - 1);
    const int cx = (pxi << shift) - nx, cy = (pyi << shift) - ny;
    const int blk_w = t.blk_w, pitch = t.pitch;
    const float tmax = t.tmax;
    float tt = 0.0f;
    int v = 0;
    if (alive) v = blk[__mul24(iy >> shift, blk_w) + (ix >> shift)];
    int guard = 0;
    while (alive) {
        // boundary that leaves the certified rectangle (v >= 1) or the current cell (v == 0)
        const int vm = nonzero_mask(v);
        const int r = v - 1;
        const int xe = bfi(vm, (ix & bmask) + cx + (r ^ nx), ix + pxi);
        const int ye = bfi(vm, (iy & bmask) + cy + (r ^ ny), iy + pyi);
        const float txe = ((float)xe - gx) * idx;
        const float tye = ((float)ye - gy) * idy;
        const int mx = sign_mask(txe - tye);                              // -1: leaves through the x side
        tt = fminf(txe, tye);
        // cell on the other axis after that crossing
        const float og = bfi(mx, gy, gx), od = bfi(mx, dy, dx);
        const float pe = og + tt * od;
        const float fl = floorf(pe);
        int on = (int)fl;
        if (fabsf((pe - fl) - 0.5f) > 0.5f - t.band) {                           // within the band of a boundary: exact count
            on = exact_other_cell(on, bfi(mx, iy, ix), bfi(mx, pyi, pxi), og, bfi(mx, idy, idx), tt, mx);
        }
        ix = bfi(mx, xe + nx, on);
        iy = bfi(mx, on, ye + ny);
        // no range test in the loop (see cast_ray_packed); measured 9 % faster on gbr, 2 % slower on barcelona
        if (++guard > 4096) break;                                        // bounds a logic error only
        v = blk[__mul24(iy >> shift, blk_w) + (ix >> shift)];
        if (v == 0) {                                                     // not certified: test the cell itself
            const uint32_t w = bits[__mul24(iy, pitch) + (ix >> 5)];
            alive = ((w >> (ix & 31)) & 1u) == 0;
        }
    }
    if (!started) return 0.0f;
    const bool ring = ix == 0 || iy == 0 || ix == t.w - 1 || iy == t.h - 1;
    return select64(!(tt < tmax) || ring || alive, RCS_MAX_RANGE, tt * t.res);
}

// Variant 3: variant 2 reading ONE table.  For 4x4 blocks a uint32 per block holds both the certified
// value (bits 16-23) and the occupancy of its 16 cells (bits 0-15), so an iteration is one LDS read and the
// cell test is branch-free: a certified block has no occupancy bits, hence `(word >> cell) & 1` is the hit
// flag for every block.  The row-major bitmap is not needed by the scan at all (LDS: 4 B per 16 cells).
__device__ __forceinline__ float cast_ray_packed(const uint32_t *pk, const RcTrackDev &t, float gx, float gy,
                                                 float dx, float dy) {
    int ix = (int)floorf(gx), iy = (int)floorf(gy);
    bool alive = (unsigned)ix < (unsigned)t.w && (unsigned)iy < (unsigned)t.h;
    const int row_bytes = t.packed_w * 4;
    const char *pkb = reinterpret_cast<const char *>(pk);
    uint32_t word = 0;
    if (alive) {
        word = *reinterpret_cast<const uint32_t *>(pkb + __mul24(iy >> 2, row_bytes) + (ix & ~3));
        alive = ((word >> (((iy & 3) << 2) | (ix & 3))) & 1u) == 0;
    }
    const bool started = alive;                                           // false: the sensor sits in a stop cell
    const float idx = select64(dx != 0.0f, 1.0f / dx, 3.0e38f);
    const float idy = select64(dy != 0.0f, 1.0f / dy, 3.0e38f);
    const int pxi = dx >= 0.0f ? 1 : 0, pyi = dy >= 0.0f ? 1 : 0;      // 1: the boundary ahead is the upper one
    int nx = pxi - 1, ny = pyi - 1;                                       // -1 for a negative direction
    // r * sx = (r ^ nx) - nx.  Hidden from LLVM's value tracking, otherwise it becomes compare + select, and
    // the e32 v_cndmask reading VCC that hipcc picks costs ~16 cycles (tools/ubench/valu_issue3.hip).
    asm("" : "+v"(nx));
    asm("" : "+v"(ny));
    const int cx = (pxi << 2) - nx, cy = (pyi << 2) - ny;
    const float tmax = t.tmax;
    float tt = 0.0f;
    int guard = 0;
    while (alive) {
        const int v = (int)(word >> 16);
        const int vm = nonzero_mask(v);
        const int r = v - 1;
        const int xe = bfi(vm, (ix & ~3) + cx + (r ^ nx), ix + pxi);
        const int ye = bfi(vm, (iy & ~3) + cy + (r ^ ny), iy + pyi);
        const float txe = ((float)xe - gx) * idx;
        const float tye = ((float)ye - gy) * idy;
        const int mx = sign_mask(txe - tye);                              // -1: leaves through the x side
        tt = fminf(txe, tye);
        const float og = bfi(mx, gy, gx), od = bfi(mx, dy, dx);
        const float pe = og + tt * od;
        const float fl = floorf(pe);
        int on = (int)fl;
        if (fabsf((pe - fl) - 0.5f) > 0.5f - t.band) {                           // within the band of a boundary: exact count
            on = exact_other_cell(on, bfi(mx, iy, ix), bfi(mx, pyi, pxi), og, bfi(mx, idy, idx), tt, mx);
        }
        ix = bfi(mx, xe + nx, on);
        iy = bfi(mx, on, ye + ny);
        // No range test here: boundary times only grow, so a ray that passes 15 m is still "no return" when
        // it finally stops (at a wall or at the sentinel ring) - decided once, after the loop.
        if (++guard > 4096) break;                                        // bounds a logic error only
        word = *reinterpret_cast<const uint32_t *>(pkb + __mul24(iy >> 2, row_bytes) + (ix & ~3));
        alive = ((word >> (((iy & 3) << 2) | (ix & 3))) & 1u) == 0;
    }
    if (!started) return 0.0f;
    const bool ring = ix == 0 || iy == 0 || ix == t.w - 1 || iy == t.h - 1;
    return select64(!(tt < tmax) || ring || alive, RCS_MAX_RANGE, tt * t.res);
}

// Variant 5: per-CELL certificates.  A byte per cell holds the chessboard distance D to the nearest stop
// cell (0 = stop cell); the square of half-width D - 1 around the current cell is free.  No block arithmetic,
// no blend between "rectangle" and "cell", no bit extraction (hit <=> D == 0): ~30 VALU per trip and ~12 %
// fewer trips than 4x4 blocks.  The table (1 B per cell: 0.25-1 MB) lives in global memory and is served by
// L1/L2; the kernel uses no LDS.
__device__ __forceinline__ float cast_ray_cells(const uint8_t *cd, const RcTrackDev &t, float gx, float gy,
                                                float dx, float dy) {
    int ix = (int)floorf(gx), iy = (int)floorf(gy);
    const int cpitch = t.cell_pitch;
    int D = 0;
    if ((unsigned)ix < (unsigned)t.w && (unsigned)iy < (unsigned)t.h) D = cd[__mul24(iy, cpitch) + ix];
    const bool started = D != 0;                                          // false: the sensor sits in a stop cell
    const float idx = select64(dx != 0.0f, 1.0f / dx, 3.0e38f);
    const float idy = select64(dy != 0.0f, 1.0f / dy, 3.0e38f);
    const int pxi = dx >= 0.0f ? 1 : 0, pyi = dy >= 0.0f ? 1 : 0;      // 1: the boundary ahead is the upper one
    int nx = pxi - 1, ny = pyi - 1;                                       // -1 for a negative direction
    asm("" : "+v"(nx));                                                   // see cast_ray_packed
    asm("" : "+v"(ny));
    const int cx = pxi - nx, cy = pyi - ny;                               // xe = ix + cx + ((D - 1) ^ nx)
    const float tmax = t.tmax;
    float tt = 0.0f;
    int guard = 0;
    while (D != 0) {
        const int r = D - 1;
        const int xe = ix + cx + (r ^ nx);                                // boundary that leaves the free square
        const int ye = iy + cy + (r ^ ny);
        const float txe = ((float)xe - gx) * idx;
        const float tye = ((float)ye - gy) * idy;
        const int mx = sign_mask(txe - tye);                              // -1: leaves through the x side
        tt = fminf(txe, tye);
        const float og = bfi(mx, gy, gx), od = bfi(mx, dy, dx);
        const float pe = og + tt * od;
        const float fl = floorf(pe);
        int on = (int)fl;
        if (fabsf((pe - fl) - 0.5f) > 0.5f - t.band) {                           // within the band of a boundary: exact count
            on = exact_other_cell(on, bfi(mx, iy, ix), bfi(mx, pyi, pxi), og, bfi(mx, idy, idx), tt, mx);
        }
        ix = bfi(mx, xe + nx, on);
        iy = bfi(mx, on, ye + ny);
        if (++guard > 4096) break;                                        // bounds a logic error only
        D = cd[__mul24(iy, cpitch) + ix];
    }
    if (!started) return 0.0f;
    const bool ring = ix == 0 || iy == 0 || ix == t.w - 1 || iy == t.h - 1;
    return select64(!(tt < tmax) || ring || D != 0, RCS_MAX_RANGE, tt * t.res);
}
// Variant 6: per-cell, per-QUADRANT free rectangles.  The ray only moves into its direction quadrant, so the
// certificate is a rectangle with the current cell at its corner (width | height << 8 in a uint16 per cell, one
// plane per quadrant, chosen once per ray): it reaches as far as the walls ahead allow, where variant 5's
// symmetric square is limited by the nearest wall in any direction.  Half the trips of variant 5
// (tools/analysis/skip_stats.py quadrant: 4.1 instead of 8.9 for the slowest ray of a wave on austria); same exit arithmetic.
// One ray of the per-ray kernel (variant 6): direction made safe, start entry read from its quadrant plane.
__device__ __forceinline__ float cast_ray_rects(const uint16_t *qr, const RcTrackDev &t, float gx, float gy,
                                                float dx, float dy, int ix, int iy) {
    // A non-finite direction (diverged car state) would make the cell arithmetic meaningless and could walk the table
    // index anywhere; v_max / v_min (IEEE maxNum / minNum: a NaN operand yields the other one) force it into
    // [-2, 2].  Any legal component has magnitude <= 1.0000002, so legal rays are untouched; an illegal one becomes
    // some finite ray that ends at the ring like every other.  The spec steps towards + iff d >= 0, which includes
    // -0.0, hence the + 0.0f (-0.0 + 0.0 = +0.0) in front of the sign extraction.
    dx = min_with(max_with(dx, -2.0f), 2.0f) + 0.0f;
    dy = min_with(max_with(dy, -2.0f), 2.0f) + 0.0f;
    const int nx = sign_mask(dx), ny = sign_mask(dy);
    float idx, idy;
    ray_reciprocals(dx, dy, idx, idy);
    const bool in_grid = (unsigned)ix < (unsigned)t.w && (unsigned)iy < (unsigned)t.h;
    return ray_traverse<true>(qr, t, trav_const(t), gx, gy, dx, dy, idx, idy, nx, ny, ix, iy, 0u, in_grid);
}

template <int A, int VARIANT>
__global__ __launch_bounds__(1024) void rc_raycast_kernel(RcParams p, int total_rays) {
    extern __shared__ uint32_t lds_words[];
    const RcTrackDev &t = p.trk;
    const int nwords = t.h * t.pitch;
    const uint8_t *lds_blk = reinterpret_cast<const uint8_t *>(lds_words + ((nwords + 15) & ~15));
    if (VARIANT >= 4) {
        // tables read from global memory (L2 / L1): no LDS, any map size
    } else if (VARIANT == 3) {
        stage_bitmap(lds_words, t.packed_blocks, t.packed_bytes >> 2);
    } else {
        if (VARIANT != 0) {
            uint4 *d4 = reinterpret_cast<uint4 *>(lds_words + ((nwords + 15) & ~15));
            const uint4 *s4 = reinterpret_cast<const uint4 *>(t.free_blocks);
            for (int i = threadIdx.x; i < (t.blk_bytes >> 4); i += blockDim.x) d4[i] = s4[i];
        }
        stage_bitmap(lds_words, t.ray_words, nwords);
    }
    for (unsigned base = blockIdx.x * blockDim.x; base < (unsigned)total_rays; base += gridDim.x * blockDim.x) {
        const unsigned g = base + threadIdx.x;          // unsigned indices: 32-bit offsets from scalar bases
        if (g >= (unsigned)total_rays) break;
        const unsigned car = g / RC_N_BEAMS;
        const unsigned beam = g - car * RC_N_BEAMS;
        const float ct = p.st.ct[car], st = p.st.st[car];
        const float lx = p.st.x[car] + RCS_LIDAR_X * ct;
        const float ly = p.st.y[car] + RCS_LIDAR_X * st;
        const float cb = t.beams[2 * beam], sb = t.beams[2 * beam + 1];
        const float dx = ct * cb - st * sb;
        const float dy = st * cb + ct * sb;
        const float gx = (lx - t.org_x) * t.inv_res;
        const float gy = (ly - t.org_y) * t.inv_res;
        float rng = VARIANT == 6   ? cast_ray_rects(t.quad_rect, t, gx, gy, dx, dy, (int)floorf(gx), (int)floorf(gy))
                    : VARIANT == 5 ? cast_ray_cells(t.cell_dist, t, gx, gy, dx, dy)
                    : VARIANT == 4 ? cast_ray_packed(t.packed_blocks, t, gx, gy, dx, dy)
                    : VARIANT == 3 ? cast_ray_packed(lds_words, t, gx, gy, dx, dy)
                    : VARIANT == 2 ? cast_ray_fast(lds_words, lds_blk, t, gx, gy, dx, dy)
                    : VARIANT == 1 ? cast_ray_skip(lds_words, lds_blk, t, gx, gy, dx, dy)
                                   : cast_ray_dda(lds_words, t, gx, gy, dx, dy);
        if (A > 1) {
            const unsigned env = car / A;
#pragma unroll
            for (unsigned o = 0; o < (unsigned)A; ++o) {
                const unsigned oc = env * A + o;
                if (oc != car) {
                    const float tc = ray_vs_car(lx, ly, dx, dy, p.st.x[oc], p.st.y[oc], p.st.ct[oc], p.st.st[oc]);
                    rng = tc < rng ? tc : rng;
                }
            }
        }
        if (p.lidar_transform == 1) rng = rng / RCS_MAX_RANGE - 0.5f;                 // dreamer/tools.py:274
        else if (p.lidar_transform == 2) rng = rng * (1.0f / RCS_MAX_RANGE);          // single_agent.py:92-99
        p.out.lidar[g] = rng;
    }
}

// The instrumented build of the same kernel (rc_debug_scan_stamps; one car per env, analysis only).
__global__ __launch_bounds__(256) void rc_raycast_car_stamps_kernel(RcParams p, int split, unsigned long long *stamps, int n_waves) {
    extern __shared__ uint32_t lds_words[];
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)lds_words;
    const uint32_t lds_row = __builtin_amdgcn_readfirstlane(lds_base + (threadIdx.x >> 6) * kCarLdsBytes);
    const unsigned wave = __builtin_amdgcn_readfirstlane(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
    const unsigned lane = threadIdx.x & 63u;
    const unsigned car = wave / (unsigned)split, part = wave - car * (unsigned)split;
    if (car >= (unsigned)p.n_cars) return;
    scan_car<1, true>(p, car, part, split, lane, lds_row, wave < (unsigned)n_waves ? stamps + (size_t)wave * RC_STAMP_SLOTS : nullptr);
}
// A co-tenant for the scan (tools/cohabit_sweep.py): `gridDim.x` workgroups that stay resident for `ticks` of the chip-wide 100 MHz
// clock - mode 0 asleep (they hold wave slots and nothing else), mode 1 copying the first half of `buf` to its second half over
// and over, 16 bytes per lane (what a collective's or a copy's workgroups do), mode 2 the same with non-temporal loads and
// stores.  Every wave leaves when the time is up.
__global__ __launch_bounds__(1024) void rclab_cohabit_kernel(unsigned long long ticks, uint4 *buf, unsigned long long n_vec, int mode,
                                                           unsigned long long *copied) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long half = n_vec / 2, stride = (unsigned long long)gridDim.x * blockDim.x;
    unsigned long long k = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x, done = 0;
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) {
        if (mode == 0 || half == 0) {
            __builtin_amdgcn_s_sleep(32);
        } else {
#pragma unroll 1
            for (int i = 0; i < 8; ++i) {
                if (k >= half) k -= (k / half) * half;
                if (mode == 2) __builtin_nontemporal_store(__builtin_nontemporal_load(reinterpret_cast<const v4u *>(buf) + k), reinterpret_cast<v4u *>(buf) + half + k);      // streamed past L2's retention
                else buf[half + k] = buf[k];
                k += stride;
            }
            done += 8;
        }
    }
    if (copied != nullptr && done != 0) atomicAdd(copied, done);          // 16-byte vectors this lane moved
}

template <typename K, typename... Args>
inline void launch(hipEvent_t a, hipEvent_t b, K kernel, dim3 grid, dim3 block, size_t lds, hipStream_t s, Args... args) {
    hipExtLaunchKernelGGL(kernel, grid, block, (uint32_t)lds, s, a, b, 0u, args...);
}

}  // namespace

#define DISPATCH_A(A, ...)                                     \
    switch (A) {                                               \
        case 1: { constexpr int kA = 1; __VA_ARGS__; } break;  \
        case 2: { constexpr int kA = 2; __VA_ARGS__; } break;  \
        case 3: { constexpr int kA = 3; __VA_ARGS__; } break;  \
        default: { constexpr int kA = 4; __VA_ARGS__; } break; \
    }

#ifndef RC_BUILD_ID
#define RC_BUILD_ID "unknown"
#endif

extern "C" {

// (the marker build.py looks for in the file's bytes: RC_BUILD_ID=<hash of the lab's sources and flags>)
const char *rclab_build_id(void) { return "RC_BUILD_ID=" RC_BUILD_ID; }

// what this library was built against (racecar_kernels.hip, rck_lab_abi_string: the loader compares the whole string)
#ifndef RC_HEADERS_ID
#define RC_HEADERS_ID "unhashed"
#endif
const char *rclab_abi(void) {
    static const std::string s = "RcParams " + std::to_string(sizeof(RcParams)) + " RcLaunchInfo " + std::to_string(sizeof(RcLaunchInfo)) +
                                 " headers " RC_HEADERS_ID;
    return s.c_str();
}

// Dynamic LDS of the variants that stage tables there, and the check that the stamps kernel - which addresses its dynamic
// LDS from 0 like the shipped scan - has no static LDS.
int rclab_set_lds_limits(size_t lds_bytes) {
    hipError_t e;
    const int b = (int)lds_bytes;
#define SET(k) e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, b); if (e != hipSuccess) return (int)e;
#define SET_V(v) SET((rc_raycast_kernel<1, v>)) SET((rc_raycast_kernel<2, v>)) SET((rc_raycast_kernel<3, v>)) SET((rc_raycast_kernel<4, v>))
    SET_V(0) SET_V(1) SET_V(2) SET_V(3) SET_V(4) SET_V(5) SET_V(6)
#undef SET_V
#undef SET
    hipFuncAttributes fa;
    e = hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(rc_raycast_car_stamps_kernel));
    if (e != hipSuccess) return (int)e;
    if (fa.sharedSizeBytes != 0) return (int)hipErrorInvalidValue;
    return (int)hipSuccess;
}

// The co-tenant kernel on `s`: workgroups x threads, resident for `microseconds` (at most 100 000); copied_dev (device, may be
// null) receives the number of 16-byte vectors moved.
int rclab_launch_cohabit(hipStream_t s, int workgroups, int threads, unsigned microseconds, void *buf, size_t bytes, int mode,
                         unsigned long long *copied_dev) {
    if (workgroups < 1 || workgroups > 65536 || threads < 64 || threads > 1024 || threads % 64 || microseconds > 100000u) return (int)hipErrorInvalidValue;
    if (mode != 0 && (buf == nullptr || bytes < 64 || ((uintptr_t)buf & 15u))) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(rclab_cohabit_kernel, dim3(workgroups), dim3(threads), 0, s, (unsigned long long)microseconds * 100ull,
                       reinterpret_cast<uint4 *>(buf), (unsigned long long)(bytes / 16), mode, copied_dev);
    return (int)hipGetLastError();
}

// One scan launch of a lab kernel on `s`: what rck_launch_raycast (racecar_kernels.hip) hands over when the handle's variant is
// not 7 or its stamps buffer is set.  ev_start / ev_stop: the launch-attached timer of the caller, or null.
int rclab_launch_raycast(const RcParams *pp, const RcLaunchInfo *lp, hipStream_t s, hipEvent_t ev_start, hipEvent_t ev_stop) {
    const RcParams &p = *pp;
    const RcLaunchInfo &li = *lp;
    const int total = p.n_cars * RC_N_BEAMS;
    if (li.raycast_variant == 7) {
        if (li.scan_stamps == nullptr || p.cars_per_env != 1) return (int)hipErrorInvalidValue;
        const int threads = li.car_threads, per = threads / 64;
        const long long waves = (long long)p.n_cars * li.car_split;
        launch(ev_start, ev_stop, rc_raycast_car_stamps_kernel, dim3((unsigned)((waves + per - 1) / per)), dim3(threads), (size_t)per * kCarLdsBytes, s, p,
               li.car_split, li.scan_stamps, li.scan_stamp_waves);
    } else if (li.raycast_variant == 6) {
        DISPATCH_A(p.cars_per_env, launch(ev_start, ev_stop, (rc_raycast_kernel<kA, 6>), dim3(li.ray_blocks), dim3(li.ray_threads), 0, s, p, total));
    } else if (li.raycast_variant == 5) {
        DISPATCH_A(p.cars_per_env, launch(ev_start, ev_stop, (rc_raycast_kernel<kA, 5>), dim3(li.ray_blocks), dim3(li.ray_threads), 0, s, p, total));
    } else if (li.raycast_variant == 4) {
        DISPATCH_A(p.cars_per_env, launch(ev_start, ev_stop, (rc_raycast_kernel<kA, 4>), dim3(li.ray_blocks), dim3(li.ray_threads), 0, s, p, total));
    } else if (li.raycast_variant == 3) {
        DISPATCH_A(p.cars_per_env, launch(ev_start, ev_stop, (rc_raycast_kernel<kA, 3>), dim3(li.ray_blocks), dim3(li.ray_threads), li.lds_bytes_packed, s, p, total));
    } else if (li.raycast_variant == 2) {
        DISPATCH_A(p.cars_per_env, launch(ev_start, ev_stop, (rc_raycast_kernel<kA, 2>), dim3(li.ray_blocks), dim3(li.ray_threads), li.lds_bytes_skip, s, p, total));
    } else if (li.raycast_variant == 1) {
        DISPATCH_A(p.cars_per_env, launch(ev_start, ev_stop, (rc_raycast_kernel<kA, 1>), dim3(li.ray_blocks), dim3(li.ray_threads), li.lds_bytes_skip, s, p, total));
    } else if (li.raycast_variant == 0) {
        DISPATCH_A(p.cars_per_env, launch(ev_start, ev_stop, (rc_raycast_kernel<kA, 0>), dim3(li.ray_blocks), dim3(li.ray_threads), li.lds_bytes, s, p, total));
    } else {
        return (int)hipErrorInvalidValue;
    }
    return (int)hipGetLastError();
}

}  // extern "C"
