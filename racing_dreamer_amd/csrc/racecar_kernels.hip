// HIP kernels of the batched racecar environment for MI355X (gfx950, wave64).
//
//   rc_dynamics_kernel  one lane per env: action remap, bicycle integrator (H2), wall / car-car
//                       collision (H5), progress + lap state machine, reward, done (H4, H15),
//                       in-kernel action repeat (H9), time limit (H10), auto-reset (H6)
//   rc_raycast_car_kernel  the LiDAR scan (H3), inter-car returns (H18): one wave per car, 17 rounds of 64 beams;
//                       exact grid traversal that crosses certified-free rectangles in one trip - the first from
//                       the per-cell first-trip table (by quadrant and slope bin), the later ones from the four
//                       quadrant planes; ranges staged in LDS and flushed at the end of the wave
//                       (device code: racecar_scan.h; the earlier forms of the scan, variants 0-6, and the instrumented
//                       build are NOT in this library: racecar_lab.hip, built on request, loaded on first use)
//   rc_build_quad_kernel, rc_build_first_kernel  build the quadrant planes and the first-trip table on the device
//                       at rc_load_track
//   rc_patch_car_kernel lidar_occupancy 64x64 ego patch (H11), one wave per car, drivable bitmap staged in LDS
//   rc_reset_kernel     masked reset from the centerline spawn table with Philox4x32-10 (H6)
//
// Numerics: fp32, one IEEE operation per written operator (-ffp-contract=off), same order as
// oracle/racecar_oracle.py, so results are bit-identical to the CPU oracle.
#include <dlfcn.h>
#include <mutex>
#include <string>

#include "racecar_scan.h"      // the traversal and scan_car (shared with the lab library, racecar_lab.hip)
#include "racecar_patch_exact.h"   // obs_type lidar_occupancy_reference

#define RC_PATCH 64

namespace {

__device__ __forceinline__ void cell_of(const RcTrackDev &t, float wx, float wy, int &ix, int &iy) {
    ix = (int)floorf((wx - t.org_x) * t.inv_res);
    iy = (int)floorf((wy - t.org_y) * t.inv_res);
}

__device__ __forceinline__ float progress_at(const RcTrackDev &t, float wx, float wy) {
    int ix, iy;
    cell_of(t, wx, wy, ix, iy);
    // branch-free (an off-grid car reads cell (0, 0) and discards it), so the load is issued next to the footprint's
    const bool inb = (unsigned)ix < (unsigned)t.w && (unsigned)iy < (unsigned)t.h;
    const float pr = t.progress[inb ? iy * t.w + ix : 0];
    return inb ? pr : -1.0f;
}

struct Car {
    float x, y, th, ct, st, v, dl, om, ac, pr, rew;
    int lap, cp;
    int wall, opp, wrong, done, trunc, fresh;
};

// What rc_patch_car_kernel needs of a car (RcStateDev::patch_pose): the start cell, the pixel step of the 64 x 64 patch in
// 16.16 cells (heading = + column, 3.125 cells per pixel) - or the mark of an all-zero patch: the first observation of an
// episode (dreamer/wrappers.py:413) and a car tens of thousands of cells off the grid (a diverged state sees nothing; it also
// keeps the fixed-point taps in range).  Computed by whoever writes the pose (one lane per car), so that the render, one wave
// per car, starts from ONE scalar 16-byte load and spends no vector instruction on wave-uniform values.
__device__ __forceinline__ int4 patch_pose_of(const RcTrackDev &t, float x, float y, float ct, float st, int fresh) {
    int icx, icy;
    cell_of(t, x, y, icx, icy);
    icy += 1;
    const bool sane = (unsigned)(icx + 16384) < 32768u && (unsigned)(icy + 16384) < 32768u;
    const int a = (int)__builtin_rintf(ct * RCS_PATCH_STEP_Q16), b = (int)__builtin_rintf(st * RCS_PATCH_STEP_Q16);
    return make_int4((fresh != 0 || !sane) ? RC_PATCH_SKIP : icx, icy, a, b);
}

// Footprint perimeter vs occupancy (H5): the 34 border points of the 12 x 7 body lattice (0.05 m pitch, rear axle at
// lattice node (2, 3)) in 16.16 fixed-point cell coordinates - oracle/racecar_oracle.py, _wall_hit.  With the lattice
// vectors e = rne(65536 k (cos, sin)) and f = (-e.y, e.x) a point is two integer multiply-adds of the rear-axle
// position, its cell two shifts, and "outside the grid counts as wall" is one unsigned min per axis: a negative or
// too large index clamps to the last row / column, which belongs to the sentinel ring and is always set.  About 10
// vector instructions per point, all 34 words requested back to back and waited for once (the fp32 rotation +
// floor + bounds select of the first version took 30 per point: 2/3 of the kernel, which runs one wave per SIMD and
// is therefore bound by its own instruction stream).
__device__ __forceinline__ int wall_hit(const RcTrackDev &t, const Car &c) {
    const float k = RCS_FOOT_STEP * t.inv_res;
    const float gx = (c.x - t.org_x) * t.inv_res, gy = (c.y - t.org_y) * t.inv_res;
    const bool bad = !(fabsf(gx) <= 8192.0f && fabsf(gy) <= 8192.0f);      // not a position: counts as contact
    const int ex = (int)__builtin_rintf((c.ct * k) * 65536.0f), ey = (int)__builtin_rintf((c.st * k) * 65536.0f);
    const int x0 = (int)__builtin_rintf(gx * 65536.0f), y0 = (int)__builtin_rintf(gy * 65536.0f);
    const uint32_t wm1 = (uint32_t)(t.w - 1), hm1 = (uint32_t)(t.h - 1);
    uint32_t hit = bad ? 1u : 0u;
    auto probe = [&](int li, int lj) {
        const int px = x0 + (li - 2) * ex - (lj - 3) * ey, py = y0 + (li - 2) * ey + (lj - 3) * ex;
        const uint32_t ix = min((uint32_t)(px >> 16), wm1), iy = min((uint32_t)(py >> 16), hm1);
        hit |= t.ray_words[iy * (uint32_t)t.pitch + (ix >> 5)] >> (ix & 31u);
    };
#pragma unroll
    for (int i = 0; i < 12; ++i) { probe(i, 0); probe(i, 6); }
#pragma unroll
    for (int j = 1; j < 6; ++j) { probe(0, j); probe(11, j); }
    return (int)(hit & 1u);
}

// Oriented-rectangle overlap by separating axes (car-car collision, H5/H18).
__device__ __forceinline__ int obb_overlap(const Car &a, const Car &b) {
    const float ax = a.x + RCS_BOX_CX * a.ct, ay = a.y + RCS_BOX_CX * a.st;
    const float bx = b.x + RCS_BOX_CX * b.ct, by = b.y + RCS_BOX_CX * b.st;
    const float dx = bx - ax, dy = by - ay;
    const float c = fabsf(a.ct * b.ct + a.st * b.st);
    const float s = fabsf(a.st * b.ct - a.ct * b.st);
    const float ra = RCS_BOX_HL + (RCS_BOX_HL * c + RCS_BOX_HW * s);
    const float rb = RCS_BOX_HW + (RCS_BOX_HL * s + RCS_BOX_HW * c);
    bool sep = fabsf(dx * a.ct + dy * a.st) > ra;
    sep |= fabsf(dy * a.ct - dx * a.st) > rb;
    sep |= fabsf(dx * b.ct + dy * b.st) > ra;
    sep |= fabsf(dy * b.ct - dx * b.st) > rb;
    return sep ? 0 : 1;
}

// Reset of one env in two halves.  `prepare_reset` is everything that does not depend on how the current step ends -
// the Philox draw keyed by (global env id, episode counter) and the gather of the spawn poses - so the dynamics kernel
// issues it next to the state loads, ahead of the integrator, instead of behind the step (that kernel runs one wave
// per SIMD: its duration is the length of its dependent chain, and a reset used to add three round trips to it in
// nearly every wave of a random-action rollout).  `apply_reset` installs the prepared poses when the env did finish.
struct Spawn { float x, y, th, ct, st, pr; int cp; };

// The spawn table (RcTrackDev::spawn): per row the pose, sin / cos of its heading (the spec's sincos32), the progress value of
// its cell, its checkpoint (+ the anchor bin of a multi-car start, see below), and the room a random start has there - computed
// once per track ON THE DEVICE with the very functions a reset would call, so the centre-line part of a reset is one 32-byte
// gather with no arithmetic behind it.
//   Row i is centre-line bin u(i) = the first USABLE bin among i, i + 1, ... (around the lap, RCS_SPAWN_SAFE_SEARCH of them; i if
// none): usable = the footprint test of H5 on the bin's own pose finds no wall (oracle: spawn_usable, spawn_rows).  On hand-drawn
// maps with boxes on the track the most central cell of a BFS distance bin can lie where a car does not fit; no start goes there.
//   Lateral room (oracle: spawn_width): d2 = squared cell distance from the point's cell to the nearest cell that is not drivable
// (outside the grid included) in the window of +- RCS_SPAWN_CLEAR_R cells, at most (R + 1)^2;
// w = clamp(isqrt(d2) * res - RCS_SPAWN_MARGIN, 0, RCS_SPAWN_W_MAX) - integers up to the last two operations.
//   Heading room (oracle: spawn_heading_room): RCS_HEADING_JITTER where w > 0 (the margin holds for every heading); where w = 0,
// HEADING_ROOM[k], k = the smallest over the 34 footprint points of isqrt(squared cell distance to the nearest non-drivable cell
// within +- RCS_SPAWN_FOOT_R), capped at 5.  The row's last word holds w if w > 0, else - (heading room): one float, no bit fields.
__device__ __forceinline__ void foot_cell(const RcTrackDev &t, float x, float y, float ct, float st, int li, int lj, int &ix, int &iy) {
    const float k = RCS_FOOT_STEP * t.inv_res;
    const float gx = (x - t.org_x) * t.inv_res, gy = (y - t.org_y) * t.inv_res;
    const int ex = (int)__builtin_rintf((ct * k) * 65536.0f), ey = (int)__builtin_rintf((st * k) * 65536.0f);
    const int x0 = (int)__builtin_rintf(gx * 65536.0f), y0 = (int)__builtin_rintf(gy * 65536.0f);
    ix = (x0 + (li - 2) * ex - (lj - 3) * ey) >> 16;
    iy = (y0 + (li - 2) * ey + (lj - 3) * ex) >> 16;
}

template <typename F>
__device__ __forceinline__ void for_each_foot_point(F &&f) {
    for (int i = 0; i < 12; ++i) { f(i, 0); f(i, 6); }
    for (int j = 1; j < 6; ++j) { f(0, j); f(11, j); }
}

__device__ __noinline__ bool bin_usable(const RcTrackDev &t, int i) {
    Car c;
    c.x = t.centerline[4 * i]; c.y = t.centerline[4 * i + 1];
    sincos32(t.centerline[4 * i + 2], c.st, c.ct);
    return wall_hit(t, c) == 0;
}

__global__ __launch_bounds__(256) void rc_build_spawn_kernel(RcTrackDev t, float4 *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = t.n_centerline;
    if (i >= n) return;
    int u = i;
    for (int s = 0; s < RCS_SPAWN_SAFE_SEARCH && s < n; ++s)
        if (bin_usable(t, (i + s) % n)) { u = (i + s) % n; break; }
    const float x = t.centerline[4 * u], y = t.centerline[4 * u + 1], th = t.centerline[4 * u + 2];
    float sn, cs;
    sincos32(th, sn, cs);
    float pr = progress_at(t, x, y);
    pr = pr < 0.0f ? 0.0f : pr;
    int cp = (int)(pr * (float)RCS_N_CHECKPOINTS);
    cp = cp < RCS_N_CHECKPOINTS - 1 ? cp : RCS_N_CHECKPOINTS - 1;
    auto clearance2 = [&](int ix, int iy, int R) {           // squared cell distance to the nearest non-drivable cell within +- R
        int d2 = (R + 1) * (R + 1);
        if (!((unsigned)ix < (unsigned)t.w && (unsigned)iy < (unsigned)t.h)) return 0;
        for (int dy = -R; dy <= R; ++dy)
            for (int dx = -R; dx <= R; ++dx) {
                const int jx = ix + dx, jy = iy + dy;
                const bool inside = (unsigned)jx < (unsigned)t.w && (unsigned)jy < (unsigned)t.h;
                const bool blocked = !inside || bit_at(t.drv_words, t.pitch, jx, jy) == 0;
                const int q = dx * dx + dy * dy;
                d2 = (blocked && q < d2) ? q : d2;
            }
        return d2;
    };
    auto isqrt = [](int d2) { int k = 0; while ((k + 1) * (k + 1) <= d2) ++k; return k; };
    int ix, iy;
    cell_of(t, x, y, ix, iy);
    const float w = clampf((float)isqrt(clearance2(ix, iy, RCS_SPAWN_CLEAR_R)) * t.res - RCS_SPAWN_MARGIN, 0.0f, RCS_SPAWN_W_MAX);
    float room = w;
    if (!(w > 0.0f)) {
        int kmin = RCS_SPAWN_FOOT_R;
        for_each_foot_point([&](int li, int lj) {
            int px, py;
            foot_cell(t, x, y, cs, sn, li, lj, px, py);
            const int k = isqrt(clearance2(px, py, RCS_SPAWN_FOOT_R));
            kmin = k < kmin ? k : kmin;
        });
        const float rooms[6] = RCS_HEADING_ROOM_INIT;
        room = -rooms[kmin];
    }
    // Where a multi-car start drawn at this bin really goes (oracle: spawn_safe): the first bin j among i, i + 1, ... (around the
    // lap, RCS_SPAWN_SAFE_SEARCH of them) at which the centre-line poses of RC_MAX_CARS cars RCS_BALL_GAP_BINS apart touch no wall
    // and do not overlap pairwise; i itself if there is none.  Where the progress grid's wavefronts fold (columbia_slam's last bins
    // run back along the bins before them) bins 1.2 m apart along the table are centimetres apart on the ground.
    int safe = i;
    for (int s = 0; s < RCS_SPAWN_SAFE_SEARCH && s < n; ++s) {
        const int j = (i + s) % n;
        Car c[4];
        int clash = 0;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            int idx = (j - a * RCS_BALL_GAP_BINS) % n;
            if (idx < 0) idx += n;
            c[a].x = t.centerline[4 * idx]; c[a].y = t.centerline[4 * idx + 1];
            sincos32(t.centerline[4 * idx + 2], c[a].st, c[a].ct);
            clash |= bin_usable(t, idx) ? 0 : 1;
        }
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = a + 1; b < 4; ++b) clash |= obb_overlap(c[a], c[b]);
        if (!clash) { safe = j; break; }
    }
    out[2 * i] = make_float4(x, y, th, cs);
    out[2 * i + 1] = make_float4(sn, pr, __int_as_float(cp | (safe << 8)), room);      // checkpoint < 256; the bin above it
}

__device__ __forceinline__ float unit_pm1(uint32_t w) { return ((float)(w >> 8) * 5.9604644775390625e-8f) * 2.0f - 1.0f; }   // [-1, 1), exact

__device__ __forceinline__ int obb_overlap_spawn(const Spawn &a, const Spawn &b) {
    Car ca, cb;
    ca.x = a.x; ca.y = a.y; ca.ct = a.ct; ca.st = a.st;
    cb.x = b.x; cb.y = b.y; cb.ct = b.ct; cb.st = b.st;
    return obb_overlap(ca, cb);
}

// The reset law (H6; oracle: _reset_envs).  `grid`: the cars on the centre line behind the start.  `random` / `random_ball`:
// word 0 of Philox(global env id, episode, 0, 0) picks a centre-line bin uniformly over the lap; car a stands at bin
// idx0 - a * BALL_GAP (row of the spawn table), moved sideways by u * w (w: the row's lateral room) and turned by v * h (h: its
// heading room, RCS_HEADING_JITTER wherever w > 0) off the track's direction; (u, v) = words 1, 2 of call 0 for car 0, words (0, 1) / (2, 3) of call 1 + (a - 1) / 2 for the others.
// If two proposed cars overlap, ALL cars of the env take the centre-line poses.
template <int A>
__device__ __forceinline__ void prepare_reset(const RcParams &p, int e, uint32_t ep, Spawn (&sp)[A]) {
    const RcTrackDev &t = p.trk;
    const uint32_t g = p.first_env + (uint32_t)e;
    const bool jitter = p.reset_mode != 0;
    rcd::u32x4 r[1 + A / 2];
    r[0] = rcd::philox4x32(g, ep, 0u, 0u, p.seed_lo, p.seed_hi);
    if (jitter) {
#pragma unroll
        for (int k = 1; k < 1 + A / 2; ++k) r[k] = rcd::philox4x32(g, ep, (uint32_t)k, 0u, p.seed_lo, p.seed_hi);
    }
    const int n = t.n_centerline;
    int idx0 = !jitter ? RCS_BALL_GAP_BINS * (A - 1) + RCS_GRID_LEAD_BINS : (int)__umulhi(r[0].x, (uint32_t)n);
    if (A > 1) idx0 = __float_as_int(t.spawn[2 * idx0 + 1].z) >> 8;      // never anchor several cars where the centre line folds (the grid too)
    Spawn centre[A];
#pragma unroll
    for (int a = 0; a < A; ++a) {
        int idx = (idx0 - a * RCS_BALL_GAP_BINS) % n;
        if (idx < 0) idx += n;
        const float4 s0 = t.spawn[2 * idx], s1 = t.spawn[2 * idx + 1];
        Spawn &c = centre[a];
        c.x = s0.x; c.y = s0.y; c.th = s0.z; c.ct = s0.w;
        c.st = s1.x; c.pr = s1.y; c.cp = __float_as_int(s1.z) & 0xff;
        sp[a] = c;
        if (jitter) {
            const rcd::u32x4 &q = r[a == 0 ? 0 : 1 + (a - 1) / 2];
            const uint32_t wu = a == 0 ? q.y : (((a - 1) & 1) ? q.z : q.x);
            const uint32_t wv = a == 0 ? q.z : (((a - 1) & 1) ? q.w : q.y);
            const float room = s1.w;                     // w > 0: lateral room, any heading; else - (heading room), no lateral room
            const float off = unit_pm1(wu) * fmaxf(room, 0.0f);
            Spawn &j = sp[a];
            j.x = c.x - off * c.st;
            j.y = c.y + off * c.ct;
            float th = c.th + unit_pm1(wv) * (room > 0.0f ? RCS_HEADING_JITTER : -room);
            th = th > RCS_PI ? th - RCS_TWO_PI : th;
            th = th < -RCS_PI ? th + RCS_TWO_PI : th;
            j.th = th;
            sincos32(th, j.st, j.ct);
            float pr = progress_at(t, j.x, j.y);
            pr = pr < 0.0f ? 0.0f : pr;
            j.pr = pr;
            const int cp = (int)(pr * (float)RCS_N_CHECKPOINTS);
            j.cp = cp < RCS_N_CHECKPOINTS - 1 ? cp : RCS_N_CHECKPOINTS - 1;
        }
    }
    if (A > 1 && jitter) {
        int clash = 0;
#pragma unroll
        for (int a = 0; a < A; ++a)
#pragma unroll
            for (int b = a + 1; b < A; ++b) clash |= obb_overlap_spawn(sp[a], sp[b]);
        if (clash) {
#pragma unroll
            for (int a = 0; a < A; ++a) sp[a] = centre[a];
        }
    }
}

template <int A>
__device__ __forceinline__ void apply_reset(const RcParams &p, int e, Car (&car)[A], const Spawn (&sp)[A], uint32_t ep,
                                            int &steps, int &agent_steps) {
    p.st.episode[e] = ep + 1u;
#pragma unroll
    for (int a = 0; a < A; ++a) {
        Car &c = car[a];
        c.x = sp[a].x; c.y = sp[a].y; c.th = sp[a].th; c.ct = sp[a].ct; c.st = sp[a].st;
        c.pr = sp[a].pr; c.cp = sp[a].cp;
        c.v = c.dl = c.om = c.ac = 0.0f;
        c.wall = c.opp = c.wrong = c.done = c.trunc = 0;
        c.lap = 1;
        c.fresh = 1;
        if (p.car_task[a] == 2) {                 // n_step_progress: the window starts at the spawn progress
            float *h = p.st.nstep_hist + (size_t)(e * A + a) * RC_NSTEP_MAX;
            for (int k = 0; k < RC_NSTEP_MAX; ++k) h[k] = sp[a].pr;
        }
    }
    steps = 0;
    agent_steps = 0;
}

template <int A>
__device__ __forceinline__ void load_cars(const RcParams &p, int e, Car (&car)[A]) {
#pragma unroll
    for (int a = 0; a < A; ++a) {
        const int i = e * A + a;
        Car &c = car[a];
        c.x = p.st.x[i]; c.y = p.st.y[i]; c.th = p.st.theta[i]; c.ct = p.st.ct[i]; c.st = p.st.st[i];
        c.v = p.st.v[i]; c.dl = p.st.delta[i]; c.om = p.st.omega[i]; c.ac = p.st.accel[i];
        c.pr = p.st.progress[i]; c.lap = p.st.lap[i]; c.cp = p.st.cp[i];
        c.wall = p.st.wall[i]; c.opp = p.st.opp[i]; c.wrong = p.st.wrong[i];
        c.done = p.st.done[i]; c.trunc = p.st.trunc[i]; c.fresh = 0;
        c.rew = 0.0f;
    }
}

template <int A>
__device__ __forceinline__ void store_state_and_obs(const RcParams &p, int e, const Car (&car)[A], int steps,
                                                    int agent_steps) {
#pragma unroll
    for (int a = 0; a < A; ++a) {
        const int i = e * A + a;
        const Car &c = car[a];
        p.st.x[i] = c.x; p.st.y[i] = c.y; p.st.theta[i] = c.th; p.st.ct[i] = c.ct; p.st.st[i] = c.st;
        p.st.v[i] = c.v; p.st.delta[i] = c.dl; p.st.omega[i] = c.om; p.st.accel[i] = c.ac;
        p.st.progress[i] = c.pr; p.st.lap[i] = c.lap; p.st.cp[i] = c.cp;
        p.st.wall[i] = c.wall; p.st.opp[i] = c.opp; p.st.wrong[i] = c.wrong;
        p.st.done[i] = c.done; p.st.trunc[i] = c.trunc; p.st.fresh[i] = c.fresh;
        p.st.scan_pose[i] = make_float4(c.x, c.y, c.ct, c.st);
        if (p.render_patch) p.st.patch_pose[i] = patch_pose_of(p.trk, c.x, c.y, c.ct, c.st, c.fresh);
        // observation of the current state (post auto-reset)
        float *pose = p.out.pose + 6 * i, *vel = p.out.velocity + 6 * i;
        pose[0] = c.x; pose[1] = c.y; pose[2] = 0.0f; pose[3] = 0.0f; pose[4] = 0.0f; pose[5] = c.th;
        vel[0] = c.v; vel[1] = 0.0f; vel[2] = 0.0f; vel[3] = 0.0f; vel[4] = 0.0f; vel[5] = c.om;
        p.out.speed[i] = fabsf(c.v);
        p.out.accel[i] = c.ac;
        p.out.steer[i] = c.dl;
        p.out.fresh[i] = c.fresh;
    }
    p.st.steps[e] = steps;
    p.st.agent_steps[e] = agent_steps;
}

// Results of the step that just ran (terminal values when the env finished).
template <int A>
__device__ __forceinline__ void store_step_results(const RcParams &p, int e, const Car (&car)[A], int steps) {
    const float time = (float)steps * RCS_DT;
#pragma unroll
    for (int a = 0; a < A; ++a) {
        const int i = e * A + a;
        const Car &c = car[a];
        p.out.reward[i] = c.rew;
        p.out.discount[i] = 1.0f - (float)c.done;
        p.out.progress_total[i] = (float)(c.lap - 1) + c.pr;
        p.out.time[i] = time;
        p.out.progress[i] = c.pr;
        p.out.lap[i] = c.lap;
        p.out.cp[i] = c.cp;
        p.out.done[i] = c.done;
        p.out.trunc[i] = c.trunc;
        p.out.wall[i] = c.wall;
        p.out.opp[i] = c.opp;
        p.out.wrong[i] = c.wrong;
    }
}

// ------------------------------------------------------------------------------------------------
template <int A>
// rand_on != 0: the actions are not read but drawn here, U(-1, 1)^2 from Philox keyed by (seed, step, global car id) -
// the same numbers rc_random_actions_kernel writes (synthetic random-action rollouts without a launch of their own);
// they are stored to `actions` as well, so the buffer shows what was applied.
__device__ __forceinline__ void dynamics_env(const RcParams &p, float *__restrict__ actions, const int repeat, const int rand_on,
                                             const uint32_t rand_lo, const uint32_t rand_hi, const uint32_t rand_step, const int e) {
    const RcTrackDev &t = p.trk;
    // the episode counter first: it is the oldest load in flight, so the reset's Philox draw can wait for it alone
    const uint32_t episode = p.auto_reset ? p.st.episode[e] : 0u;
    Car car[A];
    load_cars<A>(p, e, car);
    int steps = p.st.steps[e], agent_steps = p.st.agent_steps[e];
    Spawn spawn[A];
    if (p.auto_reset) prepare_reset<A>(p, e, episode, spawn);      // ahead of the step: see prepare_reset
    float motor[A], steer[A];
    bool any_done = false;
#pragma unroll
    for (int a = 0; a < A; ++a) {
        const int i = e * A + a;
        float a0, a1;
        if (rand_on) {
            const rcd::u32x4 r = rcd::philox4x32(p.first_env * (uint32_t)A + (uint32_t)i, rand_step, 1u, 0u, rand_lo, rand_hi);
            a0 = (float)(r.x >> 8) * 5.9604644775390625e-8f * 2.0f - 1.0f;      // as rc_random_actions_kernel
            a1 = (float)(r.y >> 8) * 5.9604644775390625e-8f * 2.0f - 1.0f;
            actions[2 * i] = a0;
            actions[2 * i + 1] = a1;
        } else {
            a0 = actions[2 * i];
            a1 = actions[2 * i + 1];
        }
        p.out.action[2 * i] = a0;
        p.out.action[2 * i + 1] = a1;
        float m = a0, s = a1;
        if (p.remap_actions) {   // ReduceActionSpace, dreamer/wrappers.py:128-130
            m = ((a0 + 1.0f) * 0.5f) * (p.act_hi0 - p.act_lo0) + p.act_lo0;
            s = ((a1 + 1.0f) * 0.5f) * (p.act_hi1 - p.act_lo1) + p.act_lo1;
        }
        motor[a] = clampf(m, -1.0f, 1.0f);
        steer[a] = clampf(s, -1.0f, 1.0f);
        any_done |= car[a].done != 0;
    }

    if (!any_done) {
        for (int sub = 0; sub < repeat; ++sub) {   // ActionRepeat, dreamer/wrappers.py:107-116
            // --- kinematic bicycle, explicit Euler (H2)
#pragma unroll
            for (int a = 0; a < A; ++a) {
                Car &c = car[a];
                const float m = motor[a];
                const float force = fabsf(m) * RCS_ACCEL_MAX;
                const float acc = (m >= 0.0f ? force : -force) - RCS_DRAG * c.v;
                c.v = clampf(c.v + acc * RCS_DT, 0.0f, RCS_MAX_VEL);
                const float dd = clampf(steer[a] * RCS_STEER_GAIN - c.dl, -RCS_STEER_STEP, RCS_STEER_STEP);
                c.dl = c.dl + dd;
                float sd, cd;
                sincos32(c.dl, sd, cd);
                c.om = (c.v / RCS_WHEELBASE) * (sd / cd);
                c.x = c.x + (c.v * c.ct) * RCS_DT;
                c.y = c.y + (c.v * c.st) * RCS_DT;
                float th = c.th + c.om * RCS_DT;
                th = th > RCS_PI ? th - RCS_TWO_PI : th;
                th = th < -RCS_PI ? th + RCS_TWO_PI : th;
                c.th = th;
                sincos32(th, c.st, c.ct);
                c.ac = acc;
            }
            steps += 1;
            // --- collisions (H5)
#pragma unroll
            for (int a = 0; a < A; ++a) {
                car[a].wall = wall_hit(t, car[a]);
                car[a].opp = 0;
            }
#pragma unroll
            for (int a = 0; a < A; ++a)
#pragma unroll
                for (int b = a + 1; b < A; ++b) {
                    const int o = obb_overlap(car[a], car[b]);
                    car[a].opp |= o;
                    car[b].opp |= o;
                }
            // --- progress, lap, reward, done (H4, H15)
            const float time = (float)steps * RCS_DT;
            bool stop = false;
#pragma unroll
            for (int a = 0; a < A; ++a) {
                Car &c = car[a];
                float p_new = progress_at(t, c.x, c.y);
                const float p_old = c.pr;
                const int lap_old = c.lap, cp_old = c.cp;
                p_new = p_new >= 0.0f ? p_new : p_old;
                int cp_new = (int)(p_new * (float)RCS_N_CHECKPOINTS);
                cp_new = cp_new < RCS_N_CHECKPOINTS - 1 ? cp_new : RCS_N_CHECKPOINTS - 1;
                int d = cp_new - cp_old;
                d = d < 0 ? d + RCS_N_CHECKPOINTS : d;
                const bool fwd = d > 0 && d <= RCS_N_CHECKPOINTS / 2;
                const bool bwd = d > RCS_N_CHECKPOINTS / 2;
                const int lap = lap_old + ((fwd && cp_new < cp_old) ? 1 : 0) - ((bwd && cp_new > cp_old) ? 1 : 0);
                c.wrong = fwd ? 0 : (bwd ? 1 : c.wrong);
                c.cp = (fwd || bwd) ? cp_new : cp_old;
                c.lap = lap;
                c.pr = p_new;
                const bool collided = (c.wall | c.opp) != 0;
                float r;
                bool done;
                const int task = p.car_task[a];
                if (task == 2) {
                    // n_step_progress, the secondary agents' task of baselines/scenarios/max_progress/columbia.yml:17-18:
                    // total progress gained over the last n_steps sub-steps, no collision term, never done
                    float *h = p.st.nstep_hist + (size_t)(e * A + a) * RC_NSTEP_MAX + (steps % p.n_steps);
                    const float total = (float)(lap - 1) + p_new;
                    r = (total - *h) * RCS_PROGRESS_REWARD;
                    *h = total;
                    done = false;
                } else if (task == 0) {
                    const float delta = (float)(lap - lap_old) + (p_new - p_old);
                    r = delta * RCS_PROGRESS_REWARD + (collided ? p.collision_reward : 0.0f);
                    done = (collided && p.terminate_on_collision) || lap > p.laps || time > p.time_limit;
                } else {   // baselines/racing/environment/tasks.py:6-18
                    r = c.wall ? -1.0f : -rcd::exp32(fabsf(steer[a]) - c.v);
                    done = false;
                }
                c.rew = c.rew + r;
                c.done = done ? 1 : 0;
                stop |= done;
            }
            if (stop) break;
        }
        agent_steps += 1;
        if (p.time_limit_steps > 0 && agent_steps >= p.time_limit_steps) {   // TimeLimit, wrappers.py:147-154
#pragma unroll
            for (int a = 0; a < A; ++a) { car[a].done = 1; car[a].trunc = 1; }
        }
    }
    store_step_results<A>(p, e, car, steps);
    if (p.auto_reset) {
        bool fin = false;
#pragma unroll
        for (int a = 0; a < A; ++a) fin |= car[a].done != 0;
        if (fin) apply_reset<A>(p, e, car, spawn, episode, steps, agent_steps);
    }
    store_state_and_obs<A>(p, e, car, steps, agent_steps);
}

template <int A>
__global__ __launch_bounds__(256) void rc_dynamics_kernel(RcParams p, float *__restrict__ actions, int repeat,
                                                          int rand_on, uint32_t rand_lo, uint32_t rand_hi, uint32_t rand_step) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= p.num_envs) return;
    dynamics_env<A>(p, actions, repeat, rand_on, rand_lo, rand_hi, rand_step, e);
}

// Where the car of rank r (position along the track) stands in RcStateDev::order - see rc_order_place_kernel.
__device__ __forceinline__ uint32_t order_slot_of_rank(uint32_t r, uint32_t n) {
    const uint32_t whole = (n / (8u * RC_ORDER_REGION)) * (8u * RC_ORDER_REGION);
    if (r >= whole) return r;
    const uint32_t g = r / RC_ORDER_REGION, o = r % RC_ORDER_REGION;
    return ((g >> 3) * RC_ORDER_REGION + o) * 8u + (g & 7u);
}

// ---- Several handles in ONE launch (rc_step_group: one handle per track, each filling its block of cars of one arena).  A
// wave belongs to one block: it finds it among <= RC_GROUP_MAX wave ranges (scalar compares) and works from THAT block's
// RcParams, read from a device table with scalar loads - the same loads, from another address, that the single-handle
// kernels do from their argument block.  The device code is the single-handle kernels' (dynamics_env, scan_car).
__device__ __forceinline__ int group_block(const RcGroup &g, const int wave) {
    int b = 0;
#pragma unroll
    for (int k = 1; k < RC_GROUP_MAX; ++k) b += (k < g.n && wave >= g.wave_start[k]) ? 1 : 0;
    return b;
}

template <int A>
__global__ __launch_bounds__(256) void rc_dynamics_group_kernel(const RcParams *__restrict__ params, RcGroup g, int repeat, int rand_on,
                                                                uint32_t rand_lo, uint32_t rand_hi, uint32_t rand_step) {
    const int wave = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)));
    if (wave >= g.wave_start[g.n]) return;
    const int b = group_block(g, wave);
    const RcParams p = params[b];           // a COPY (top-level pointer argument + uniform index: scalar loads; a reference would be
                                            // re-read after every store, the table not being known to stay as it is)
    const int e = (wave - g.wave_start[b]) * 64 + (int)(threadIdx.x & 63u);
    if (e >= p.num_envs) return;
    dynamics_env<A>(p, g.actions[b], repeat, rand_on, rand_lo, rand_hi, rand_step, e);
}

template <int A>
__global__ __launch_bounds__(256) void rc_reset_kernel(RcParams p, const uint8_t *__restrict__ mask) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= p.num_envs) return;
    if (mask != nullptr && mask[e] == 0) {
        // an env that keeps running: its next observation is no longer the first of an episode
        return;
    }
    Car car[A];
    int steps, agent_steps;
#pragma unroll
    for (int a = 0; a < A; ++a) car[a].rew = 0.0f;
    Spawn spawn[A];
    const uint32_t episode = p.st.episode[e];
    prepare_reset<A>(p, e, episode, spawn);
    apply_reset<A>(p, e, car, spawn, episode, steps, agent_steps);
#pragma unroll
    for (int a = 0; a < A; ++a) {
        const int i = e * A + a;
        p.out.action[2 * i] = 0.0f;       // Collect.reset, dreamer/wrappers.py:232
        p.out.action[2 * i + 1] = 0.0f;
    }
    store_step_results<A>(p, e, car, steps);
    store_state_and_obs<A>(p, e, car, steps, agent_steps);
}

// Device self-test of that property (rc_selftest_reciprocal): every fp32 with biased exponent in [exp_lo, exp_hi],
// both signs, fast path against IEEE division.
__global__ __launch_bounds__(256) void rc_selftest_rcp_kernel(uint32_t exp_lo, uint32_t exp_hi,
                                                              unsigned long long *mismatches) {
    const uint64_t total = (uint64_t)(exp_hi - exp_lo + 1) << 23;
    unsigned long long bad = 0;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t bits = ((exp_lo + (uint32_t)(i >> 23)) << 23) | (uint32_t)(i & 0x7fffffu);
        const float d = __uint_as_float(bits), one = 1.0f;
        const float rp = __builtin_amdgcn_rcpf(d), rn = __builtin_amdgcn_rcpf(-d);          // the fast path of ray_reciprocals
        const float fx = __builtin_fmaf(__builtin_fmaf(-d, rp, 1.0f), rp, rp), fy = __builtin_fmaf(__builtin_fmaf(d, rn, 1.0f), rn, rn);
        bad += __float_as_uint(fx) != __float_as_uint(one / d);
        bad += __float_as_uint(fy) != __float_as_uint(one / -d);
    }
    if (bad) atomicAdd(mismatches, bad);
}

// ---- First-trip table (RcTrackDev::first_rect) --------------------------------------------------------------
// All 1080 rays of a car start in the same cell, so the FIRST rectangle of every ray can come from a much richer
// table than the four quadrant planes without any cache cost: a car reads one 512-byte line per step.  Per cell
// the line holds RC_FIRST_PLANES = 4 quadrants x RC_FIRST_BINS entries, the bin being the ray's slope |dy / dx|
// in eight steps per octave over 2^-4 .. 2^4 (the outer bins open-ended): exactly what the scan gets from the float
// bits of |dy| * |1 / dx| (exponent and three mantissa bits) in two instructions.  An entry is a rectangle anchored at the cell like the plane entries, but
// it only has to be free INSIDE THE SECTOR that rays of its bin can touch (start point anywhere in the cell, slope
// anywhere in the bin, both widened by a margin far above the traversal's rounding) - its far corners may lie
// inside walls.  The exit arithmetic is unchanged: a ray of that bin visits only sector cells before it leaves
// the rectangle, and those are free.  A ray heading down a diagonal straight thus crosses it in one trip where
// fully free rectangles need one per stair of the wall.  tools/analysis/skip_stats.py firsttrip-slope / sector: 3.1 trips for the slowest
// ray of a wave on austria against 4.1 with the quadrant planes alone; specialising the later trips as well
// would need the big table in L2 and gain little more (tools/analysis/skip_stats.py firsttrip-angle).
//
// Bin parameters (set by rck_build_first_table): slope range in the bin's own frame (bins >= RC_FIRST_BINS / 2 are
// y-dominant and handled with the axes swapped, slope = |dx / dy|) and 1/cos, 1/sin of two sample directions.
struct RcFirstBin { float s1, s2, ka0, kb0, ka1, kb1; };
__constant__ RcFirstBin c_first_bins[RC_FIRST_BINS];

// One thread per (cell, quadrant, bin).  Column c of the rectangle (offset along the bin's dominant axis) is touched
// by rays of the bin in rows floor(s1 (c - 1) - 0.01) .. floor(1 + s2 (c + 1) + 0.01): the ray is inside column c
// for travelled distances in (c - 1, c + 1) along the dominant axis and starts anywhere in [0, 1]^2.  The first
// stop cell in that range caps the height of every rectangle that includes the column; among the rectangles
// (c + 1) x cap(c) the one at whose exit the most sample rays stop is kept, then the one with the largest summed exit
// distance for two sample directions.
__global__ __launch_bounds__(256) void rc_build_first_kernel(RcTrackDev t, uint16_t *__restrict__ out) {
    const unsigned gid = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned total = (unsigned)t.h * (unsigned)t.w * RC_FIRST_PLANES;
    if (gid >= total) return;
    const int bin = (int)(gid % RC_FIRST_BINS), q = (int)((gid / RC_FIRST_BINS) & 3u);
    const unsigned cell = gid / RC_FIRST_PLANES;
    const int ix = (int)(cell % (unsigned)t.w), iy = (int)(cell / (unsigned)t.w);
    uint16_t &e = out[((size_t)iy * t.cell_pitch + ix) * RC_FIRST_PLANES + q * RC_FIRST_BINS + bin];
    if (ix == 0 || iy == 0 || ix == t.w - 1 || iy == t.h - 1) { e = 0x0100; return; }     // sentinel ring: "no return"
    if (bit_at(t.ray_words, t.pitch, ix, iy)) { e = 0; return; }                            // wall
    const int sx = (q & 1) ? -1 : 1, sy = (q & 2) ? -1 : 1;       // plane group q = (dy < 0) * 2 + (dx < 0)
    const bool swap = bin >= RC_FIRST_BINS / 2;
    const RcFirstBin b = c_first_bins[bin];
    const int cap = 255;
    int hmax = cap, bw = 1, bh = 1;
    float best = -1.0f;
    for (int c = 0; c < cap; ++c) {
        const int lo = max(0, (int)floorf(b.s1 * (float)max(0, c - 1) - 0.01f));
        const int hi = min(hmax - 1, (int)floorf(1.0f + b.s2 * (float)(c + 1) + 0.01f));
        for (int r = lo; r <= hi; ++r) {
            const int x = swap ? ix + sx * r : ix + sx * c, y = swap ? iy + sy * c : iy + sy * r;
            const bool stop = (unsigned)x >= (unsigned)t.w || (unsigned)y >= (unsigned)t.h || bit_at(t.ray_words, t.pitch, x, y);
            if (stop) { hmax = r; break; }
        }
        if (hmax <= 0) break;
        const int pw = swap ? hmax : c + 1, ph = swap ? c + 1 : hmax;          // extents along x and y
        float sc = fminf((float)pw * b.ka0, (float)ph * b.kb0) + fminf((float)pw * b.ka1, (float)ph * b.kb1);
        // ... after the number of sample rays (from the cell centre, five slopes across the bin) that STOP where they leave
        // the rectangle, i.e. whose exit cell is a stop cell: such a ray is finished after one trip, and a wave's round is as
        // long as its slowest ray (A/B on one box: 0.1845 -> 0.181 ms; the longest rectangle is not the one with the fewest
        // second trips - thinner sectors from sub-cell start positions made longer rectangles AND more trips)
        int stops = 0;
        for (int k = 0; k < 5; ++k) {
            const float m = b.s1 + (b.s2 - b.s1) * (0.1f + 0.2f * (float)k);         // own-frame slope (rows per column)
            const float yfar = 0.5f + m * ((float)c + 0.5f);
            int ec, er;                                                                 // exit cell, own-frame (column, row)
            if (yfar < (float)hmax) { ec = c + 1; er = (int)floorf(yfar); }
            else { ec = (int)floorf(0.5f + ((float)hmax - 0.5f) / fmaxf(m, 1e-6f)); er = hmax; }
            const int x = swap ? ix + sx * er : ix + sx * ec, y = swap ? iy + sy * ec : iy + sy * er;
            stops += ((unsigned)x >= (unsigned)t.w || (unsigned)y >= (unsigned)t.h || bit_at(t.ray_words, t.pitch, x, y)) ? 1 : 0;
        }
        sc += 1.0e4f * (float)stops;
        if (sc > best) { best = sc; bw = pw; bh = ph; }
    }
    e = (uint16_t)(bw | (bh << 8));
}

// ---- Quadrant planes (RcTrackDev::quad_rect), built on the device --------------------------------------------------
// Free run length from every cell towards -x and towards +x (capped at 255; 0 on a stop cell): one thread per row.
__global__ __launch_bounds__(256) void rc_build_runs_kernel(RcTrackDev t, uint8_t *__restrict__ run_neg, uint8_t *__restrict__ run_pos) {
    const int iy = blockIdx.x * blockDim.x + threadIdx.x;
    if (iy >= t.h) return;
    int r = 0;
    for (int ix = 0; ix < t.w; ++ix) {                          // towards -x: cells ix, ix - 1, ... are free
        r = bit_at(t.ray_words, t.pitch, ix, iy) ? 0 : min(r + 1, 255);
        run_neg[(size_t)iy * t.w + ix] = (uint8_t)r;
    }
    r = 0;
    for (int ix = t.w - 1; ix >= 0; --ix) {
        r = bit_at(t.ray_words, t.pitch, ix, iy) ? 0 : min(r + 1, 255);
        run_pos[(size_t)iy * t.w + ix] = (uint8_t)r;
    }
}

// One thread per (cell, quadrant): among the free rectangles anchored at the cell (width = min over its rows of the free
// run towards sx) the one with the largest geometric mean of the exit distances of rays at 11.25, 33.75, 56.25 and
// 78.75 degrees inside the quadrant.
__global__ __launch_bounds__(256) void rc_build_quad_kernel(RcTrackDev t, const uint8_t *__restrict__ run_neg,
                                                            const uint8_t *__restrict__ run_pos, uint16_t *__restrict__ out) {
    const unsigned gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (unsigned)t.h * (unsigned)t.w * 4u) return;
    const int q = (int)(gid & 3u);
    const unsigned cell = gid >> 2;
    const int ix = (int)(cell % (unsigned)t.w), iy = (int)(cell / (unsigned)t.w);
    const int sx = (q & 1) ? -1 : 1, sy = (q & 2) ? -1 : 1;       // plane q = (dy < 0) * 2 + (dx < 0)
    // mirrored storage: a ray heading -x reads its plane with the columns reversed (likewise -y and the rows)
    const int rx = sx > 0 ? ix : t.w - 1 - ix, ry = sy > 0 ? iy : t.h - 1 - iy;
    uint16_t &e = out[(size_t)q * (t.quad_plane_bytes / 2) + (size_t)ry * t.cell_pitch + rx];
    if (ix == 0 || iy == 0 || ix == t.w - 1 || iy == t.h - 1) { e = 0x0100; return; }       // sentinel ring: "no return"
    if (bit_at(t.ray_words, t.pitch, ix, iy)) { e = 0; return; }                              // wall
    const uint8_t *run = sx > 0 ? run_pos : run_neg;
    // 1 / cos and 1 / sin of the four sample directions
    const float ka[4] = {1.0195911f, 1.2026898f, 1.7999525f, 5.1258309f};
    const float kb[4] = {5.1258309f, 1.7999525f, 1.2026898f, 1.0195911f};
    const float log_ka_sum = __logf(ka[0]) + __logf(ka[1]) + __logf(ka[2]) + __logf(ka[3]);
    int cur = 255, bw = 1, bh = 1;
    float best = -1.0e30f;
    for (int n = 1; n <= 255; ++n) {
        const int y = iy + (n - 1) * sy;
        if (y < 0 || y >= t.h) break;
        cur = min(cur, (int)run[(size_t)y * t.w + ix]);
        // the width only shrinks from here on and the score is at most sum log(width * ka)
        if (cur == 0 || 4.0f * __logf((float)cur) + log_ka_sum <= best) break;
        float sc = 0.0f;
#pragma unroll
        for (int k = 0; k < 4; ++k) sc += __logf(fminf((float)cur * ka[k], (float)n * kb[k]));
        if (sc > best) { best = sc; bw = cur; bh = n; }
    }
    e = (uint16_t)(bw | (bh << 8));
}

// Variant 7: the traversal of variant 6 with ONE WAVE PER CAR.  The car index is wave-uniform, so the car state
// comes through scalar loads, everything that depends only on the car (sensor position, start cell, its range
// test) is computed once instead of once per 64 beams, and the wave walks its car's 1080 beams in 17 rounds of
// 64 (the last with 56 active lanes) - consecutive rounds start from the same cell, so their table lines are
// still in L1.  No persistent loop: 65 536 independent waves are balanced by the hardware dispatcher, where
// equal shares of chunks per resident workgroup left the slowest workgroup's tail exposed.
// `split` waves share a car (wave part k takes rounds k, k + split, ...): small batches still fill the chip.
template <int A, bool OVERLAP, bool GUARD>
__global__ __launch_bounds__(256) void rc_raycast_car_kernel(RcParams p, int split) {
    // One car (or 1 / split of one) per wave and nothing more: several cars in sequence per wave were measured slower
    // (2 per wave + 6 %, 8 per wave + 20 %) - the hardware dispatcher balances 65 536 short waves better than any
    // static share, and a finished wave's flush is not waited for by anybody.
    extern __shared__ uint32_t lds_words[];                              // 17 x 64 floats per wave of the workgroup
    // LDS address of this wave's row
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)lds_words;
    const uint32_t lds_row = __builtin_amdgcn_readfirstlane(lds_base + (threadIdx.x >> 6) * kCarLdsBytes);
    const unsigned wave = __builtin_amdgcn_readfirstlane(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
    const unsigned lane = threadIdx.x & 63u;
    const unsigned slot = wave / (unsigned)split, part = wave - slot * (unsigned)split;
    if (slot >= (unsigned)p.n_cars) return;
    // (the waves take the cars in track order - see RcStateDev::order -: 5 % of the scan at 65 536 cars, EXPERIMENTS I.11)
    const unsigned car = p.st.order != nullptr ? (unsigned)p.st.order[slot] : slot;
    scan_car<A, false, OVERLAP, GUARD>(p, car, part, split, lane, lds_row);
}

template <int A, bool OVERLAP>
__global__ __launch_bounds__(256) void rc_raycast_group_kernel(const RcParams *__restrict__ params, RcGroup g, int split) {
    extern __shared__ uint32_t lds_words[];
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)lds_words;
    const uint32_t lds_row = __builtin_amdgcn_readfirstlane(lds_base + (threadIdx.x >> 6) * kCarLdsBytes);
    // Workgroups go to the 8 XCDs in turn, each with an L2 of its own: workgroup i takes wave (i mod 8) x ceil(W / 8) + i / 8, so
    // that an XCD works through ONE stretch of the car order - one track's tables in its L2, two at a block boundary - instead
    // of every eighth car of all tracks.
    const int launched = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)));
    const int per_xcd = (g.wave_start[g.n] + 7) >> 3;
    const int wave = (launched & 7) * per_xcd + (launched >> 3);
    if ((launched >> 3) >= per_xcd || wave >= g.wave_start[g.n]) return;
    const int b = group_block(g, wave);
    const RcParams p = params[b];           // (a copy: see rc_dynamics_group_kernel)
    const unsigned local = (unsigned)(wave - g.wave_start[b]);
    const unsigned slot = local / (unsigned)split, part = local - slot * (unsigned)split;
    if (slot >= (unsigned)p.n_cars) return;
    // (this launch hands every XCD a STRETCH of the slots, not every eighth: it takes the cars by rank)
    const unsigned car = p.st.order != nullptr ? (unsigned)p.st.order[order_slot_of_rank(slot, (uint32_t)p.n_cars)] : slot;
    scan_car<A, false, OVERLAP, false>(p, car, part, split, threadIdx.x & 63u, lds_row);
}

// lidar_occupancy (H11, dreamer/wrappers.py:390-408): ego-aligned 64x64 patch of the drivable area,
// heading = +col, 3.125 cells per pixel, 1 = drivable.  Direct inverse map of the reference's
// crop -> rotate -> centre-crop -> resize chain: centred on the north-west corner of the car's cell,
// one nearest-cell tap per pixel, taps outside the reference's 220-cell crop window read 0.
//
// Tap positions are the spec's 16.16 fixed-point walk (oracle/racecar_oracle.py, render_patch): with
// (a, b) = rne(3.125 * 65536 * (cos, sin)) pixel (row r, col c) taps cell offset (X >> 16, Y >> 16),
// X = X00 + c a + r b, Y = Y00 + c b - r a.  One lane renders 16 adjacent pixels of a row (a quarter row) and stores
// them as one 16-byte vector; per pixel that is two integer adds, and because the start cell is folded into X and Y the
// byte address of the tap in the LDS bitmap is one shift and one 16 x 16-bit multiply-add that reads Y's high half
// directly: add, add, shift, mad, bit index, LDS byte read, shift into the pixel's byte lane = 6 vector instructions per
// pixel (20 in the fp32 form this replaces).  The window / grid test is hoisted out of the pixel loop: the valid taps of a
// run lie in the rectangle window-intersected-with-grid, which is convex, so a run whose two end taps are valid
// is valid throughout - a wave whose 64 runs all pass that test takes the test-free loop; the others (runs that
// cross a rotated corner of the window) take the loop that tests every tap (11 per pixel).
__device__ __forceinline__ uint32_t bfe_u32(uint32_t v, uint32_t offset, uint32_t width) {       // offset, width mod 32
    uint32_t r;
    asm("v_bfe_u32 %0, %1, %2, %3" : "=v"(r) : "v"(v), "v"(offset), "v"(width));
    return r;
}
__device__ __forceinline__ int mad_i24(int a, int b, int c) {
    int r;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(b), "v"(c));
    return r;
}

// (Round 4: every operand that used to be wave-uniform - the pixel step, the grid limits, the window - sits in a VECTOR
// register: on gfx950 a vector instruction with a scalar-register operand issues at half rate (4.3 cycles against 2.35,
// tools/ubench/valu_issue4.hip), and the render is bound by exactly that - profiles/r04_a_pmc_patch_kernel_round3_build.txt:
// 47.1 M wave-level vector instructions per 65 536 cars, 97.6 % lane use, 58 % of the wave-cycles spent waiting to issue;
// 4.85 LDS cycles per byte gather of which 2.8 are bank conflicts, but the LDS pipe is only 40 % busy.)
__device__ __forceinline__ uint32_t min_u32(uint32_t a, uint32_t b) {
    uint32_t r;
    asm("v_min_u32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ uint32_t mad_hi16_s(uint32_t y, uint32_t pitch, uint32_t c) {         // (y >> 16) * pitch + c, pitch scalar
    uint32_t r;
    asm("v_mad_u32_u16 %0, %1, %2, %3 op_sel:[1,0,0,0]" : "=v"(r) : "v"(y), "s"(pitch), "v"(c));
    return r;
}
// byte BYTE of acc = low byte of (v >> sh), the other bytes of acc kept: extraction and packing of a pixel in ONE instruction
// (sub-dword addressing writes the result into the byte lane; the bits above the wanted one are cleared once per word)
template <int BYTE>
__device__ __forceinline__ uint32_t shr_into_byte(uint32_t acc, uint32_t sh, uint32_t v) {
    if (BYTE == 0) {
        asm("v_lshrrev_b32 %0, %1, %2" : "=v"(acc) : "v"(sh), "v"(v));
    } else if (BYTE == 1) {
        asm("v_lshrrev_b32_sdwa %0, %1, %2 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(acc) : "v"(sh), "v"(v));
    } else if (BYTE == 2) {
        asm("v_lshrrev_b32_sdwa %0, %1, %2 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(acc) : "v"(sh), "v"(v));
    } else {
        asm("v_lshrrev_b32_sdwa %0, %1, %2 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(acc) : "v"(sh), "v"(v));
    }
    return acc;
}
template <int I>
__device__ __forceinline__ void patch_put(uint32_t (&words)[4], uint32_t sh, uint32_t v) {
    words[I >> 2] = shr_into_byte<I & 3>(words[I >> 2], sh, v);
}
struct PatchConst {          // per car, all in vector registers (see above)
    int a, b;                // pixel step
    uint32_t xmax, ymax;     // last cell of the grid, 16.16 + 0xffff
    uint32_t wx, wy, span;   // the reference's crop window: origin and extent - 1 ulp
    uint32_t zero_addr;      // LDS address of the all-zero word behind the bitmap
    uint32_t three, one;     // the widths of the two bit-field extracts
    uint32_t sixteen;
    uint32_t ones;           // 0x01010101: bit 0 of every pixel byte
};

// One 16-pixel run: taps (X, Y) += (a, b).  CLAMP: coordinates are forced into the grid first - a tap left of / below
// the grid has a negative coordinate, i.e. a huge unsigned one, and clamps like one beyond the far edge to the last
// column / row, which is never drivable (rc_load_track clears the bitmap's outermost ring; so does the oracle).
// Per pixel: add, add, shift, multiply-add, bit index, LDS byte read, shift-into-its-byte-lane (sub-dword addressing: the
// extract and the pack in one instruction), + one mask per word = 6.25 vector instructions.
template <bool CLAMP, bool TESTED, int I>
struct PatchStep {
    static __device__ __forceinline__ void run(const __attribute__((address_space(3))) uint8_t *lds, int X, int Y, const PatchConst &k,
                                               uint32_t pitch_b, uint32_t (&words)[4]) {
        const uint32_t xc = CLAMP ? min_u32((uint32_t)X, k.xmax) : (uint32_t)X;
        const uint32_t yc = CLAMP ? min_u32((uint32_t)Y, k.ymax) : (uint32_t)Y;
        uint32_t addr = mad_hi16_s(yc, pitch_b, xc >> 19);                        // iy * pitch + ix / 8
        if (TESTED) {
            // inside the window <=> both window-relative coordinates (unsigned: below the window = huge) are <= 220 cells - 1 ulp
            const uint32_t far = max((uint32_t)X - k.wx, (uint32_t)Y - k.wy);
            addr = far <= k.span ? addr : k.zero_addr;
        }
        const uint32_t byte = lds[addr];
        patch_put<I>(words, bfe_u32(xc, k.sixteen, k.three), byte);              // byte I & 3 of its word = byte >> (ix % 8)
        PatchStep<CLAMP, TESTED, I + 1>::run(lds, X + k.a, Y + k.b, k, pitch_b, words);
    }
};
template <bool CLAMP, bool TESTED>
struct PatchStep<CLAMP, TESTED, 16> {
    static __device__ __forceinline__ void run(const __attribute__((address_space(3))) uint8_t *, int, int, const PatchConst &k, uint32_t,
                                               uint32_t (&words)[4]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) words[j] &= k.ones;                            // keep bit 0 of every byte
    }
};
template <bool CLAMP>
__device__ __forceinline__ void patch_run(const __attribute__((address_space(3))) uint8_t *lds, int X, int Y, const PatchConst &k,
                                          uint32_t pitch_b, uint32_t (&words)[4]) {
    PatchStep<CLAMP, false, 0>::run(lds, X, Y, k, pitch_b, words);
}

// The same with every tap tested against the crop window (the corner runs of a car whose window corner cuts them).
template <bool CLAMP>
__device__ __forceinline__ void patch_run_tested(const __attribute__((address_space(3))) uint8_t *lds, int X, int Y, const PatchConst &k,
                                                 uint32_t pitch_b, uint32_t (&words)[4]) {
    PatchStep<CLAMP, true, 0>::run(lds, X, Y, k, pitch_b, words);
}

// ---- lidar_occupancy, ONE WAVE PER CAR -----------------------------------------------------------------------------
// (Round 3 gave a car to one wave - EXPERIMENTS.md I.1; round 4 took the wave-uniform work out of the vector unit.)
// * the car arrives as ONE scalar 16-byte load of integers (RcStateDev::patch_pose, written by the kernel that moved the car):
//   start cell, pixel step, or the mark of an all-zero patch; the window tests, the tap of pixel (0, 0) and the store base are
//   scalar arithmetic; the vector unit only walks pixels;
// * a lane renders four runs of 16 pixels, one per 16-row group g, and a store instruction writes group g of every lane:
//   lanes 4 i .. 4 i + 3 hold the four pieces of row 16 g + i, so every store covers whole 64-byte lines (16 rows = 1 KB
//   of consecutive bytes) - no line is split between waves or between instructions (HBM writes = 1.00 x the pixels);
// * only the 64 runs in the corner blocks (rows 0-15 and 48-63, columns 0-15 and 48-63) can be cut by the reference's
//   220-cell crop window.  In group 0 they are the lanes with l & 3 in {0, 3}; group 3 stores its pieces in the order
//   1, 0, 3, 2 (column block (l & 3) ^ 1: still whole lines per instruction), so there they are the lanes with l & 3 in
//   {1, 2}: EVERY lane owns exactly one corner run and renders it in one slot - with the tested loop only if some corner
//   run of the wave is in fact cut (its two end taps outside the window, which is convex) - and its three other runs with
//   the test-free loop.  That those cannot be cut is a property of the pixel step alone: a tap of a run outside the corner
//   blocks lies at most 31.5 |a| + 15.5 |b| (or the same with a and b exchanged) from the start cell on either axis, and
//   the SCALAR unit checks that bound against the window per car (it holds for every unit heading: 35.1 x 3.125 = 109.7
//   < 110 cells); a car that failed it would take the tested loop in all four slots.
struct PatchImage {          // the bitmap as it lies in LDS (wave-uniform): plain, or with the zero border of RC_PATCH_PAD cells
    uint32_t pitch_b, w_cells, h_cells, zero_addr;
};
template <bool CLAMP, bool NT>
__device__ __forceinline__ void patch_car(const RcParams &p, const PatchImage &img, const unsigned car, const unsigned lane, const int icx,
                                          const int icy, const int a, const int b, const bool all_tested) {
    typedef unsigned v4u_t __attribute__((ext_vector_type(4)));
    typedef const __attribute__((address_space(3))) uint8_t *lds_u8_ptr;
    const lds_u8_ptr lds_bytes = (lds_u8_ptr)(uint32_t)0;                // the bitmap starts at LDS address 0 (rck_set_lds_limits)
    const uint32_t pitch_b = img.pitch_b;
    PatchConst k;
    k.a = pin_vgpr(a); k.b = pin_vgpr(b);
    k.xmax = pin_vgpr(((img.w_cells - 1u) << 16) | 0xffffu);
    k.ymax = pin_vgpr(((img.h_cells - 1u) << 16) | 0xffffu);
    k.zero_addr = pin_vgpr(img.zero_addr);                               // a word that reads 0
    // the reference's [-110, 110) crop window around the start cell, in 16.16
    k.wx = pin_vgpr((uint32_t)(icx - RCS_PATCH_WINDOW_I) << 16);
    k.wy = pin_vgpr((uint32_t)(icy - RCS_PATCH_WINDOW_I) << 16);
    k.span = pin_vgpr(((uint32_t)(2 * RCS_PATCH_WINDOW_I) << 16) - 1u);
    k.three = pin_vgpr(3u); k.one = pin_vgpr(1u); k.sixteen = pin_vgpr(16u);
    k.ones = pin_vgpr(0x01010101u);
    const int x00 = ((63 * (-a - b)) >> 1) + icx * 65536, y00 = ((63 * (a - b)) >> 1) + icy * 65536;      // scalar
    const int rl = (int)(lane >> 2), cb = (int)(lane & 3u);
    const bool corner_first = cb == 0 || cb == 3;          // this lane's corner run is in group 0 (else in group 3)
    // start taps of a run: (row, first column c0)
    auto start = [&](int row, int c0, int &X, int &Y) {
        X = mad_i24(c0, a, mad_i24(row, b, x00));
        Y = mad_i24(c0, b, mad_i24(-row, a, y00));
    };
    auto ends_inside = [&](int X, int Y) {
        const uint32_t f0 = max((uint32_t)X - k.wx, (uint32_t)Y - k.wy);
        const uint32_t f1 = max((uint32_t)(X + 15 * k.a) - k.wx, (uint32_t)(Y + 15 * k.b) - k.wy);
        return (uint32_t)max(f0, f1) <= k.span;
    };
    // CLAMP cars (the window is not wholly inside the grid): a slot whose 64 runs all have BOTH end taps inside the grid - a
    // rectangle, convex - has every tap inside it and takes the loop without the two clamps per pixel; the others clamp
    auto slot_in_grid = [&](int X, int Y) {
        if (!CLAMP) return true;
        const uint32_t fx = max((uint32_t)X, (uint32_t)(X + 15 * (int)k.a)), fy = max((uint32_t)Y, (uint32_t)(Y + 15 * (int)k.b));
        return __builtin_amdgcn_ballot_w64(fx > k.xmax || fy > k.ymax) == 0;
    };
    // piece (row, column block) of the car's 4 KB lies at 16-byte index 4 row + block: groups 0 - 2 at 64 g + lane, group 3
    // (pieces in the order 1, 0, 3, 2) at 192 + (lane ^ 1)
    v4u_t *out = reinterpret_cast<v4u_t *>(p.out.patch) + (size_t)car * 256u;
    auto store = [&](int g, const uint32_t (&w)[4]) {
        const v4u_t px = {w[0], w[1], w[2], w[3]};
        v4u_t *dst = out + 64 * g + (g == 3 ? (lane ^ 1u) : lane);
        if (NT) __builtin_nontemporal_store(px, dst);
        else *dst = px;
    };
    // slot A: the corner run - group 0 piece cb for the lanes with cb in {0, 3}, group 3 piece cb ^ 1 for the others
    uint32_t wa[4], wb[4];
    {
        int X, Y;
        start(corner_first ? rl : 48 + rl, corner_first ? 16 * cb : 16 * (cb ^ 1), X, Y);
        // the end taps of every corner run inside the window, which is convex: so is every tap of the wave
        if (!all_tested && __builtin_amdgcn_ballot_w64(!ends_inside(X, Y)) == 0) {
            if (slot_in_grid(X, Y)) patch_run<false>(lds_bytes, X, Y, k, pitch_b, wa);
            else patch_run<CLAMP>(lds_bytes, X, Y, k, pitch_b, wa);
        } else {
            patch_run_tested<CLAMP>(lds_bytes, X, Y, k, pitch_b, wa);
        }
    }
    // slot B: the lane's piece of the OTHER outer group (columns 16 - 47: never cut); then groups 1 and 2
#pragma unroll 1
    for (int s = 0; s < 3; ++s) {
        const int row = s == 0 ? (corner_first ? 48 + rl : rl) : 16 * s + rl;
        const int c0 = (s == 0 && corner_first) ? 16 * (cb ^ 1) : 16 * cb;
        int X, Y;
        start(row, c0, X, Y);
        uint32_t w[4];
        if (all_tested) patch_run_tested<CLAMP>(lds_bytes, X, Y, k, pitch_b, w);
        else if (slot_in_grid(X, Y)) patch_run<false>(lds_bytes, X, Y, k, pitch_b, w);
        else patch_run<CLAMP>(lds_bytes, X, Y, k, pitch_b, w);
        if (s == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) wb[i] = w[i];
        } else {
            store(s, w);
        }
    }
    // (selecting the ADDRESS instead of the data - one select per store, each instruction then writing half of group 0's and half of
    // group 3's lines - saves 33 instructions per car and gains nothing: 0.0707 against 0.0701 ms, HBM writes 1.018 x; round 4)
    uint32_t g0[4], g3[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        g0[i] = corner_first ? wa[i] : wb[i];
        g3[i] = corner_first ? wb[i] : wa[i];
    }
    store(0, g0);
    store(3, g3);
}

template <bool NT>
__global__ __launch_bounds__(1024) void rc_patch_car_kernel(RcParams p, int padded_bytes) {
    extern __shared__ uint32_t lds_words[];
    const RcTrackDev &t = p.trk;
    const int pad = padded_bytes > 0 ? RC_PATCH_PAD : 0;
    PatchImage img;
    if (pad) {
        // the bitmap inside a border of `pad` zero cells (pad = 5 words: rows keep their word alignment)
        const int pw = (t.w + 2 * pad + 31) / 32, rows = t.h + 2 * pad;
        uint4 *d4 = reinterpret_cast<uint4 *>(lds_words);
        const uint4 z4 = {0u, 0u, 0u, 0u};
        for (int i = threadIdx.x; i < (padded_bytes >> 4); i += blockDim.x) d4[i] = z4;
        __syncthreads();
        for (int i = threadIdx.x; i < t.h * t.pitch; i += blockDim.x) {
            const int row = i / t.pitch, word = i - row * t.pitch;
            if (word < pw - pad / 32) lds_words[(row + pad) * pw + pad / 32 + word] = t.drv_words[i];
        }
        img.pitch_b = (uint32_t)pw * 4u; img.w_cells = (uint32_t)(pw * 32); img.h_cells = (uint32_t)rows; img.zero_addr = 0u;
    } else {
        stage_bitmap(lds_words, t.drv_words, t.h * t.pitch + 1);        // + the all-zero word behind the bitmap (rc_load_track)
        img.pitch_b = (uint32_t)t.pitch * 4u; img.w_cells = (uint32_t)t.w; img.h_cells = (uint32_t)t.h;
        img.zero_addr = (uint32_t)(t.h * t.pitch) * 4u;
    }
    __syncthreads();
    const unsigned lane = threadIdx.x & 63u;
    const unsigned waves = gridDim.x * (blockDim.x >> 6);
    typedef unsigned v4u_t __attribute__((ext_vector_type(4)));
    // cars are dealt to the waves of the grid in turn (a wave's cars are `waves` apart: neighbouring waves write
    // neighbouring patches).  The car is wave-uniform, so its header comes through the SCALAR unit: vector loads share one
    // in-order counter with the stores, and every car would start by waiting for the previous car's 4 KB to be
    // acknowledged.  (Inline assembly with its own wait: the compiler takes the load for a vector one because the kernel
    // also stores.)
    for (unsigned car = __builtin_amdgcn_readfirstlane(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)); car < (unsigned)p.n_cars;
         car = __builtin_amdgcn_readfirstlane(car + waves)) {
        typedef int v4i_t __attribute__((ext_vector_type(4)));
        v4i_t h;
        asm volatile("s_load_dwordx4 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=&s"(h) : "s"(p.st.patch_pose + car) : "memory");
        const int a = h.z, b = h.w;
        if (h.x == RC_PATCH_SKIP) {              // reset observation is all zeros, dreamer/wrappers.py:413; a diverged car sees nothing
            v4u_t *out = reinterpret_cast<v4u_t *>(p.out.patch) + (size_t)car * 256u + lane;
            const v4u_t z = {0u, 0u, 0u, 0u};
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (NT) __builtin_nontemporal_store(z, out + 64 * g);
                else out[64 * g] = z;
            }
            continue;
        }
        // No tap can leave the image, and nothing is clamped, if the crop window lies inside the grid - or, with the zero border,
        // if the car's own cell does (the border is wider than any tap is far)
        const int lx = h.x - RCS_PATCH_WINDOW_I, ly = h.y - RCS_PATCH_WINDOW_I, span = 2 * RCS_PATCH_WINDOW_I - 1;
        const bool inside = pad ? ((unsigned)h.x < (unsigned)t.w && (unsigned)(h.y - 1) < (unsigned)t.h)
                                : (lx >= 0 && ly >= 0 && lx + span <= t.w - 1 && ly + span <= t.h - 1);
        const int icx = h.x + pad, icy = h.y + pad;            // the start cell in the image's coordinates
        // runs outside the corner blocks stay inside the window: |offset| <= (63 |a| + 31 |b|) / 2 + 1 on either axis (and with a, b exchanged)
        const int aa = a < 0 ? -a : a, ab = b < 0 ? -b : b;
        const int reach = (63 * (aa > ab ? aa : ab) + 31 * (aa > ab ? ab : aa)) / 2 + 2;
        const bool all_tested = reach > RCS_PATCH_WINDOW_I * 65536 - 2;
        if (inside) patch_car<false, NT>(p, img, car, lane, icx, icy, a, b, all_tested);
        else patch_car<true, NT>(p, img, car, lane, icx, icy, a, b, all_tested);
    }
}

// Follow-the-gap on the device: one wave per car, lane l owns the 13 consecutive beams FTG_LO + 13 l ...
// Wave-level steps use shuffles only: (value, index) arg-min for the closest return, and an ordered
// tree reduction of run summaries (leading / trailing / best run of gap beams) for the widest gap.
#define FTG_LO 135
#define FTG_N 810
#define FTG_PER_LANE 13
#define FTG_BUBBLE 60
#define FTG_GAP_RANGE 2.0f      // a beam belongs to a gap if its smoothed range exceeds this [m]
#define FTG_CLIP 6.0f           // ranges are clipped here first (with the 0.19 rad lock the car must see a corner early)

struct RunSummary { int len, pre, suf, best, bstart, all; };

__device__ __forceinline__ RunSummary run_combine(const RunSummary &a, const RunSummary &b, int a_end) {
    // a covers [.., a_end), b starts at a_end; ties keep the earlier run
    RunSummary r;
    r.len = a.len + b.len;
    r.all = a.all & b.all;
    r.pre = a.all ? a.len + b.pre : a.pre;
    r.suf = b.all ? b.len + a.suf : b.suf;
    const int cross = a.suf + b.pre, cstart = a_end - a.suf;
    r.best = a.best; r.bstart = a.bstart;
    if (cross > r.best) { r.best = cross; r.bstart = cstart; }
    if (b.best > r.best) { r.best = b.best; r.bstart = b.bstart; }
    return r;
}

__global__ __launch_bounds__(256) void rc_ftg_kernel(RcParams p, float *__restrict__ actions, float motor_straight,
                                                      float motor_corner) {
    const int lane = threadIdx.x & 63;
    const int car = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (car >= p.n_cars) return;
    const float *scan = p.out.lidar + (size_t)car * RC_N_BEAMS;
    const int e0 = lane * FTG_PER_LANE;                              // first element (relative to FTG_LO)
    // The arc goes through LDS: read from memory with consecutive lanes on consecutive beams (256-byte requests; a lane
    // reading its own 17 beams directly makes every load touch 52 lines), clipped, then each lane takes its 13 beams
    // plus a halo of 2 on each side (stride 13 dwords: conflict-free).  Slots -2, -1 and >= FTG_N are zero padding.
    __shared__ float arc[4][FTG_PER_LANE * 64 + 8];
    float *row = arc[threadIdx.x >> 6] + 2;
#pragma unroll
    for (int k = 0; k < FTG_PER_LANE; ++k) {
        const int e = lane + 64 * k;
        float v = 0.0f;
        if (e < FTG_N) {
            v = scan[FTG_LO + e];
            v = v > FTG_CLIP ? FTG_CLIP : v;
        }
        row[e] = v;
    }
    if (lane < 2) { row[lane - 2] = 0.0f; row[FTG_PER_LANE * 64 + lane] = 0.0f; }
    __builtin_amdgcn_wave_barrier();                                 // one wave per car: its own LDS writes, in order
    float r[FTG_PER_LANE + 4];
#pragma unroll
    for (int k = 0; k < FTG_PER_LANE + 4; ++k) r[k] = row[e0 + k - 2];
    float sm[FTG_PER_LANE];
    float best_v = INFINITY;
    int best_i = 0x7fffffff;
#pragma unroll
    for (int k = 0; k < FTG_PER_LANE; ++k) {
        const int e = e0 + k;
        sm[k] = ((((r[k] + r[k + 1]) + r[k + 2]) + r[k + 3]) + r[k + 4]) * 0.2f;
        if (e < FTG_N && sm[k] < best_v) { best_v = sm[k]; best_i = e; }
    }
    // closest return: wave arg-min, first index wins ties
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const float ov = __shfl_xor(best_v, off);
        const int oi = __shfl_xor(best_i, off);
        if (ov < best_v || (ov == best_v && oi < best_i)) { best_v = ov; best_i = oi; }
    }
    const int closest = best_i;
    // gap beams: positive after the bubble; summarise this lane's 13 beams
    RunSummary s;
    s.len = 0; s.pre = 0; s.suf = 0; s.best = 0; s.bstart = e0; s.all = 1;
    int run = 0;
#pragma unroll
    for (int k = 0; k < FTG_PER_LANE; ++k) {
        const int e = e0 + k;
        const bool inside = e < FTG_N;
        const bool gap = inside && sm[k] > FTG_GAP_RANGE && (e < closest - FTG_BUBBLE || e > closest + FTG_BUBBLE);
        if (inside) {
            s.len += 1;
            if (gap) {
                run += 1;
                if (run > s.best) { s.best = run; s.bstart = e - run + 1; }
            } else {
                if (s.all) s.pre = run;
                s.all = 0;
                run = 0;
            }
        }
    }
    if (s.all) s.pre = run;
    s.suf = run;
    // ordered tree reduction: lane i absorbs lane i + off
    int my_end = e0 + s.len;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        RunSummary o;
        o.len = __shfl_down(s.len, off); o.pre = __shfl_down(s.pre, off); o.suf = __shfl_down(s.suf, off);
        o.best = __shfl_down(s.best, off); o.bstart = __shfl_down(s.bstart, off); o.all = __shfl_down(s.all, off);
        if ((lane & (2 * off - 1)) == 0 && lane + off < 64) {
            s = run_combine(s, o, my_end);
            my_end += o.len;
        }
    }
    if (lane == 0) {
        float motor = 0.0f, steering = 0.0f;
        if (s.best > 0) {
            const float centre = (float)FTG_LO + ((float)(2 * s.bstart + s.best - 1)) * 0.5f;
            const float angle = 2.35619449019234492885f - centre * 0.00436737625568553f;   // 135 deg - i * 270/1079 deg
            steering = clampf(angle / RCS_STEER_GAIN, -1.0f, 1.0f);          // the command that points the wheels at the gap (+ = right)
            motor = fabsf(steering) > 0.35f ? motor_corner : motor_straight;
        }
        actions[2 * car] = motor;
        actions[2 * car + 1] = steering;
    }
}

// ---- The REFERENCE's follow-the-gap law on the device (ros_agent/agents/follow_the_gap/src/agent.py:128-234 of the
// reference: disparity extender + percentile heading + P/D steering) - oracle/racecar_oracle.py, follow_the_gap_reference,
// is the binary32 spec this kernel follows operation for operation; oracle/ftg_reference_port.py restates the node in
// float64 and tests/golden/ftg_golden.npz pins both to the node's own outputs.  One wave per car; arc element
// a = ROS beam 179 + a = this build's beam 900 - a, 721 of them, lane l holds a = l + 64 k.
#define FR_FIRST 179
#define FR_N 721
#define FR_HALF 19                      // the 10-degree filter: 39 beams
#define FR_PER_LANE 12
// (binary32 values of the oracle's float64 expressions, as hexadecimal literals: no decimal rounding in between)
constexpr float kFrInc = 0x1.1e3842p-8f;                 // fp32(1.5 pi / 1079) = 0.004367367
constexpr float kFrAmin = -0x1.2d97c8p+1f;               // fp32(-0.75 pi)
constexpr float kFrLookahead = 0x1.7ba938p+2f;           // fp32(2 x 7^2 / (2 x 8.26)) = 5.9322033
constexpr float kFrW2 = 0x1.418c7p-3f;                   // fp32((1.2 x 0.3302)^2) = 0.15700614
constexpr float kFrMaxSteer = 0x1.aceeap-2f;             // fp32(24 deg)
constexpr float kFrDeg5 = 0x1.657184p-4f;                // fp32(5 deg)

__device__ __forceinline__ float fr_asin_small(float t) {            // |t| <= 0.5 (cephes asinf)
    const float z = t * t;
    const float pz = ((((4.2163199048e-2f * z + 2.4181311049e-2f) * z + 4.5470025998e-2f) * z + 7.4953002686e-2f) * z + 1.6666752422e-1f) * z;
    return pz * t + t;
}
// Correctly rounded binary32 square root for 2^-96 <= x < 2^96 (what the spec's np.sqrt is).  `__fsqrt_rn` compiles to the bare
// v_sqrt_f32 here, which is good to 1 ulp only: one scan in ~150 000 put an extension's end within that ulp of a beam index and
// the device agent's heading half a beam off the spec's (found in round 5 on the re-mapped columbia; tests/test_gpu_parity.py).
// The fix-up is the standard one: with s the instruction's result and s-, s+ its neighbours, the residuals x - s- s and x - s+ s
// (each ONE fma, exact enough to carry the sign) say on which side of s the root lies.
__device__ __forceinline__ float fr_sqrt_rn(float x) {
    const float s = __builtin_amdgcn_sqrtf(x);
    const float s_dn = __uint_as_float(__float_as_uint(s) - 1u), s_up = __uint_as_float(__float_as_uint(s) + 1u);
    const float r_dn = __builtin_fmaf(-s_dn, s, x), r_up = __builtin_fmaf(-s_up, s, x);
    float r = r_dn <= 0.0f ? s_dn : s;
    r = r_up > 0.0f ? s_up : r;
    return (x == 0.0f || !(x == x)) ? s : r;                             // (zero and NaN: the instruction's own answer)
}
// Device self-test of that (rc_selftest_sqrt): every binary32 in [lo_bits, hi_bits] against the double-precision root rounded once
__global__ __launch_bounds__(256) void rc_selftest_sqrt_kernel(uint32_t lo_bits, uint32_t hi_bits, unsigned long long *mismatches) {
    unsigned long long bad = 0;
    for (uint64_t b = (uint64_t)lo_bits + blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; b <= hi_bits; b += (uint64_t)gridDim.x * blockDim.x) {
        const float x = __uint_as_float((uint32_t)b);
        bad += __float_as_uint(fr_sqrt_rn(x)) != __float_as_uint((float)sqrt((double)x));
    }
    if (bad) atomicAdd(mismatches, bad);
}

__device__ __forceinline__ float fr_acos(float x) {                  // racecar_oracle.acos32
    const float ax = fabsf(x);
    if (!(ax <= 1.0f)) return __builtin_nanf("");
    if (ax > 0.5f) {
        const float a = 2.0f * fr_asin_small(fr_sqrt_rn((1.0f - ax) * 0.5f));
        return x < 0.0f ? 3.14159274101257324f - a : a;
    }
    return 1.57079637050628662f - fr_asin_small(x);
}
__device__ __forceinline__ float fr_angle(int a) { return (float)(FR_FIRST + a) * kFrInc + kFrAmin; }
__device__ __forceinline__ int wave_count(bool c) { return __builtin_popcountll(__builtin_amdgcn_ballot_w64(c)); }
// Reductions over the 64 lanes on the DPP paths of the vector unit (one instruction per step, no LDS crossbar): within rows of
// 16 by quad permutes and mirrors, then lane 15 of a row into the next (row_bcast:15, rows 1 and 3) and lane 31 into the upper
// half (row_bcast:31); lane 63 holds the result.
template <int CTRL, int ROWS>
__device__ __forceinline__ uint32_t dpp_pull(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, ROWS, 0xF, false); }
template <typename Op>
__device__ __forceinline__ uint32_t wave_reduce(uint32_t v, Op op) {
    v = op(v, dpp_pull<0xB1, 0xF>(v));       // quad_perm:[1,0,3,2]
    v = op(v, dpp_pull<0x4E, 0xF>(v));       // quad_perm:[2,3,0,1]
    v = op(v, dpp_pull<0x141, 0xF>(v));      // row_half_mirror
    v = op(v, dpp_pull<0x140, 0xF>(v));      // row_mirror: every lane of a row holds the row's result
    const uint32_t r1 = dpp_pull<0x142, 0xA>(v);     // (rows not named keep their own value: see the selects)
    v = (__lane_id() & 16) ? op(v, r1) : v;
    const uint32_t r2 = dpp_pull<0x143, 0xC>(v);
    v = (__lane_id() & 32) ? op(v, r2) : v;
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) { return wave_reduce(v, [](uint32_t a, uint32_t b) { return a < b ? a : b; }); }
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) { return wave_reduce(v, [](uint32_t a, uint32_t b) { return a + b; }); }

__global__ __launch_bounds__(256) void rc_ftg_reference_kernel(RcParams p, float *__restrict__ actions, float *__restrict__ prev_heading,
                                                               float dt, float *__restrict__ detail) {
    const int lane = threadIdx.x & 63;
    const int car = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (car >= p.n_cars) return;
    // Two arrays of 832 floats per wave (6.6 KB: six waves per SIMD).  `ra`: the clipped arc r[a] - overwritten by the window
    // maxima while they are built, written again from registers afterwards; `jp`: the jumps with 19 mirrored values on each
    // side (jp[i] = jump[i - 19]), kept to the end.  Both end in zeros (reads beyond the data).
    constexpr int kBuf = FR_PER_LANE * 64 + 64;
    __shared__ float lds_a[4][kBuf], lds_b[4][kBuf];
    float *ra = lds_a[threadIdx.x >> 6], *jp = lds_b[threadIdx.x >> 6];
    const float *scan = p.out.lidar + (size_t)car * RC_N_BEAMS;
    // the arc, clipped at the look-ahead distance (agent.py:141-146); consecutive lanes read consecutive beams
    float rv[FR_PER_LANE];
#pragma unroll
    for (int k = 0; k < FR_PER_LANE; ++k) {
        const int a = lane + 64 * k;
        float v = 0.0f;
        if (a < FR_N) {
            v = scan[900 - a];
            v = v > 0.0f ? v : 0.0f;                                 // (also turns a NaN into 0)
            v = v < kFrLookahead ? v : kFrLookahead;
        }
        rv[k] = v;
        ra[a] = v;
    }
    ra[FR_PER_LANE * 64 + lane] = 0.0f;
    jp[kBuf - 128 + lane] = 0.0f;                                    // [704, 768) - the jumps below overwrite what they own -
    jp[kBuf - 64 + lane] = 0.0f;                                     // and [768, 832)
    __builtin_amdgcn_wave_barrier();
    // (758 padded values: 720 jumps + 2 x 19)
    float jv[FR_PER_LANE];
#pragma unroll
    for (int k = 0; k < FR_PER_LANE; ++k) {
        const int a = lane + 64 * k;
        jv[k] = a < FR_N - 1 ? fabsf(ra[a + 1] - rv[k]) : 0.0f;     // agent.py:148
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < FR_PER_LANE; ++k) {
        const int a = lane + 64 * k;
        if (a < FR_N - 1) jp[a + FR_HALF] = jv[k];
        if (k == 0 && a < FR_HALF) jp[FR_HALF - 1 - a] = jv[k];                                  // scipy's 'reflect' border
        if (k >= 10 && a >= FR_N - 1 - FR_HALF && a < FR_N - 1) jp[2 * (FR_N - 1) - 1 - a + FR_HALF] = jv[k];
    }
    __builtin_amdgcn_wave_barrier();
    // The maximum of every 39-beam window (agent.py:154) by doubling: windows of 2, 4, 8, 16, 32 - each pass one neighbour read
    // and one maximum per element, in place in `ra` (all reads of a pass before its writes) - and 39 = 32 and 32 seven further
    // on.  The window of beam a is padded [a, a + 38].
    // (jumps are >= +0 and never NaN: their bit patterns order like the values, and an integer maximum is ONE instruction
    // where the floating-point select is a compare and a move)
    uint32_t *rau = reinterpret_cast<uint32_t *>(ra);
    const uint32_t *jpu = reinterpret_cast<const uint32_t *>(jp);
    uint32_t w[FR_PER_LANE], o[FR_PER_LANE];
#pragma unroll
    for (int k = 0; k < FR_PER_LANE; ++k) {
        const int i = lane + 64 * k;
        const uint32_t x = jpu[i], y = jpu[i + 1];
        w[k] = y > x ? y : x;
    }
#pragma unroll
    for (int k = 0; k < FR_PER_LANE; ++k) rau[lane + 64 * k] = w[k];
#pragma unroll
    for (int sft = 2; sft <= 16; sft <<= 1) {
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 0; k < FR_PER_LANE; ++k) o[k] = rau[lane + 64 * k + sft];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 0; k < FR_PER_LANE; ++k) {
            w[k] = o[k] > w[k] ? o[k] : w[k];
            rau[lane + 64 * k] = w[k];
        }
    }
    __builtin_amdgcn_wave_barrier();
    // candidates (agent.py:150-156): a jump that is the maximum of its window and exceeds 0.2 m; bit k of `cbits` = this
    // lane's element k is one
    uint32_t cbits = 0u;
#pragma unroll
    for (int k = 0; k < FR_PER_LANE; ++k) {
        const int a = lane + 64 * k;
        const uint32_t x = rau[a + 7];
        const uint32_t peak = x > w[k] ? x : w[k];                    // windows [a, a + 31] and [a + 7, a + 38] of the padded array
        if (a < FR_N - 1 && __float_as_uint(jv[k]) == peak && jv[k] > 0.2f) cbits |= 1u << k;
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < FR_PER_LANE; ++k) ra[lane + 64 * k] = rv[k];  // the arc again (rv[] becomes the adjusted arc below)
    __builtin_amdgcn_wave_barrier();
    // One candidate at a time, the wave together (their order does not matter: the extension is a minimum).  A candidate is a
    // disparity if it exceeds nine times the MEDIAN of its window (agent.py:157-159, ends repeated) - and since x -> fl(9 x) is
    // monotone, "jump > 9 median" holds exactly when at least 20 of the 39 samples satisfy "jump > 9 sample": no sorting.
    unsigned long long todo = __builtin_amdgcn_ballot_w64(cbits != 0u);
    while (todo != 0) {
        const int l = __builtin_ctzll(todo);
        const uint32_t bits = (uint32_t)__builtin_amdgcn_readlane((int)cbits, l);
        const int ac = l + 64 * __builtin_ctz(bits);                  // wave-uniform
        if (lane == l) cbits &= cbits - 1u;
        if ((bits & (bits - 1u)) == 0u) todo &= todo - 1;
        int i = ac - FR_HALF + lane;
        i = i < 0 ? 0 : (i > FR_N - 2 ? FR_N - 2 : i);
        const float mine = jp[i + FR_HALF];
        const float jc = jp[ac + FR_HALF];
        if (wave_count(lane <= 2 * FR_HALF && jc > mine * 9.0f) <= FR_HALF) continue;
        // extend the nearer side by the half-width of the vehicle as seen at that range (agent.py:165-176)
        const float near = fminf(fminf(ra[ac > 0 ? ac - 1 : 0], ra[ac]), ra[ac + 1]);
        const float two = 2.0f * (near * near);
        const float half = fr_acos((two - kFrW2) / two);
        int ia = 0, ib = 0;
        if (half == half) {
            const float a0 = fr_angle(0), at = fr_angle(ac);
            const float lo = ((at - half) - a0) / kFrInc, hi = ((at + half) - a0) / kFrInc;
            ia = (int)lo; ib = (int)hi;
            ia = ia < 0 ? 0 : (ia > FR_N - 1 ? FR_N - 1 : ia);
            ib = ib < 0 ? 0 : (ib > FR_N - 1 ? FR_N - 1 : ib);
        }
#pragma unroll
        for (int e = 0; e < FR_PER_LANE; ++e) {
            const int a = lane + 64 * e;
            if (a >= ia && a <= ib) rv[e] = __uint_as_float(min(__float_as_uint(rv[e]), __float_as_uint(near)));      // (both >= +0)
        }
    }
    float adj[FR_PER_LANE];
#pragma unroll
    for (int k = 0; k < FR_PER_LANE; ++k) adj[k] = lane + 64 * k < FR_N ? rv[k] : INFINITY;   // (slots beyond the arc: above every rank)
    // the 601st and 602nd smallest adjusted range (agent.py:183, np.percentile at q = 83.3): ranges are >= 0, so their
    // bit patterns order like the values; binary search on the pattern, counts by ballot.  [lo, lo + 2^bit) always holds
    // the wanted key (c_lo keys below it, c_hi below its end, c_lo <= 600 < c_hi): once it holds ONE key the search is
    // over - about half way for a scan's spread of ranges; ties run to the last bit.
    uint32_t key[FR_PER_LANE];
#pragma unroll
    for (int k = 0; k < FR_PER_LANE; ++k) key[k] = __float_as_uint(adj[k]);
    uint32_t x600 = 0u;
    int c_lo = 0, c_hi = FR_PER_LANE * 64, bit = 30;
    for (; bit >= 0; --bit) {
        const uint32_t trial = x600 | (1u << bit);
        int below = 0;
#pragma unroll
        for (int k = 0; k < FR_PER_LANE; ++k) below += wave_count(key[k] < trial);
        if (below <= 600) { x600 = trial; c_lo = below; } else c_hi = below;
        if (c_hi - c_lo == 1) break;
    }
    if (bit >= 0) {                                                   // the one key at or above the bucket's start
        uint32_t only = 0xffffffffu;
#pragma unroll
        for (int k = 0; k < FR_PER_LANE; ++k) only = key[k] >= x600 && key[k] < only ? key[k] : only;
        x600 = wave_min_u32(only);
    }
    int not_above = 0;
    uint32_t next = 0x7f800000u;
#pragma unroll
    for (int k = 0; k < FR_PER_LANE; ++k) {
        not_above += wave_count(key[k] <= x600);
        if (key[k] > x600 && key[k] < next) next = key[k];
    }
    const uint32_t x601 = not_above >= 602 ? x600 : wave_min_u32(next);
    // NumPy's linear interpolation at virtual index 600.0000000000001: a + (b - a) * 2^-43 in binary64; a binary32 range is
    // at or above that threshold exactly when it is at or above the threshold rounded UP to binary32
    const double a64 = (double)__uint_as_float(x600), b64 = (double)__uint_as_float(x601);
    const double thr = a64 + (b64 - a64) * 1.1368683772161603e-13;
    float thr32 = (float)thr;
    if ((double)thr32 < thr) thr32 = __uint_as_float(__float_as_uint(thr32) + 1u);      // (thr >= 0 and finite)
    int count = 0;
    uint32_t sum_k = 0u, sum_q = 0u;
#pragma unroll
    for (int k = 0; k < FR_PER_LANE; ++k) {
        const int a = lane + 64 * k;
        const bool chosen = a < FR_N && adj[k] >= thr32 && adj[k] < RCS_MAX_RANGE;           // np.digitize(...) == 2
        count += wave_count(chosen);
        sum_k += chosen ? (uint32_t)a : 0u;
        sum_q += chosen ? (uint32_t)__builtin_rintf(ra[a] * 524288.0f) : 0u;
    }
    sum_k = wave_sum_u32(sum_k);
    sum_q = wave_sum_u32(sum_q);
    if (lane == 0) {
        const float cnt = (float)count;
        const float heading = (((float)(int)sum_k / cnt) + (float)FR_FIRST) * kFrInc + kFrAmin;          // agent.py:184
        const float hd = ((float)sum_q / cnt) * (1.0f / 524288.0f);                                 // agent.py:185
        // agent.py:200-234 with PID.calculate (kp 1.4, kd 0.1): no derivative term on an episode's first command
        const float prev = p.st.fresh[car] ? __builtin_nanf("") : prev_heading[car];
        const float d_term = prev == prev ? (0.1f * (prev - heading)) / dt : 0.0f;
        float steer = 1.4f * heading - d_term;
        steer = steer > -kFrMaxSteer ? steer : -kFrMaxSteer;
        steer = steer < kFrMaxSteer ? steer : kFrMaxSteer;
        float speed = fabsf(steer) > kFrDeg5 ? 6.0f - (fabsf(steer) / kFrMaxSteer) * 1.8f : 6.0f;
        if (hd < 5.0f) { const float lim = (hd / 5.0f) * 4.0f; speed = lim < speed ? lim : speed; }
        speed = speed > 1.5f ? speed : 1.5f;
        prev_heading[car] = heading;
        // the car's actuators: target speed over its top speed, steering angle over its steering limit
        float motor = clampf(speed / RCS_MAX_VEL, -1.0f, 1.0f), steering = clampf(steer / RCS_STEER_GAIN, -1.0f, 1.0f);
        if (p.remap_actions) {                 // the caller's convention is ReduceActionSpace's (wrappers.py:128-130): invert it
            motor = ((motor - p.act_lo0) * 2.0f) / (p.act_hi0 - p.act_lo0) - 1.0f;
            steering = ((steering - p.act_lo1) * 2.0f) / (p.act_hi1 - p.act_lo1) - 1.0f;
        }
        actions[2 * car] = motor;
        actions[2 * car + 1] = steering;
        if (detail != nullptr) {
            detail[4 * car] = heading; detail[4 * car + 1] = hd; detail[4 * car + 2] = steer; detail[4 * car + 3] = speed;
        }
    }
}

__global__ __launch_bounds__(256) void rc_set_pose_kernel(RcParams p, const float *__restrict__ xyyaw) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.n_cars) return;
    const float x = xyyaw[3 * i], y = xyyaw[3 * i + 1], th = xyyaw[3 * i + 2];
    float sn, cs;
    sincos32(th, sn, cs);
    p.st.x[i] = x; p.st.y[i] = y; p.st.theta[i] = th; p.st.st[i] = sn; p.st.ct[i] = cs;
    p.st.scan_pose[i] = make_float4(x, y, cs, sn);
    if (p.render_patch) p.st.patch_pose[i] = patch_pose_of(p.trk, x, y, cs, sn, 0);
    p.st.fresh[i] = 0;
    p.out.fresh[i] = 0;
    float *pose = p.out.pose + 6 * i;
    pose[0] = x; pose[1] = y; pose[5] = th;
}

__global__ __launch_bounds__(256) void rc_random_actions_kernel(float *__restrict__ actions, int n_cars,
                                                                 uint32_t first_car, uint32_t seed_lo,
                                                                 uint32_t seed_hi, uint32_t step) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_cars) return;
    const rcd::u32x4 r = rcd::philox4x32(first_car + (uint32_t)i, step, 1u, 0u, seed_lo, seed_hi);
    const float u0 = (float)(r.x >> 8) * 5.9604644775390625e-8f;   // 2^-24
    const float u1 = (float)(r.y >> 8) * 5.9604644775390625e-8f;
    actions[2 * i] = u0 * 2.0f - 1.0f;
    actions[2 * i + 1] = u1 * 2.0f - 1.0f;
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// launchers
// Kernel timing without extra packets on the queue: the events handed to rck_set_launch_events are attached to the
// NEXT launch itself (hipExtLaunchKernelGGL: start / stop timestamps of the dispatch, what rocprofv3 reports), where a
// hipEventRecord before and after costs two barrier packets, ~3 us of device time per timed kernel.
namespace {
thread_local hipEvent_t g_ev_start = nullptr, g_ev_stop = nullptr;

template <typename K, typename... Args>
inline void launch(K kernel, dim3 grid, dim3 block, size_t lds, hipStream_t s, Args... args) {
    const hipEvent_t a = g_ev_start, b = g_ev_stop;
    g_ev_start = g_ev_stop = nullptr;
    hipExtLaunchKernelGGL(kernel, grid, block, (uint32_t)lds, s, a, b, 0u, args...);
}
}  // namespace

void rck_set_launch_events(hipEvent_t start, hipEvent_t stop) { g_ev_start = start; g_ev_stop = stop; }

hipError_t rck_build_quad_planes(const RcTrackDev &t, uint16_t *quad_rect_dev, hipStream_t s) {
    uint8_t *runs = nullptr;
    const size_t plane = (size_t)t.h * t.w;
    hipError_t e = hipMalloc((void **)&runs, 2 * plane);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(rc_build_runs_kernel, dim3((unsigned)((t.h + 255) / 256)), dim3(256), 0, s, t, runs, runs + plane);
    const long long total = (long long)plane * 4;
    hipLaunchKernelGGL(rc_build_quad_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, t, runs, runs + plane, quad_rect_dev);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(runs);
    return e;
}

hipError_t rck_build_spawn_table(const RcTrackDev &t, float4 *spawn_dev, hipStream_t s) {
    hipLaunchKernelGGL(rc_build_spawn_kernel, dim3((t.n_centerline + 255) / 256), dim3(256), 0, s, t, spawn_dev);
    return hipGetLastError();
}

hipError_t rck_build_first_table(const RcTrackDev &t, uint16_t *first_rect_dev, hipStream_t s) {
    RcFirstBin bins[RC_FIRST_BINS];
    // bin b = the float bits of the slope >> RC_FIRST_SHIFT, less RC_FIRST_BIAS: exponent -4 + b / 8 and the top three
    // mantissa bits b % 8, i.e. the slopes [2^e (1 + m / 8), 2^e (1 + (m + 1) / 8)) - eight LINEAR steps per octave
    constexpr int kPerOctave = RC_FIRST_BINS / 8, kMantBits = 23 - RC_FIRST_SHIFT;
    static_assert((1 << kMantBits) == kPerOctave && RC_FIRST_BIAS == (123u << kMantBits), "bins = exponent and top mantissa bits over 2^-4 .. 2^4");
    auto edge = [](int b) { return std::exp2(-4.0 + (double)(b / kPerOctave)) * (1.0 + (double)(b % kPerOctave) / kPerOctave); };
    for (int b = 0; b < RC_FIRST_BINS; ++b) {
        const double lo = edge(b), hi = edge(b + 1);                 // slope |dy / dx| of the bin
        const bool swap = b >= RC_FIRST_BINS / 2;
        // slope range in the bin's own frame, widened by 1e-6; the outermost bins are open-ended
        const double e1 = swap ? 1.0 / hi : lo, e2 = swap ? 1.0 / lo : hi;
        bins[b].s1 = (b == 0 || b == RC_FIRST_BINS - 1) ? 0.0f : (float)(e1 * (1.0 - 1e-6));
        bins[b].s2 = (float)(e2 * (1.0 + 1e-6));
        for (int k = 0; k < 2; ++k) {
            const double ang = std::atan(lo + (hi - lo) * (k ? 0.75 : 0.25));
            (k ? bins[b].ka1 : bins[b].ka0) = (float)(1.0 / std::cos(ang));
            (k ? bins[b].kb1 : bins[b].kb0) = (float)(1.0 / std::sin(ang));
        }
    }
    hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(c_first_bins), bins, sizeof(bins));
    if (e != hipSuccess) return e;
    const long long total = (long long)t.h * t.w * RC_FIRST_PLANES;
    if (total >= (1LL << 32)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(rc_build_first_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, t, first_rect_dev);
    return hipGetLastError();
}


// Validation of a freshly built track's tables (rc_load_track): the BOUNDED build of the default scan from every cell a
// sensor can stand in - the centre of every non-stop cell, two opposite headings, so that all 4 x 64 first-trip entries of
// the cell and both signs of every direction are used - with no output kept.  A ray that uses up its trip budget (a table
// entry that sends it in circles or off the grid's ring) is counted; the caller refuses the track if any did.
__global__ __launch_bounds__(256) void rc_validation_poses_kernel(RcTrackDev t, float4 *__restrict__ poses, uint32_t *__restrict__ count) {
    const unsigned gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (unsigned)t.h * (unsigned)t.w) return;
    const int ix = (int)(gid % (unsigned)t.w), iy = (int)(gid / (unsigned)t.w);
    if (bit_at(t.ray_words, t.pitch, ix, iy)) return;                     // stop cell (wall or ring): no sensor scans from here
    const float cx = t.org_x + ((float)ix + 0.5f) * t.res, cy = t.org_y + ((float)iy + 0.5f) * t.res;
    float sn, cs;
    sincos32(0.3f, sn, cs);
    const unsigned k = atomicAdd(count, 2u);
    poses[k] = make_float4(cx - RCS_LIDAR_X * cs, cy - RCS_LIDAR_X * sn, cs, sn);          // sensor at the cell's centre
    poses[k + 1] = make_float4(cx + RCS_LIDAR_X * cs, cy + RCS_LIDAR_X * sn, -cs, -sn);
}

hipError_t rck_validate_tables(const RcTrackDev &t, float band, hipStream_t s, unsigned long long *n_scans, unsigned *n_overruns) {
    const size_t cells = (size_t)t.h * t.w;
    float4 *poses = nullptr;
    uint32_t *counters = nullptr, host[2] = {0u, 0u};
    hipError_t e = hipMalloc((void **)&poses, 2 * cells * sizeof(float4));
    if (e == hipSuccess) e = hipMalloc((void **)&counters, 2 * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMemsetAsync(counters, 0, 2 * sizeof(uint32_t), s);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(rc_validation_poses_kernel, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, s, t, poses, counters);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(host, counters, sizeof(uint32_t), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e == hipSuccess && host[0] != 0u) {
        RcParams p{};
        p.trk = t;
        p.trk.band = band; p.trk.band_mh = band - 0.5f; p.trk.band2 = 2.0f * band;
        p.st.scan_pose = poses;
        p.num_envs = p.n_cars = (int32_t)host[0];
        p.cars_per_env = 1;
        p.scan_overrun = counters + 1;
        const int threads = 64;
        hipLaunchKernelGGL((rc_raycast_car_kernel<1, false, true>), dim3(host[0]), dim3(threads), kCarLdsBytes, s, p, 1);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpyAsync(host, counters, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
    }
    if (poses) (void)hipFree(poses);
    if (counters) (void)hipFree(counters);
    *n_scans = host[0];
    *n_overruns = host[1];
    return e;
}

// ---- rows out of a ring of arenas (rc_gather_rows: the window gather of replay.TrajectoryRing.sample) --------------------
// One wave per output row r: the record of car car_idx[r] in ring slot slot_idx[r], field by field, into the field's section
// of the output (row r of section f at out + sec[f] + r * bpc[f]).  The LiDAR row goes as 270 16-byte vectors, the small
// fields as words / bytes.
struct RcGatherRows {
    size_t src_off[RC_GATHER_MAX_FIELDS], dst_off[RC_GATHER_MAX_FIELDS];
    uint32_t bpc[RC_GATHER_MAX_FIELDS];
    int32_t n_fields;
    // rc_sample_batch (one launch for a whole training batch): fields of `obs_mask` (by position in this table) read
    // slot_obs_idx instead of slot_idx - a terminal row takes its observation from the record before it - and the first row
    // of a window that starts an episode (meta[4 w + 3]) gets `reset_word` in the fields of `reset_mask`: the reference's
    // reset row (action 0, reward 0, discount 1, time 0, progress -1: dreamer/wrappers.py:221-226).  length = 0: plain gather
    uint32_t obs_mask, reset_mask;
    uint32_t reset_word[RC_GATHER_MAX_FIELDS];
    const int32_t *slot_obs_idx, *meta;
    int32_t length;
};
__global__ __launch_bounds__(256) void rc_gather_rows_kernel(const char *__restrict__ ring, size_t slot_bytes, const int32_t *__restrict__ slot_idx,
                                                             const int32_t *__restrict__ car_idx, int n_rows, RcGatherRows g, char *__restrict__ out) {
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= n_rows) return;
    const unsigned lane = threadIdx.x & 63u;
    const size_t slot_plain = (size_t)slot_idx[row] * slot_bytes;
    const size_t slot_obs = g.length > 0 ? (size_t)g.slot_obs_idx[row] * slot_bytes : slot_plain;
    const size_t car = (size_t)car_idx[row];
    const bool reset_row = g.length > 0 && g.reset_mask != 0u && row % g.length == 0 && g.meta[4 * (row / g.length) + 3] != 0;
    for (int f = 0; f < g.n_fields; ++f) {
        const uint32_t n = g.bpc[f];
        const size_t slot = ((g.obs_mask >> f) & 1u) ? slot_obs : slot_plain;
        const char *src = ring + slot + g.src_off[f] + car * n;
        char *dst = out + g.dst_off[f] + (size_t)row * n;
        if (reset_row && ((g.reset_mask >> f) & 1u)) {          // (reset fields are 4 or 8 bytes of float32)
            for (uint32_t o = lane * 4u; o < n; o += 64u * 4u) *reinterpret_cast<uint32_t *>(dst + o) = g.reset_word[f];
            continue;
        }
        if ((n & 15u) == 0u) {                                   // (sections are 64-byte aligned and n is a multiple of 16: aligned vectors)
            for (uint32_t o = lane * 16u; o < n; o += 64u * 16u) *reinterpret_cast<v4u *>(dst + o) = *reinterpret_cast<const v4u *>(src + o);
        } else if ((n & 3u) == 0u) {
            for (uint32_t o = lane * 4u; o < n; o += 64u * 4u) *reinterpret_cast<uint32_t *>(dst + o) = *reinterpret_cast<const uint32_t *>(src + o);
        } else {
            for (uint32_t o = lane; o < n; o += 64u) dst[o] = src[o];
        }
    }
}

hipError_t rck_gather_rows(const void *ring, size_t slot_bytes, const int32_t *slot_idx, const int32_t *car_idx, int n_rows,
                           const size_t *src_off, const size_t *dst_off, const uint32_t *bpc, int n_fields, void *out, hipStream_t s,
                           const RcBatchRows *batch) {
    RcGatherRows g{};
    g.n_fields = n_fields;
    for (int f = 0; f < n_fields; ++f) { g.src_off[f] = src_off[f]; g.dst_off[f] = dst_off[f]; g.bpc[f] = bpc[f]; }
    if (batch != nullptr) {
        g.obs_mask = batch->obs_mask; g.reset_mask = batch->reset_mask; g.slot_obs_idx = batch->slot_obs_idx; g.meta = batch->meta;
        g.length = batch->length;
        for (int f = 0; f < n_fields; ++f) g.reset_word[f] = batch->reset_word[f];
    }
    hipLaunchKernelGGL(rc_gather_rows_kernel, dim3((unsigned)((n_rows + 3) / 4)), dim3(256), 0, s, (const char *)ring, slot_bytes, slot_idx, car_idx,
                       n_rows, g, (char *)out);
    return hipGetLastError();
}

// ---- the order in which the scan takes the cars (RcStateDev::order): a counting sort by progress along the track, RC_ORDER_BUCKETS
// buckets; within a bucket the order is whatever the atomics give - results do not depend on it, every car is scanned on its own.
__device__ __forceinline__ int order_bucket(float progress) {
    // a key that is not a number (rows never written) or out of range lands in an end bucket: the order stays a permutation
    const float k = progress * (float)RC_ORDER_BUCKETS;
    if (!(k >= 0.0f)) return 0;
    return k >= (float)(RC_ORDER_BUCKETS - 1) ? RC_ORDER_BUCKETS - 1 : (int)k;
}
__global__ __launch_bounds__(256) void rc_order_count_kernel(const float *__restrict__ progress, int n, uint32_t *__restrict__ counts) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) atomicAdd(&counts[order_bucket(progress[i])], 1u);
}
__global__ __launch_bounds__(RC_ORDER_BUCKETS) void rc_order_offsets_kernel(uint32_t *__restrict__ counts) {
    // exclusive prefix sum of the bucket counts, in place (one workgroup): the buckets' first positions
    __shared__ uint32_t s[RC_ORDER_BUCKETS];
    const int t = threadIdx.x;
    const uint32_t mine = counts[t];
    s[t] = mine;
    __syncthreads();
    for (int off = 1; off < RC_ORDER_BUCKETS; off <<= 1) {
        const uint32_t add = t >= off ? s[t - off] : 0u;
        __syncthreads();
        s[t] += add;
        __syncthreads();
    }
    counts[t] = s[t] - mine;
}
__global__ __launch_bounds__(256) void rc_order_place_kernel(const float *__restrict__ progress, int n, uint32_t *__restrict__ cursor,
                                                             int32_t *__restrict__ order) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t pos = atomicAdd(&cursor[order_bucket(progress[i])], 1u);
    // Workgroups go to the 8 XCDs in turn.  Plain rank order would hand every XCD each eighth car of the stretch in flight - the same
    // table lines in all eight L2s; instead XCD x gets whole regions x, x + 8, ... of RC_ORDER_REGION ranks (region g, offset o ->
    // slot ((g / 8) R + o) 8 + g mod 8): the stretch in flight is the same, each L2 holds an eighth of it (1.8 % of the scan;
    // 64 and 512 ranks per region do 1.2 %, 1 024 and more lose - the XCDs' regions then differ too much in cost).
    pos = order_slot_of_rank(pos, (uint32_t)n);
    order[pos] = i;
}

// The key of the small batches' order: 1 - mean range / 15 m of the car's last scan, i.e. long rays (many trips) first - the
// waves that start last, when the first have freed their slots, are then the short ones (a wave per car).
__global__ __launch_bounds__(256) void rc_order_cost_key_kernel(const float *__restrict__ lidar, int n, float *__restrict__ key) {
    const int car = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (car >= n) return;
    float s = 0.0f;
    for (int b = lane; b < RC_N_BEAMS; b += 64) s += lidar[(size_t)car * RC_N_BEAMS + b];
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) key[car] = 1.0f - s * (0.999f / (15.0f * RC_N_BEAMS));
}
hipError_t rck_cost_keys(const float *lidar_dev, int n_cars, float *key_dev, hipStream_t s) {
    hipLaunchKernelGGL(rc_order_cost_key_kernel, dim3((unsigned)((n_cars + 3) / 4)), dim3(256), 0, s, lidar_dev, n_cars, key_dev);
    return hipGetLastError();
}

hipError_t rck_sort_cars(const float *progress_dev, int n_cars, uint32_t *counts_dev, int32_t *order_dev, hipStream_t s) {
    hipError_t e = hipMemsetAsync(counts_dev, 0, sizeof(uint32_t) * RC_ORDER_BUCKETS, s);
    if (e != hipSuccess) return e;
    const int blocks = (n_cars + 255) / 256;
    hipLaunchKernelGGL(rc_order_count_kernel, dim3(blocks), dim3(256), 0, s, progress_dev, n_cars, counts_dev);
    hipLaunchKernelGGL(rc_order_offsets_kernel, dim3(1), dim3(RC_ORDER_BUCKETS), 0, s, counts_dev);
    hipLaunchKernelGGL(rc_order_place_kernel, dim3(blocks), dim3(256), 0, s, progress_dev, n_cars, counts_dev, order_dev);
    return hipGetLastError();
}

// ---- window starts of a replay sampler (rc_sample_windows): one wave per window.  Draw (first record, car) - Philox keyed by
// the caller's seed, counter (window, try, draw) - until the `length` records of that car from ring age t0 on stay inside one
// episode: no fresh record strictly inside, a fresh LAST record only if it is the episode's terminal one (done, written by
// auto-reset).  Lanes test the records of the window side by side.  Then the window's rows for rc_gather_rows.
__global__ __launch_bounds__(256) void rc_sample_windows_kernel(RcSampleWindows a) {
    const int win = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (win >= a.n_windows) return;
    const int lane = threadIdx.x & 63;
    int t0 = 0, car = 0;
    bool ok = false;
    for (int attempt = 0; attempt < a.max_tries && !ok; ++attempt) {
        const rcd::u32x4 r = rcd::philox4x32((uint32_t)win, (uint32_t)attempt, a.draw, 0x57494e44u, a.seed_lo, a.seed_hi);
        t0 = (int)(r.x % (uint32_t)a.n_start);
        car = (int)(r.y % (uint32_t)a.n_cars);
        bool bad = false;
        for (int j = lane; j < a.length; j += 64) {
            if (j == 0) continue;
            const size_t slot = (size_t)((a.oldest + t0 + j) % a.capacity) * a.slot_bytes;
            const bool fresh = a.ring[slot + a.fresh_off + car] != 0;
            bad |= fresh && (j < a.length - 1 || a.ring[slot + a.done_off + car] == 0);
        }
        ok = __builtin_amdgcn_ballot_w64(bad) == 0ull;
    }
    if (!ok && lane == 0) atomicAdd(a.failed, 1u);
    const size_t last = (size_t)((a.oldest + t0 + a.length - 1) % a.capacity) * a.slot_bytes;
    const bool terminal = a.length > 1 && a.ring[last + a.fresh_off + car] != 0 && a.ring[last + a.done_off + car] != 0;
    for (int j = lane; j < a.length; j += 64) {
        const int slot = (a.oldest + t0 + j) % a.capacity;
        const size_t o = (size_t)win * a.length + j;
        a.slot_idx[o] = slot;
        // a terminal row takes its OBSERVATION from the record before it: the new episode's observation is not this episode's
        a.slot_obs_idx[o] = (terminal && j == a.length - 1) ? (a.oldest + t0 + j - 1) % a.capacity : slot;
        a.car_idx[o] = car;
    }
    if (lane == 0) {
        const size_t first = (size_t)((a.oldest + t0) % a.capacity) * a.slot_bytes;
        a.meta[4 * win] = t0; a.meta[4 * win + 1] = car; a.meta[4 * win + 2] = terminal ? 1 : 0;
        a.meta[4 * win + 3] = a.ring[first + a.fresh_off + car] != 0 ? 1 : 0;          // the window starts an episode
    }
}

hipError_t rck_sample_windows(const RcSampleWindows &a, hipStream_t s) {
    hipLaunchKernelGGL(rc_sample_windows_kernel, dim3((unsigned)((a.n_windows + 3) / 4)), dim3(256), 0, s, a);
    return hipGetLastError();
}

// ---- flags of the peer-copy all-gather (rc_gather_trajectory_p2p): sequence numbers in uncached device memory that a
// PEER's kernel writes (over xGMI) and the owner's kernel polls.  Both kernels are one wave; the poll is bounded (wall
// clock) and reports a time-out instead of hanging the queue.
__global__ __launch_bounds__(64) void rc_p2p_post_kernel(RcP2pPost post) {
    // lane p stores `value` into flag p (a pointer into peer p's flag block, or null)
    const unsigned l = threadIdx.x;
    if (l < (unsigned)post.n && post.flag[l] != nullptr)
        __hip_atomic_store(post.flag[l], post.value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ __launch_bounds__(64) void rc_p2p_wait_kernel(const uint32_t *flags, int n, int skip, uint32_t value, uint32_t *timeouts,
                                                         unsigned long long limit_ticks) {
    // lane p waits until flags[p] >= value (sequence numbers only grow); every lane leaves the loop at the deadline
    const unsigned l = threadIdx.x;
    const unsigned long long t0 = wall_clock64();
    bool late = false;
    if (l < (unsigned)n && (int)l != skip) {
        while (__hip_atomic_load(flags + l, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < value) {
            if (wall_clock64() - t0 > limit_ticks) { late = true; break; }
            __builtin_amdgcn_s_sleep(8);
        }
    }
    if (late) atomicAdd(timeouts, 1u);
}

hipError_t rck_p2p_post(const RcP2pPost &post, hipStream_t s) {
    hipLaunchKernelGGL(rc_p2p_post_kernel, dim3(1), dim3(64), 0, s, post);
    return hipGetLastError();
}

hipError_t rck_p2p_wait(const uint32_t *flags, int n, int skip, uint32_t value, uint32_t *timeouts, double limit_s, hipStream_t s) {
    hipLaunchKernelGGL(rc_p2p_wait_kernel, dim3(1), dim3(64), 0, s, flags, n, skip, value, timeouts,
                       (unsigned long long)(limit_s * 1.0e8));        // wall_clock64 counts at 100 MHz
    return hipGetLastError();
}

// ---- The lab library (racecar_lab.hip: scan variants 0-6, the stamps build): not part of this library.  It is looked for next
// to this one (libracecar_lab.so in the directory libracecar_hip.so was loaded from) the first time a handle asks for one of
// its kernels, and every such request fails with the reason when it is not there.
// What both sides of the lab boundary must agree on: the sizes of the structs that cross it by pointer and the hash of the
// headers that define them (-DRC_HEADERS_ID, build.py) - in ONE string, compared as a whole.
#ifndef RC_HEADERS_ID
#define RC_HEADERS_ID "unhashed"
#endif
const char *rck_lab_abi_string() {
    static const std::string s = "RcParams " + std::to_string(sizeof(RcParams)) + " RcLaunchInfo " + std::to_string(sizeof(RcLaunchInfo)) +
                                 " headers " RC_HEADERS_ID;
    return s.c_str();
}

namespace {
struct Lab {
    void *handle = nullptr;
    int (*launch_raycast)(const RcParams *, const RcLaunchInfo *, hipStream_t, hipEvent_t, hipEvent_t) = nullptr;
    int (*set_lds_limits)(size_t) = nullptr;
    std::string why;                 // why it is not available
};
std::mutex g_lab_mutex;
Lab g_lab;
size_t g_lds_limit = 0;              // what rck_set_lds_limits was last called with (the lab's kernels get the same)

// A snapshot BY VALUE, taken under the lock (a reference to g_lab read after the lock was dropped raced with another
// thread inside lab() clearing `why` and assigning the function pointers).
Lab lab() {
    std::lock_guard<std::mutex> lock(g_lab_mutex);
    if (g_lab.handle != nullptr) return g_lab;
    g_lab.why.clear();               // (not loaded yet: look again - it may have been built since the last request)
    g_lab.launch_raycast = nullptr;
    g_lab.set_lds_limits = nullptr;
    Dl_info info;
    std::string dir = ".";
    if (dladdr(reinterpret_cast<const void *>(&rck_set_launch_events), &info) != 0 && info.dli_fname != nullptr) {
        const std::string path(info.dli_fname);
        const size_t slash = path.rfind('/');
        if (slash != std::string::npos) dir = path.substr(0, slash);
    }
    const char *over = getenv("RC_LAB_LIBRARY");
    const std::string path = over != nullptr ? std::string(over) : dir + "/libracecar_lab.so";
    void *h = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (h == nullptr) {
        const char *err = dlerror();              // (once: the call clears the message)
        g_lab.why = "the lab library is not built (" + path + ": " + std::string(err != nullptr ? err : "not found") +
                    "); build it with `python -m racing_dreamer_amd.build --lab`";
        return g_lab;
    }
    auto refuse = [&](const std::string &why) {
        g_lab.why = why;
        g_lab.launch_raycast = nullptr;
        g_lab.set_lds_limits = nullptr;
        dlclose(h);
        return g_lab;
    };
    // RcParams and RcLaunchInfo cross this boundary by pointer: a lab built against other headers reads them wrongly - wrong scans
    // or a GPU fault, no error.  The lab says what it was built against; anything else is refused.
    auto abi = reinterpret_cast<const char *(*)(void)>(dlsym(h, "rclab_abi"));
    const std::string want = rck_lab_abi_string();
    if (abi == nullptr || want != abi())
        return refuse(path + " was built against other headers (it says \"" + std::string(abi != nullptr ? abi() : "nothing: no rclab_abi") +
                      "\", this library needs \"" + want + "\"); rebuild it with `python -m racing_dreamer_amd.build --lab`");
    g_lab.launch_raycast = reinterpret_cast<decltype(g_lab.launch_raycast)>(dlsym(h, "rclab_launch_raycast"));
    g_lab.set_lds_limits = reinterpret_cast<decltype(g_lab.set_lds_limits)>(dlsym(h, "rclab_set_lds_limits"));
    if (g_lab.launch_raycast == nullptr || g_lab.set_lds_limits == nullptr)
        return refuse(path + " does not export rclab_launch_raycast / rclab_set_lds_limits");
    if (g_lds_limit != 0 && g_lab.set_lds_limits(g_lds_limit) != (int)hipSuccess) return refuse(path + ": rclab_set_lds_limits failed");
    g_lab.handle = h;
    return g_lab;
}
}  // namespace

// nullptr if the lab's kernels can be launched, else the reason (the calling thread's copy: valid until its next call)
const char *rck_lab_unavailable() {
    thread_local std::string reason;
    const Lab l = lab();
    if (l.handle != nullptr) return nullptr;
    reason = l.why;
    return reason.c_str();
}

hipError_t rck_set_lds_limits(size_t lds_bytes) {
    hipError_t e;
    const int b = (int)lds_bytes;
#define SET(k) e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, b); if (e != hipSuccess) return e;
    SET(rc_patch_car_kernel<true>)
    SET(rc_patch_car_kernel<false>)
#undef SET
    // rc_patch_car_kernel and rc_raycast_car_kernel address their dynamic LDS from LDS address 0: true only while they have
    // no static LDS
    hipFuncAttributes fa;
    for (const void *k : {reinterpret_cast<const void *>(rc_patch_car_kernel<true>), reinterpret_cast<const void *>(rc_patch_car_kernel<false>),
                          reinterpret_cast<const void *>(rc_raycast_car_kernel<1, false, false>), reinterpret_cast<const void *>(rc_raycast_car_kernel<2, false, false>),
                          reinterpret_cast<const void *>(rc_raycast_car_kernel<3, false, false>), reinterpret_cast<const void *>(rc_raycast_car_kernel<4, false, false>),
                          reinterpret_cast<const void *>(rc_raycast_car_kernel<1, true, false>), reinterpret_cast<const void *>(rc_raycast_car_kernel<2, true, false>),
                          reinterpret_cast<const void *>(rc_raycast_car_kernel<3, true, false>), reinterpret_cast<const void *>(rc_raycast_car_kernel<4, true, false>),
                          reinterpret_cast<const void *>(rc_raycast_car_kernel<1, false, true>), reinterpret_cast<const void *>(rc_raycast_car_kernel<2, false, true>),
                          reinterpret_cast<const void *>(rc_raycast_car_kernel<3, false, true>), reinterpret_cast<const void *>(rc_raycast_car_kernel<4, false, true>)}) {
        e = hipFuncGetAttributes(&fa, k);
        if (e != hipSuccess) return e;
        if (fa.sharedSizeBytes != 0) return hipErrorInvalidValue;
    }
    {
        std::lock_guard<std::mutex> lock(g_lab_mutex);
        g_lds_limit = lds_bytes > g_lds_limit ? lds_bytes : g_lds_limit;
        if (g_lab.handle != nullptr && g_lab.set_lds_limits(g_lds_limit) != (int)hipSuccess) return hipErrorInvalidValue;
    }
    return hipSuccess;
}

#define DISPATCH_A(A, ...)                                     \
    switch (A) {                                               \
        case 1: { constexpr int kA = 1; __VA_ARGS__; } break;  \
        case 2: { constexpr int kA = 2; __VA_ARGS__; } break;  \
        case 3: { constexpr int kA = 3; __VA_ARGS__; } break;  \
        default: { constexpr int kA = 4; __VA_ARGS__; } break; \
    }

hipError_t rck_launch_dynamics(const RcParams &p, float *actions, int repeat, const RcRandomActions &ra, hipStream_t s) {
    const int threads = 256, blocks = (p.num_envs + threads - 1) / threads;
    DISPATCH_A(p.cars_per_env, launch((rc_dynamics_kernel<kA>), dim3(blocks), dim3(threads), 0, s, p, actions, repeat, ra.on, ra.seed_lo, ra.seed_hi, ra.step));
    return hipGetLastError();
}

hipError_t rck_launch_dynamics_group(const RcGroup &g, int cars_per_env, int repeat, const RcRandomActions &ra, hipStream_t s) {
    const int waves = g.wave_start[g.n], blocks = (waves + 3) / 4;
    DISPATCH_A(cars_per_env, launch((rc_dynamics_group_kernel<kA>), dim3(blocks), dim3(256), 0, s, g.params, g, repeat, ra.on, ra.seed_lo, ra.seed_hi, ra.step));
    return hipGetLastError();
}

hipError_t rck_launch_raycast_group(const RcGroup &g, int cars_per_env, int split, hipStream_t s) {
    const int waves = ((g.wave_start[g.n] + 7) / 8) * 8;     // one wave per workgroup, as the single-handle scan; whole turns of the 8 XCDs
    if (split > 1) {
        DISPATCH_A(cars_per_env, launch((rc_raycast_group_kernel<kA, true>), dim3((unsigned)waves), dim3(64), (size_t)kCarLdsBytes, s, g.params, g, split));
    } else {
        DISPATCH_A(cars_per_env, launch((rc_raycast_group_kernel<kA, false>), dim3((unsigned)waves), dim3(64), (size_t)kCarLdsBytes, s, g.params, g, split));
    }
    return hipGetLastError();
}

hipError_t rck_launch_reset(const RcParams &p, const uint8_t *mask_dev, hipStream_t s) {
    const int threads = 256, blocks = (p.num_envs + threads - 1) / threads;
    DISPATCH_A(p.cars_per_env, launch((rc_reset_kernel<kA>), dim3(blocks), dim3(threads), 0, s, p, mask_dev));
    return hipGetLastError();
}

hipError_t rck_launch_raycast(const RcParams &p, const RcLaunchInfo &li, hipStream_t s) {
    const bool stamps = li.scan_stamps != nullptr && p.cars_per_env == 1;
    if (li.raycast_variant != 7 || stamps) {
        // a lab kernel (superseded variant or the instrumented build): racecar_lab.hip, loaded on first use
        const Lab l = lab();
        if (l.handle == nullptr) return hipErrorSharedObjectInitFailed;      // (rc_set_raycast_variant / rc_debug_scan_stamps refuse before it comes to this)
        const hipEvent_t a = g_ev_start, b = g_ev_stop;
        g_ev_start = g_ev_stop = nullptr;
        return (hipError_t)l.launch_raycast(&p, &li, s, a, b);
    }
    const int threads = li.car_threads, per = threads / 64;                     // waves per workgroup
    const long long waves = (long long)p.n_cars * li.car_split;
    if (li.scan_guarded) {            // a validation band is in force: the build whose trip loop counts its trips
        DISPATCH_A(p.cars_per_env, launch((rc_raycast_car_kernel<kA, false, true>), dim3((unsigned)((waves + per - 1) / per)), dim3(threads), (size_t)per * kCarLdsBytes, s, p, li.car_split));
    } else if (li.car_split > 1) {    // small batch, few waves per SIMD: prepare the next round under the first request
        DISPATCH_A(p.cars_per_env, launch((rc_raycast_car_kernel<kA, true, false>), dim3((unsigned)((waves + per - 1) / per)), dim3(threads), (size_t)per * kCarLdsBytes, s, p, li.car_split));
    } else {
        DISPATCH_A(p.cars_per_env, launch((rc_raycast_car_kernel<kA, false, false>), dim3((unsigned)((waves + per - 1) / per)), dim3(threads), (size_t)per * kCarLdsBytes, s, p, li.car_split));
    }
    return hipGetLastError();
}

hipError_t rck_launch_patch(const RcParams &p, const RcLaunchInfo &li, hipStream_t s) {
    // persistent 16-wave workgroups, the bitmap staged once per workgroup; as many as stay resident
    const int padded = (li.patch_variant & 4) ? 0 : (int)rc_patch_padded_bytes(p.trk.h, p.trk.w);      // (4: experiment, the unpadded bitmap)
    const size_t lds = padded ? (size_t)padded : li.lds_bytes;
    const int per_cu = lds <= 80 * 1024 ? 2 : 1;
    const long long need = ((long long)p.n_cars + 15) / 16, resident = (long long)li.n_cu * per_cu;
    const int blocks = (int)(need < resident ? need : resident);
    if (li.patch_variant & 2) launch(rc_patch_car_kernel<false>, dim3(blocks), dim3(1024), lds, s, p, padded);     // experiment: plain stores
    else launch(rc_patch_car_kernel<true>, dim3(blocks), dim3(1024), lds, s, p, padded);
    return hipGetLastError();
}

hipError_t rck_launch_patch_exact(const RcExactParams &p0, int chunk_cars, hipStream_t s) {
    // a chunk of cars at a time: 387 KB of spline coefficients per car in flight (two kernels per chunk: the prefilter, then the
    // rotation + resize); the launch timer spans the first launch's start to the last one's end
    const hipEvent_t a = g_ev_start, b = g_ev_stop;
    g_ev_start = g_ev_stop = nullptr;
    RcExactParams p = p0;
    for (int c0 = 0; c0 < p0.n_cars; c0 += chunk_cars) {
        const int n = p0.n_cars - c0 < chunk_cars ? p0.n_cars - c0 : chunk_cars;
        p.car0 = c0;
        hipExtLaunchKernelGGL(rc_patch_exact_prefilter_kernel, dim3(n), dim3(256), 0u, s, c0 == 0 ? a : nullptr, nullptr, 0u, p);
        if (p.check) hipExtLaunchKernelGGL(rc_patch_exact_sample_kernel<true>, dim3(n), dim3(PX_ST), 0u, s, nullptr, c0 + n >= p0.n_cars ? b : nullptr, 0u, p);
        else hipExtLaunchKernelGGL(rc_patch_exact_sample_kernel<false>, dim3(n), dim3(PX_ST), 0u, s, nullptr, c0 + n >= p0.n_cars ? b : nullptr, 0u, p);
    }
    return hipGetLastError();
}

hipError_t rck_launch_ftg(const RcParams &p, float *actions, float motor_straight, float motor_corner, hipStream_t s) {
    const int threads = 256, blocks = (p.n_cars + 3) / 4;
    launch(rc_ftg_kernel, dim3(blocks), dim3(threads), 0, s, p, actions, motor_straight, motor_corner);
    return hipGetLastError();
}

hipError_t rck_launch_ftg_reference(const RcParams &p, float *actions, float *prev_heading, float dt, float *detail, hipStream_t s) {
    const int threads = 256, blocks = (p.n_cars + 3) / 4;
    launch(rc_ftg_reference_kernel, dim3(blocks), dim3(threads), 0, s, p, actions, prev_heading, dt, detail);
    return hipGetLastError();
}

hipError_t rck_launch_set_pose(const RcParams &p, const float *xyyaw_dev, hipStream_t s) {
    const int threads = 256, blocks = (p.n_cars + threads - 1) / threads;
    launch(rc_set_pose_kernel, dim3(blocks), dim3(threads), 0, s, p, xyyaw_dev);
    return hipGetLastError();
}

hipError_t rck_launch_selftest_div6(int blocks, int threads, int per_lane, unsigned long long *mismatches_dev, hipStream_t s) {
    hipLaunchKernelGGL(rc_selftest_div6_kernel, dim3(blocks), dim3(threads), 0, s, 0x243f6a8885a308d3ull, per_lane, mismatches_dev);
    return hipGetLastError();
}

hipError_t rck_launch_selftest_sqrt(uint32_t lo_bits, uint32_t hi_bits, unsigned long long *mismatches_dev, hipStream_t s) {
    hipLaunchKernelGGL(rc_selftest_sqrt_kernel, dim3(4096), dim3(256), 0, s, lo_bits, hi_bits, mismatches_dev);
    return hipGetLastError();
}

hipError_t rck_launch_selftest_rcp(uint32_t exp_lo, uint32_t exp_hi, unsigned long long *mismatches_dev, hipStream_t s) {
    hipLaunchKernelGGL(rc_selftest_rcp_kernel, dim3(4096), dim3(256), 0, s, exp_lo, exp_hi, mismatches_dev);
    return hipGetLastError();
}

hipError_t rck_launch_random_actions(float *actions, int n_cars, uint32_t first_car, uint32_t seed_lo,
                                     uint32_t seed_hi, uint32_t step, hipStream_t s) {
    const int threads = 256, blocks = (n_cars + threads - 1) / threads;
    launch(rc_random_actions_kernel, dim3(blocks), dim3(threads), 0, s, actions, n_cars, first_car, seed_lo, seed_hi, step);
    return hipGetLastError();
}
