/*
 * racecar_oracle.c - plain-C CPU oracle of the batched racecar environment.
 * TEST INFRASTRUCTURE, NOT PRODUCT: only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may build, load or call this.  PARITY UNPINNED for the simulator core (the
 * reference's env.step() lives in the un-vendored racecar_gym + pybullet packages, see the header
 * of racecar_oracle.py); this file is the scalar restatement of the same env spec
 * as oracle/racecar_oracle.py and is pinned bit-for-bit to it by tests/test_oracle.py.
 *
 * Interface cited from the reference: step()/reset() contract dreamer/wrappers.py:62-77;
 * 1080 beams / 270 deg clockwise dreamer/tools.py:84-86; 15 m range dreamer/tools.py:274;
 * vehicle limits ros_agent/models/dreamer/racing_dreamer.py:14-16; wheelbase
 * ros_agent/agents/follow_the_gap/src/agent.py:78; task params
 * dreamer/scenarios/max_progress/columbia.yml:10; max_speed reward
 * baselines/racing/environment/tasks.py:6-18; action repeat dreamer/wrappers.py:107-116;
 * action remap dreamer/wrappers.py:128-130; time limit dreamer/wrappers.py:147-154;
 * occupancy patch dreamer/wrappers.py:390-408.
 *
 * Numerics: binary32, one IEEE operation per operator, no FMA (build with -ffp-contract=off,
 * SSE2 scalar math), no libm beyond floorf/rintf/fabsf.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define N_BEAMS 1080
#define PATCH 64
#define MAX_CARS 4
#define N_FOOT 34
#define FOOT_STEP 0.05f       /* pitch of the footprint lattice [m] */
#define DT 0.01f
#define INV_DT 100.0f
#define MAX_RANGE 15.0f
#define LIDAR_X 0.25f
#define WHEELBASE 0.3302f
#define MAX_STEER 0.42f
#define WHEEL_MAX 0.19f      /* front-wheel angle at full command; +1 = right: racecar_oracle.py, STEER_GAIN */
#define STEER_GAIN -0.19f
#define MAX_VEL 5.0f
#define ACCEL_MAX 4.0f
#define DRAG 0.8f
#define STEER_STEP 0.032f
/* The free parameters of the integrator as variables, the spec's values by default: the calibration sweeps against the
 * reference's trained agents (tools/analysis/agent_calibration.py) set them through oc_set_dynamics; nothing else does. */
static struct { float accel_max, drag, max_vel, steer_gain, steer_step; } g_dyn = {ACCEL_MAX, DRAG, MAX_VEL, STEER_GAIN, STEER_STEP};
void oc_set_dynamics(float accel_max, float drag, float max_vel, float steer_gain, float steer_step) {
    g_dyn.accel_max = accel_max; g_dyn.drag = drag; g_dyn.max_vel = max_vel; g_dyn.steer_gain = steer_gain; g_dyn.steer_step = steer_step;
}
#define BOX_CX 0.175f
#define BOX_HL 0.275f
#define BOX_HW 0.15f
#define N_CP 20
#define PROGRESS_REWARD 100.0f
#define PATCH_CELLS 3.125f
#define PATCH_WINDOW 110.0f
#define PATCH_WINDOW_I 110
#define PATCH_STEP_Q16 204800.0f   /* 3.125 cells per pixel in 16.16 fixed point */
#define BALL_GAP 12
#define GRID_LEAD 8
#define SPAWN_CLEAR_R 40         /* cells searched around a centre-line point for the nearest non-drivable cell */
#define SPAWN_MARGIN 0.60f       /* [m] footprint's farthest corner (0.474) + the two half cell diagonals (0.071) */
#define SPAWN_W_MAX 1.5f
#define HEADING_JITTER 0.35f
#define SPAWN_FOOT_R 5           /* cells searched around a footprint point of a centre-line pose without lateral room */
static const float HEADING_ROOM[6] = {0.0f, 0.0f, 0.05f, 0.155f, 0.26f, 0.35f};   /* [rad] by footprint clearance 0 .. 5 cells */
#define SPAWN_SAFE_SEARCH 256    /* several cars: bins searched forward for a start whose centre-line poses do not overlap */
#define MAX_CARS 4
#define NSTEP_MAX 16
#define PI_F 3.14159274101257324f
#define TWO_PI_F 6.28318548202514648f

typedef struct {
    const uint8_t *occ;      /* [h][w] occupancy with the sentinel ring set */
    const uint8_t *ring;     /* [h][w] 1 on the outermost cells             */
    const uint8_t *drv;      /* [h][w] drivable area                        */
    const float *progress;   /* [h][w]                                      */
    const float *centerline; /* [n][4]                                      */
    const float *beams;      /* [1080][2] cos, sin                          */
    const float *foot;       /* [34][2]                                     */
    int32_t h, w, n_centerline;
    float org_x, org_y, res, inv_res, tmax;
    const float *spawn_w;    /* [n] lateral room of a random start at centre-line point i (oc_spawn_width) */
    const int32_t *spawn_safe; /* [n] where a multi-car start drawn at bin i really goes (oc_spawn_safe) */
    const float *spawn_rows; /* [n][5] the spawn table: x, y, theta, lateral room, heading room of row i (oc_spawn_rows) */
} oc_track;

typedef struct {
    int32_t num_envs, cars_per_env;
    uint32_t first_env;
    int32_t task, laps, terminate_on_collision, remap_actions, time_limit_steps, auto_reset;
    float time_limit, collision_reward, act_lo[2], act_hi[2];
    int32_t reset_mode;
    uint32_t seed_lo, seed_hi;
    int32_t car_task[4];     /* task per car slot, -1 = `task` */
    int32_t n_steps;         /* window of task 2 (n_step_progress), 1..NSTEP_MAX sub-steps */
} oc_cfg;

typedef struct {
    float *x, *y, *theta, *ct, *st, *v, *delta, *omega, *accel, *progress;
    int32_t *lap, *cp;
    uint8_t *wall, *opp, *wrong, *done, *trunc, *fresh;
    int32_t *steps, *agent_steps;
    uint32_t *episode;
    /* per-step results */
    float *reward, *discount, *progress_total, *time, *action;
    float *out_progress;
    int32_t *out_lap, *out_cp;
    uint8_t *out_done, *out_trunc, *out_wall, *out_opp, *out_wrong;
    float *nstep_hist;       /* [n_cars][NSTEP_MAX]: total progress at sub-step s in slot s % n_steps */
} oc_state;

static inline float clampf(float d, float lo, float hi) { return d < lo ? lo : (d > hi ? hi : d); }

static inline void sincos32(float a, float *sn, float *cs) {
    const float kf = rintf(a * 0.636619772367581343f);
    const float r = ((a - kf * 1.5703125f) - kf * 4.837512969970703125e-4f) - kf * 7.54978995489188216e-8f;
    const int q = ((int)kf) & 3;
    const float z = r * r;
    const float s = r + (r * z) * (-1.6666654611e-1f + z * (8.3321608736e-3f + z * -1.9515295891e-4f));
    const float c = (1.0f - 0.5f * z) + (z * z) * (4.166664568298827e-2f + z * (-1.388731625493765e-3f + z * 2.443315711809948e-5f));
    *sn = q == 0 ? s : (q == 1 ? c : (q == 2 ? -s : -c));
    *cs = q == 0 ? c : (q == 1 ? -s : (q == 2 ? -c : s));
}

static inline float exp32(float x) {
    union { int32_t i; float f; } u;
    x = clampf(x, -80.0f, 80.0f);
    const float kf = rintf(x * 1.44269504088896341f);
    const float r = (x - kf * 0.693359375f) - kf * -2.12194440e-4f;
    const float z = r * r;
    float p = 1.9875691500e-4f;
    p = p * r + 1.3981999507e-3f;
    p = p * r + 8.3334519073e-3f;
    p = p * r + 4.1665795894e-2f;
    p = p * r + 1.6666665459e-1f;
    p = p * r + 5.0000001201e-1f;
    const float y = (p * z + r) + 1.0f;
    u.i = (((int32_t)kf) + 127) << 23;
    return y * u.f;
}

static void philox4x32(uint32_t c[4], uint32_t k0, uint32_t k1) {
    for (int i = 0; i < 10; ++i) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
        c[0] = n0; c[1] = (uint32_t)p1; c[2] = n2; c[3] = (uint32_t)p0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

static inline void cell_of(const oc_track *t, float wx, float wy, int *ix, int *iy) {
    *ix = (int)floorf((wx - t->org_x) * t->inv_res);
    *iy = (int)floorf((wy - t->org_y) * t->inv_res);
}
static inline int inb(const oc_track *t, int ix, int iy) { return ix >= 0 && ix < t->w && iy >= 0 && iy < t->h; }

static inline float progress_at(const oc_track *t, float wx, float wy) {
    int ix, iy;
    cell_of(t, wx, wy, &ix, &iy);
    return inb(t, ix, iy) ? t->progress[(size_t)iy * t->w + ix] : -1.0f;
}

void oc_random_actions(float *actions, int n_cars, uint32_t first_car, uint32_t seed_lo, uint32_t seed_hi, uint32_t step) {
    for (int i = 0; i < n_cars; ++i) {
        uint32_t c[4] = {first_car + (uint32_t)i, step, 1u, 0u};
        philox4x32(c, seed_lo, seed_hi);
        actions[2 * i] = ((float)(c[0] >> 8) * 5.9604644775390625e-8f) * 2.0f - 1.0f;
        actions[2 * i + 1] = ((float)(c[1] >> 8) * 5.9604644775390625e-8f) * 2.0f - 1.0f;
    }
}

/* Lateral room of a random start (racecar_oracle.py, spawn_width): d2 = squared cell distance from the point's cell to the
 * nearest cell that is not drivable (outside the grid included) within SPAWN_CLEAR_R cells, (R + 1)^2 if none;
 * w = clamp(isqrt(d2) * res - SPAWN_MARGIN, 0, SPAWN_W_MAX). */
void oc_spawn_width(const oc_track *t, float *out) {
    const int R = SPAWN_CLEAR_R;
    for (int i = 0; i < t->n_centerline; ++i) {
        int ix, iy;
        cell_of(t, t->centerline[4 * i], t->centerline[4 * i + 1], &ix, &iy);
        int d2 = (R + 1) * (R + 1);
        if (!inb(t, ix, iy)) d2 = 0;
        else
            for (int dy = -R; dy <= R; ++dy)
                for (int dx = -R; dx <= R; ++dx) {
                    const int jx = ix + dx, jy = iy + dy;
                    const int blocked = !inb(t, jx, jy) || !t->drv[(size_t)jy * t->w + jx];
                    if (blocked && dx * dx + dy * dy < d2) d2 = dx * dx + dy * dy;
                }
        int k = 0;
        while ((k + 1) * (k + 1) <= d2) ++k;
        out[i] = clampf((float)k * t->res - SPAWN_MARGIN, 0.0f, SPAWN_W_MAX);
    }
}

/* The 34 footprint probes of H5 on a pose (racecar_oracle.py, _wall_hit_poses): cell of probe n into (*ix, *iy). */
static inline void foot_cell(const oc_track *t, float x, float y, float ct, float st, int n, int *ix, int *iy) {
    const float k = FOOT_STEP * t->inv_res;
    const float gx = (x - t->org_x) * t->inv_res, gy = (y - t->org_y) * t->inv_res;
    const int32_t ex = (int32_t)rintf((ct * k) * 65536.0f), ey = (int32_t)rintf((st * k) * 65536.0f);
    const int32_t x0 = (int32_t)rintf(gx * 65536.0f), y0 = (int32_t)rintf(gy * 65536.0f);
    const int li = n < 24 ? n % 12 : (n < 29 ? 0 : 11);
    const int lj = n < 12 ? 0 : (n < 24 ? 6 : (n < 29 ? n - 23 : n - 28));
    *ix = (x0 + (li - 2) * ex - (lj - 3) * ey) >> 16;
    *iy = (y0 + (li - 2) * ey + (lj - 3) * ex) >> 16;
}

/* racecar_oracle.py, spawn_usable: the centre-line pose of bin i touches no wall. */
static int bin_usable(const oc_track *t, int i) {
    float sn, cs;
    sincos32(t->centerline[4 * i + 2], &sn, &cs);
    for (int n = 0; n < N_FOOT; ++n) {
        int ix, iy;
        foot_cell(t, t->centerline[4 * i], t->centerline[4 * i + 1], cs, sn, n, &ix, &iy);
        if (!inb(t, ix, iy) || t->occ[(size_t)iy * t->w + ix]) return 0;
    }
    return 1;
}

/* racecar_oracle.py, spawn_heading_room: HEADING_JITTER where bin i has lateral room `w`, else HEADING_ROOM[k], k = the
 * smallest over the 34 footprint points of isqrt(squared cell distance to the nearest non-drivable cell within SPAWN_FOOT_R). */
static float bin_heading_room(const oc_track *t, int i, float w) {
    if (w > 0.0f) return HEADING_JITTER;
    const int R = SPAWN_FOOT_R;
    float sn, cs;
    sincos32(t->centerline[4 * i + 2], &sn, &cs);
    int kmin = R;
    for (int n = 0; n < N_FOOT; ++n) {
        int ix, iy;
        foot_cell(t, t->centerline[4 * i], t->centerline[4 * i + 1], cs, sn, n, &ix, &iy);
        int d2 = (R + 1) * (R + 1);
        if (!inb(t, ix, iy)) d2 = 0;
        else
            for (int dy = -R; dy <= R; ++dy)
                for (int dx = -R; dx <= R; ++dx) {
                    const int jx = ix + dx, jy = iy + dy;
                    const int blocked = !inb(t, jx, jy) || !t->drv[(size_t)jy * t->w + jx];
                    if (blocked && dx * dx + dy * dy < d2) d2 = dx * dx + dy * dy;
                }
        int k = 0;
        while ((k + 1) * (k + 1) <= d2) ++k;
        kmin = k < kmin ? k : kmin;
    }
    return HEADING_ROOM[kmin];
}

/* The spawn table (racecar_oracle.py, spawn_rows): row i = bin u(i), the first usable bin among i, i + 1, ... (around the lap,
 * SPAWN_SAFE_SEARCH of them; i itself if none): x, y, theta, lateral room, heading room.  Needs t->spawn_w. */
void oc_spawn_rows(const oc_track *t, float *out) {
    const int n = t->n_centerline;
    for (int i = 0; i < n; ++i) {
        int u = i;
        for (int k = 0; k < SPAWN_SAFE_SEARCH && k < n; ++k)
            if (bin_usable(t, (i + k) % n)) { u = (i + k) % n; break; }
        float *r = out + 5 * (size_t)i;
        r[0] = t->centerline[4 * u]; r[1] = t->centerline[4 * u + 1]; r[2] = t->centerline[4 * u + 2];
        r[3] = t->spawn_w[u];
        r[4] = bin_heading_room(t, u, t->spawn_w[u]);
    }
}

static inline float unit_pm1(uint32_t w) { return ((float)(w >> 8) * 5.9604644775390625e-8f) * 2.0f - 1.0f; }   /* [-1, 1), exact */

static int obb_overlap_pose(const float *pa, const float *pb) {          /* pose = x, y, theta, sin, cos */
    const float cta = pa[4], sta = pa[3], ctb = pb[4], stb = pb[3];
    const float ax = pa[0] + BOX_CX * cta, ay = pa[1] + BOX_CX * sta;
    const float bx = pb[0] + BOX_CX * ctb, by = pb[1] + BOX_CX * stb;
    const float dx = bx - ax, dy = by - ay;
    const float c = fabsf(cta * ctb + sta * stb);
    const float s = fabsf(sta * ctb - cta * stb);
    const float ra = BOX_HL + (BOX_HL * c + BOX_HW * s);
    const float rb = BOX_HW + (BOX_HL * s + BOX_HW * c);
    int sep = fabsf(dx * cta + dy * sta) > ra;
    sep |= fabsf(dy * cta - dx * sta) > rb;
    sep |= fabsf(dx * ctb + dy * stb) > ra;
    sep |= fabsf(dy * ctb - dx * stb) > rb;
    return !sep;
}

/* Where a multi-car start drawn at bin i goes (racecar_oracle.py, spawn_safe): the first bin j among i, i + 1, ...,
 * i + SPAWN_SAFE_SEARCH - 1 (around the lap) at which the centre-line poses of MAX_CARS cars, BALL_GAP bins apart, touch no wall
 * and do not overlap pairwise; i itself if there is none. */
void oc_spawn_safe(const oc_track *t, int32_t *out) {
    const int n = t->n_centerline;
    uint8_t *sound = (uint8_t *)malloc((size_t)n);
    for (int j = 0; j < n; ++j) {
        float pose[MAX_CARS][5];
        for (int a = 0; a < MAX_CARS; ++a) {
            int idx = (j - a * BALL_GAP) % n;
            if (idx < 0) idx += n;
            pose[a][0] = t->centerline[4 * idx]; pose[a][1] = t->centerline[4 * idx + 1]; pose[a][2] = t->centerline[4 * idx + 2];
            sincos32(pose[a][2], &pose[a][3], &pose[a][4]);
        }
        int clash = 0;
        for (int a = 0; a < MAX_CARS; ++a) {
            int idx = (j - a * BALL_GAP) % n;
            if (idx < 0) idx += n;
            clash |= !bin_usable(t, idx);                       /* a bin whose centre-line pose touches a wall anchors nothing */
        }
        for (int a = 0; a < MAX_CARS; ++a)
            for (int b = a + 1; b < MAX_CARS; ++b) clash |= obb_overlap_pose(pose[a], pose[b]);
        sound[j] = (uint8_t)!clash;
    }
    for (int i = 0; i < n; ++i) {
        out[i] = i;
        for (int k = 0; k < SPAWN_SAFE_SEARCH && k < n; ++k)
            if (sound[(i + k) % n]) { out[i] = (i + k) % n; break; }
    }
    free(sound);
}

/* Reset law (H6; racecar_oracle.py, _reset_envs): bin from word 0; car a at row idx0 - a * BALL_GAP of the spawn table, moved
 * sideways by u * (the row's lateral room) and turned by v * (its heading room); if two proposed cars overlap, all cars of the env
 * take the centre-line poses. */
static void reset_env(const oc_track *t, const oc_cfg *c, oc_state *s, int e) {
    const int A = c->cars_per_env, n = t->n_centerline;
    uint32_t r[3][4];
    for (int k = 0; k < 1 + A / 2; ++k) {
        r[k][0] = c->first_env + (uint32_t)e; r[k][1] = s->episode[e]; r[k][2] = (uint32_t)k; r[k][3] = 0u;
        philox4x32(r[k], c->seed_lo, c->seed_hi);
    }
    s->episode[e] += 1u;
    const int jitter = c->reset_mode != 0;
    int idx0 = !jitter ? BALL_GAP * (A - 1) + GRID_LEAD : (int)(((uint64_t)r[0][0] * (uint64_t)n) >> 32);
    if (A > 1) idx0 = t->spawn_safe[idx0];                     /* never anchor several cars where the centre line folds (the grid too) */
    float centre[4][5], prop[4][5];
    for (int a = 0; a < A; ++a) {
        int idx = (idx0 - a * BALL_GAP) % n;
        if (idx < 0) idx += n;
        float *ce = centre[a], *pr = prop[a];
        const float *row = t->spawn_rows + 5 * (size_t)idx;
        ce[0] = row[0]; ce[1] = row[1]; ce[2] = row[2];
        sincos32(ce[2], &ce[3], &ce[4]);
        if (!jitter) { for (int k = 0; k < 5; ++k) pr[k] = ce[k]; continue; }
        const uint32_t wu = a == 0 ? r[0][1] : r[1 + (a - 1) / 2][2 * ((a - 1) % 2)];
        const uint32_t wv = a == 0 ? r[0][2] : r[1 + (a - 1) / 2][2 * ((a - 1) % 2) + 1];
        const float off = unit_pm1(wu) * row[3];
        pr[0] = ce[0] - off * ce[3];
        pr[1] = ce[1] + off * ce[4];
        float th = ce[2] + unit_pm1(wv) * row[4];
        th = th > PI_F ? th - TWO_PI_F : th;
        th = th < -PI_F ? th + TWO_PI_F : th;
        pr[2] = th;
        sincos32(th, &pr[3], &pr[4]);
    }
    int clash = 0;
    if (jitter)
        for (int a = 0; a < A; ++a)
            for (int b = a + 1; b < A; ++b) clash |= obb_overlap_pose(prop[a], prop[b]);
    for (int a = 0; a < A; ++a) {
        const int i = e * A + a;
        const float *p = clash ? centre[a] : prop[a];
        s->x[i] = p[0]; s->y[i] = p[1]; s->theta[i] = p[2]; s->st[i] = p[3]; s->ct[i] = p[4];
        float pr = progress_at(t, s->x[i], s->y[i]);
        pr = pr < 0.0f ? 0.0f : pr;
        s->progress[i] = pr;
        const int cp = (int)(pr * (float)N_CP);
        s->cp[i] = cp < N_CP - 1 ? cp : N_CP - 1;
        s->v[i] = s->delta[i] = s->omega[i] = s->accel[i] = 0.0f;
        s->wall[i] = s->opp[i] = s->wrong[i] = s->done[i] = s->trunc[i] = 0;
        s->lap[i] = 1;
        s->fresh[i] = 1;
        for (int k = 0; k < NSTEP_MAX; ++k) s->nstep_hist[(size_t)i * NSTEP_MAX + k] = pr;
    }
    s->steps[e] = 0;
    s->agent_steps[e] = 0;
}

static void store_results(const oc_cfg *c, oc_state *s, int e) {
    const int A = c->cars_per_env;
    const float time = (float)s->steps[e] * DT;
    for (int a = 0; a < A; ++a) {
        const int i = e * A + a;
        s->discount[i] = 1.0f - (float)s->done[i];
        s->progress_total[i] = (float)(s->lap[i] - 1) + s->progress[i];
        s->time[i] = time;
        s->out_progress[i] = s->progress[i];
        s->out_lap[i] = s->lap[i];
        s->out_cp[i] = s->cp[i];
        s->out_done[i] = s->done[i];
        s->out_trunc[i] = s->trunc[i];
        s->out_wall[i] = s->wall[i];
        s->out_opp[i] = s->opp[i];
        s->out_wrong[i] = s->wrong[i];
    }
}

void oc_reset(const oc_track *t, const oc_cfg *c, oc_state *s, const uint8_t *mask) {
    for (int e = 0; e < c->num_envs; ++e) {
        if (mask && !mask[e]) continue;
        reset_env(t, c, s, e);
        for (int a = 0; a < c->cars_per_env; ++a) {
            const int i = e * c->cars_per_env + a;
            s->reward[i] = 0.0f;
            s->action[2 * i] = s->action[2 * i + 1] = 0.0f;
        }
        store_results(c, s, e);
    }
}

/* Wall contact (H5): the 34 border points of the 12 x 7 body lattice in 16.16 fixed-point cell coordinates - see
 * racecar_oracle.py, _wall_hit. */
static int wall_hit(const oc_track *t, const oc_state *s, int i) {
    const float k = FOOT_STEP * t->inv_res;
    const float gx = (s->x[i] - t->org_x) * t->inv_res, gy = (s->y[i] - t->org_y) * t->inv_res;
    if (!(fabsf(gx) <= 8192.0f && fabsf(gy) <= 8192.0f)) return 1;
    const int32_t ex = (int32_t)rintf((s->ct[i] * k) * 65536.0f), ey = (int32_t)rintf((s->st[i] * k) * 65536.0f);
    const int32_t x0 = (int32_t)rintf(gx * 65536.0f), y0 = (int32_t)rintf(gy * 65536.0f);
    int hit = 0;
    for (int n = 0; n < N_FOOT; ++n) {
        const int li = n < 24 ? n % 12 : (n < 29 ? 0 : 11);
        const int lj = n < 12 ? 0 : (n < 24 ? 6 : (n < 29 ? n - 23 : n - 28));
        const int32_t px = x0 + (li - 2) * ex - (lj - 3) * ey, py = y0 + (li - 2) * ey + (lj - 3) * ex;
        const int ix = px >> 16, iy = py >> 16;                 /* arithmetic shifts (gcc): floor */
        hit |= inb(t, ix, iy) ? t->occ[(size_t)iy * t->w + ix] : 1;
    }
    return hit;
}

static int obb_overlap(const oc_state *s, int a, int b) {
    const float ax = s->x[a] + BOX_CX * s->ct[a], ay = s->y[a] + BOX_CX * s->st[a];
    const float bx = s->x[b] + BOX_CX * s->ct[b], by = s->y[b] + BOX_CX * s->st[b];
    const float dx = bx - ax, dy = by - ay;
    const float c = fabsf(s->ct[a] * s->ct[b] + s->st[a] * s->st[b]);
    const float sn = fabsf(s->st[a] * s->ct[b] - s->ct[a] * s->st[b]);
    const float ra = BOX_HL + (BOX_HL * c + BOX_HW * sn);
    const float rb = BOX_HW + (BOX_HL * sn + BOX_HW * c);
    int sep = fabsf(dx * s->ct[a] + dy * s->st[a]) > ra;
    sep |= fabsf(dy * s->ct[a] - dx * s->st[a]) > rb;
    sep |= fabsf(dx * s->ct[b] + dy * s->st[b]) > ra;
    sep |= fabsf(dy * s->ct[b] - dx * s->st[b]) > rb;
    return !sep;
}

/* One agent step (up to `repeat` sub-steps) for envs [e0, e1). */
void oc_step_range(const oc_track *t, const oc_cfg *c, oc_state *s, const float *actions, int repeat, int e0, int e1) {
    const int A = c->cars_per_env;
    for (int e = e0; e < e1; ++e) {
        float motor[MAX_CARS], steer[MAX_CARS];
        int any_done = 0;
        for (int a = 0; a < A; ++a) {
            const int i = e * A + a;
            const float a0 = actions[2 * i], a1 = actions[2 * i + 1];
            s->action[2 * i] = a0;
            s->action[2 * i + 1] = a1;
            float m = a0, st = a1;
            if (c->remap_actions) {
                m = ((a0 + 1.0f) * 0.5f) * (c->act_hi[0] - c->act_lo[0]) + c->act_lo[0];
                st = ((a1 + 1.0f) * 0.5f) * (c->act_hi[1] - c->act_lo[1]) + c->act_lo[1];
            }
            motor[a] = clampf(m, -1.0f, 1.0f);
            steer[a] = clampf(st, -1.0f, 1.0f);
            any_done |= s->done[i];
            s->reward[i] = 0.0f;
            s->fresh[i] = 0;
        }
        if (!any_done) {
            for (int sub = 0; sub < repeat; ++sub) {
                for (int a = 0; a < A; ++a) {
                    const int i = e * A + a;
                    const float m = motor[a];
                    const float force = fabsf(m) * g_dyn.accel_max;
                    const float acc = (m >= 0.0f ? force : -force) - g_dyn.drag * s->v[i];
                    const float v = clampf(s->v[i] + acc * DT, 0.0f, g_dyn.max_vel);
                    const float dd = clampf(steer[a] * g_dyn.steer_gain - s->delta[i], -g_dyn.steer_step, g_dyn.steer_step);
                    const float dl = s->delta[i] + dd;
                    float sd, cd;
                    sincos32(dl, &sd, &cd);
                    const float om = (v / WHEELBASE) * (sd / cd);
                    s->x[i] = s->x[i] + (v * s->ct[i]) * DT;
                    s->y[i] = s->y[i] + (v * s->st[i]) * DT;
                    float th = s->theta[i] + om * DT;
                    th = th > PI_F ? th - TWO_PI_F : th;
                    th = th < -PI_F ? th + TWO_PI_F : th;
                    s->theta[i] = th;
                    sincos32(th, &s->st[i], &s->ct[i]);
                    s->v[i] = v; s->delta[i] = dl; s->omega[i] = om;
                    s->accel[i] = acc;
                }
                s->steps[e] += 1;
                for (int a = 0; a < A; ++a) {
                    s->wall[e * A + a] = (uint8_t)wall_hit(t, s, e * A + a);
                    s->opp[e * A + a] = 0;
                }
                for (int a = 0; a < A; ++a)
                    for (int b = a + 1; b < A; ++b) {
                        const int o = obb_overlap(s, e * A + a, e * A + b);
                        s->opp[e * A + a] |= (uint8_t)o;
                        s->opp[e * A + b] |= (uint8_t)o;
                    }
                const float time = (float)s->steps[e] * DT;
                int stop = 0;
                for (int a = 0; a < A; ++a) {
                    const int i = e * A + a;
                    float p_new = progress_at(t, s->x[i], s->y[i]);
                    const float p_old = s->progress[i];
                    const int lap_old = s->lap[i], cp_old = s->cp[i];
                    p_new = p_new >= 0.0f ? p_new : p_old;
                    int cp_new = (int)(p_new * (float)N_CP);
                    cp_new = cp_new < N_CP - 1 ? cp_new : N_CP - 1;
                    int d = cp_new - cp_old;
                    d = d < 0 ? d + N_CP : d;
                    const int fwd = d > 0 && d <= N_CP / 2, bwd = d > N_CP / 2;
                    const int lap = lap_old + ((fwd && cp_new < cp_old) ? 1 : 0) - ((bwd && cp_new > cp_old) ? 1 : 0);
                    s->wrong[i] = fwd ? 0 : (bwd ? 1 : s->wrong[i]);
                    s->cp[i] = (fwd || bwd) ? cp_new : cp_old;
                    s->lap[i] = lap;
                    s->progress[i] = p_new;
                    const int collided = s->wall[i] | s->opp[i];
                    float r;
                    int done;
                    const int task = c->car_task[a] < 0 ? c->task : c->car_task[a];
                    if (task == 2) {      /* n_step_progress: progress over the last n_steps sub-steps, never done */
                        float *h = s->nstep_hist + (size_t)i * NSTEP_MAX + (s->steps[e] % c->n_steps);
                        const float total = (float)(lap - 1) + p_new;
                        r = (total - *h) * PROGRESS_REWARD;
                        *h = total;
                        done = 0;
                    } else if (task == 0) {
                        const float delta = (float)(lap - lap_old) + (p_new - p_old);
                        r = delta * PROGRESS_REWARD + (collided ? c->collision_reward : 0.0f);
                        done = (collided && c->terminate_on_collision) || lap > c->laps || time > c->time_limit;
                    } else {
                        r = s->wall[i] ? -1.0f : -exp32(fabsf(steer[a]) - s->v[i]);
                        done = 0;
                    }
                    s->reward[i] = s->reward[i] + r;
                    s->done[i] = (uint8_t)done;
                    stop |= done;
                }
                if (stop) break;
            }
            s->agent_steps[e] += 1;
            if (c->time_limit_steps > 0 && s->agent_steps[e] >= c->time_limit_steps)
                for (int a = 0; a < A; ++a) { s->done[e * A + a] = 1; s->trunc[e * A + a] = 1; }
        }
        store_results(c, s, e);
        if (c->auto_reset) {
            int fin = 0;
            for (int a = 0; a < A; ++a) fin |= s->done[e * A + a];
            if (fin) reset_env(t, c, s, e);
        }
    }
}

static float ray_vs_car(float lx, float ly, float dx, float dy, float ox, float oy, float ct2, float st2) {
    const float cx = ox + BOX_CX * ct2, cy = oy + BOX_CX * st2;
    const float rx = lx - cx, ry = ly - cy;
    const float p[2] = {rx * ct2 + ry * st2, ry * ct2 - rx * st2};
    const float e[2] = {dx * ct2 + dy * st2, dy * ct2 - dx * st2};
    const float h[2] = {BOX_HL, BOX_HW};
    float tn = -INFINITY, tf = INFINITY;
    int miss = 0;
    for (int k = 0; k < 2; ++k) {
        if (e[k] != 0.0f) {
            const float inv = 1.0f / e[k];
            const float t1 = (-h[k] - p[k]) * inv, t2 = (h[k] - p[k]) * inv;
            const float lo = t1 < t2 ? t1 : t2, hi = t1 < t2 ? t2 : t1;
            tn = lo > tn ? lo : tn;
            tf = hi < tf ? hi : tf;
        } else {
            miss |= fabsf(p[k]) > h[k];
        }
    }
    const int hit = !miss && tn <= tf && tf >= 0.0f;
    const float tt = tn > 0.0f ? tn : 0.0f;
    return (hit && tt < MAX_RANGE) ? tt : INFINITY;
}

static float cast_ray(const oc_track *t, float gx, float gy, float dx, float dy) {
    int ix = (int)floorf(gx), iy = (int)floorf(gy);
    if (!inb(t, ix, iy) || t->occ[(size_t)iy * t->w + ix]) return 0.0f;
    const float idx = dx != 0.0f ? 1.0f / dx : 0.0f, idy = dy != 0.0f ? 1.0f / dy : 0.0f;
    const int sx = dx > 0.0f ? 1 : -1, sy = dy > 0.0f ? 1 : -1;
    float bx = (float)(ix + (dx > 0.0f ? 1 : 0)), by = (float)(iy + (dy > 0.0f ? 1 : 0));
    float tx = dx != 0.0f ? (bx - gx) * idx : INFINITY;
    float ty = dy != 0.0f ? (by - gy) * idy : INFINITY;
    for (;;) {
        const int stepx = tx < ty;
        const float tt = stepx ? tx : ty;
        if (tt >= t->tmax) return MAX_RANGE;
        if (stepx) { ix += sx; bx += (float)sx; tx = (bx - gx) * idx; }
        else       { iy += sy; by += (float)sy; ty = (by - gy) * idy; }
        const size_t c = (size_t)iy * t->w + ix;
        if (t->occ[c]) return t->ring[c] ? MAX_RANGE : tt * t->res;
    }
}

/* LiDAR scans of cars [c0, c1). */
void oc_raycast_range(const oc_track *t, const oc_cfg *c, const oc_state *s, float *lidar, int c0, int c1) {
    const int A = c->cars_per_env;
    for (int car = c0; car < c1; ++car) {
        const float ct = s->ct[car], st = s->st[car];
        const float lx = s->x[car] + LIDAR_X * ct, ly = s->y[car] + LIDAR_X * st;
        const float gx = (lx - t->org_x) * t->inv_res, gy = (ly - t->org_y) * t->inv_res;
        for (int b = 0; b < N_BEAMS; ++b) {
            const float cb = t->beams[2 * b], sb = t->beams[2 * b + 1];
            const float dx = ct * cb - st * sb, dy = st * cb + ct * sb;
            float rng = cast_ray(t, gx, gy, dx, dy);
            if (A > 1) {
                const int env = car / A;
                for (int o = 0; o < A; ++o) {
                    const int oc = env * A + o;
                    if (oc == car) continue;
                    const float tc = ray_vs_car(lx, ly, dx, dy, s->x[oc], s->y[oc], s->ct[oc], s->st[oc]);
                    rng = tc < rng ? tc : rng;
                }
            }
            lidar[(size_t)car * N_BEAMS + b] = rng;
        }
    }
}

/* lidar_occupancy patches of cars [c0, c1) (dreamer/wrappers.py:390-408): the fixed-point tap walk described in
 * racecar_oracle.py, render_patch. */
void oc_patch_range(const oc_track *t, const oc_state *s, uint8_t *patch, int c0, int c1) {
    for (int car = c0; car < c1; ++car) {
        uint8_t *out = patch + (size_t)car * PATCH * PATCH;
        if (s->fresh[car]) { memset(out, 0, PATCH * PATCH); continue; }
        const int32_t a = (int32_t)rintf(s->ct[car] * PATCH_STEP_Q16), b = (int32_t)rintf(s->st[car] * PATCH_STEP_Q16);
        int icx, icy;
        cell_of(t, s->x[car], s->y[car], &icx, &icy);
        const int32_t x00 = (63 * (-a - b)) >> 1, y00 = (63 * (a - b)) >> 1;      /* arithmetic shifts (gcc): floor */
        for (int row = 0; row < PATCH; ++row) {
            int32_t X = x00 + row * b, Y = y00 - row * a;
            for (int col = 0; col < PATCH; ++col) {
                const int fx = X >> 16, fy = Y >> 16;
                const int inwin = fx >= -PATCH_WINDOW_I && fx < PATCH_WINDOW_I && fy >= -PATCH_WINDOW_I && fy < PATCH_WINDOW_I;
                const int ix = icx + fx, iy = (icy + 1) + fy;
                out[row * PATCH + col] = (inwin && inb(t, ix, iy)) ? t->drv[(size_t)iy * t->w + ix] : 0;
                X += a;
                Y += b;
            }
        }
    }
}

/* ------------------------------------------------------------------------------------------------------------------------
 * obs_type lidar_occupancy_reference: the reference's OccupancyMapObs.step (dreamer/wrappers.py:396-406 = to_pixel, 220 x 220
 * crop, scipy.ndimage.rotate, centre 200 x 200, PIL resize to 64 x 64) restated down to the binary64 operation.  The spec, with
 * the library lines each step follows, is oracle/patch_reference.py (render_patch_exact); this is the same arithmetic in C. */
#define PX_CROP 220
#define PX_WIN 200
#define PX_Z (-0.2679491924311227)            /* scipy ni_splines.c: the pole of the cubic spline, sqrt(3) - 2 correctly rounded */
#define PX_ZN (-5.539710763905135e-126)       /* pow(PX_Z, 219) */
#define PX_PI180 1.74532925199432957692e-2
#define PX_KSIZE 15                           /* ceil(2 * 3.125) * 2 + 1 */
#define PX_BITS 22

typedef struct {
    int32_t fh, r_top, c0;                    /* source image height; north-up pixel (R, C) = cell (gx, gy) = (C - c0, r_top - R) */
    double ox, oy, res;                       /* world position of the full frame's lower left corner; metres per cell */
} oc_frame;

static void px_filter_line(double *c, int stride) {          /* one line of PX_CROP samples, `stride` doubles apart */
    const double z = PX_Z, zn = PX_ZN;
    const int n = PX_CROP;
    const double gain = (1.0 - 1.0 / z) * (1.0 - z);
    for (int i = 0; i < n; ++i) c[(size_t)i * stride] *= gain;
    double c0 = c[0] + zn * c[(size_t)(n - 1) * stride], zi = z;
    for (int i = 1; i < n - 1; ++i) {
        c0 = c0 + zi * (c[(size_t)i * stride] + zn * c[(size_t)(n - 1 - i) * stride]);
        zi *= z;
    }
    c[0] = c0 / (1.0 - zn * zn);
    for (int i = 1; i < n; ++i) c[(size_t)i * stride] += z * c[(size_t)(i - 1) * stride];
    c[(size_t)(n - 1) * stride] = (z * c[(size_t)(n - 2) * stride] + c[(size_t)(n - 1) * stride]) * z / (z * z - 1.0);
    for (int i = n - 2; i >= 0; --i) c[(size_t)i * stride] = z * (c[(size_t)(i + 1) * stride] - c[(size_t)i * stride]);
}

static void px_sincos_deg(double x, double *cosv, double *sinv) {       /* x >= 0 degrees: patch_reference.py, sincos_degrees */
    double y = floor(x / 45.0);
    int j = (int)(y - 8.0 * floor(y / 8.0));
    if (j & 1) { y = y + 1.0; j += 1; }
    j &= 7;
    const double z = (x - y * 45.0) * PX_PI180, zz = z * z;
    const double sp = z + z * (zz * (-1.0 / 6.0 + zz * (1.0 / 120.0 + zz * (-1.0 / 5040.0 + zz * (1.0 / 362880.0 + zz * (-1.0 / 39916800.0 + zz * (
        1.0 / 6227020800.0 + zz * (-1.0 / 1307674368000.0))))))));
    const double cp = 1.0 - zz * (0.5 - zz * (1.0 / 24.0 - zz * (1.0 / 720.0 - zz * (1.0 / 40320.0 - zz * (1.0 / 3628800.0 - zz * (1.0 / 479001600.0 - zz * (
        1.0 / 87178291200.0 - zz * (1.0 / 20922789888000.0))))))));
    *cosv = j == 0 ? cp : (j == 2 ? -sp : (j == 4 ? -cp : sp));
    *sinv = j == 0 ? sp : (j == 2 ? cp : (j == 4 ? -sp : -cp));
}

static inline void px_weights(double cc, double *w) {
    const double y = cc - floor(cc), z = 1.0 - y;
    w[1] = ((y * y) * (y - 2.0) * 3.0 + 4.0) / 6.0;
    w[2] = ((z - 2.0) * (z * z) * 3.0 + 4.0) / 6.0;
    w[0] = ((z * z) * z) / 6.0;
    w[3] = ((1.0 - w[0]) - w[1]) - w[2];
}

static inline int px_mirror(int i) {
    i = i < 0 ? -i : i;
    return i >= PX_CROP ? 2 * PX_CROP - 2 - i : i;
}

static double px_bicubic(double x) {
    const double a = -0.5;
    if (x < 0.0) x = -x;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
    return 0.0;
}

/* Pillow's precompute_coeffs + normalize_coeffs_8bpc (src/libImaging/Resample.c) for 200 -> 64 pixels, bicubic. */
void oc_resize_coefficients(int32_t *kk /* [64][15] */, int32_t *bounds /* [64][2] */) {
    const double scale = (double)PX_WIN / PATCH, filterscale = scale, support = 2.0 * filterscale, ss = 1.0 / filterscale;
    for (int xx = 0; xx < PATCH; ++xx) {
        const double center = 0 + (xx + 0.5) * scale;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > PX_WIN) xmax = PX_WIN;
        xmax -= xmin;
        double k[PX_KSIZE], ww = 0.0;
        for (int x = 0; x < xmax; ++x) { k[x] = px_bicubic((x + xmin - center + 0.5) * ss); ww += k[x]; }
        for (int x = 0; x < PX_KSIZE; ++x) {
            double w = x < xmax ? k[x] : 0.0;
            if (x < xmax && ww != 0.0) w /= ww;
            kk[xx * PX_KSIZE + x] = w < 0 ? (int32_t)(-0.5 + w * (1 << PX_BITS)) : (int32_t)(0.5 + w * (1 << PX_BITS));
        }
        bounds[2 * xx] = xmin;
        bounds[2 * xx + 1] = xmax;
    }
}

void oc_patch_exact_range(const oc_track *t, const oc_frame *f, const oc_state *s, uint8_t *patch, int c0, int c1) {
    int32_t kk[PATCH * PX_KSIZE], bounds[PATCH * 2];
    oc_resize_coefficients(kk, bounds);
    double *coef = (double *)malloc(sizeof(double) * PX_CROP * PX_CROP);
    uint8_t *win = (uint8_t *)malloc(PX_WIN * PX_WIN), *tmp = (uint8_t *)malloc(PX_WIN * PATCH);
    for (int car = c0; car < c1; ++car) {
        uint8_t *out = patch + (size_t)car * PATCH * PATCH;
        if (s->fresh[car]) { memset(out, 0, PATCH * PATCH); continue; }
        const double x = (double)s->x[car], y = (double)s->y[car], yaw = (double)s->theta[car];
        const int pr = (int)((double)f->fh - (y - f->oy) / f->res), pc = (int)((x - f->ox) / f->res);
        /* 1. the crop, north-up, as float64 samples */
        for (int r = 0; r < PX_CROP; ++r)
            for (int c = 0; c < PX_CROP; ++c) {
                const int gy = f->r_top - (pr - PX_CROP / 2 + r), gx = (pc - PX_CROP / 2 + c) - f->c0;
                coef[r * PX_CROP + c] = (inb(t, gx, gy) && t->drv[(size_t)gy * t->w + gx]) ? 1.0 : 0.0;
            }
        /* 2. spline coefficients: along axis 0 (columns), then along axis 1 (rows) */
        for (int c = 0; c < PX_CROP; ++c) px_filter_line(coef + c, PX_CROP);
        for (int r = 0; r < PX_CROP; ++r) px_filter_line(coef + (size_t)r * PX_CROP, 1);
        /* 3. the rotation */
        double cs, sn;
        px_sincos_deg((2.0 * 3.141592653589793 - yaw) * (180.0 / 3.141592653589793), &cs, &sn);
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
        /* 4. the centre window of the rotated image */
        for (int i = 0; i < PX_WIN; ++i)
            for (int j = 0; j < PX_WIN; ++j) {
                const double o0 = (double)(S0 / 2 - PX_WIN / 2 + i), o1 = (double)(S1 / 2 - PX_WIN / 2 + j);
                const double cc0 = ((0.0 + o0 * cs) + o1 * sn) + off0, cc1 = ((0.0 + o0 * (-sn)) + o1 * cs) + off1;
                double tv = 0.0;
                if (!(cc0 < 0 || cc0 > PX_CROP - 1 || cc1 < 0 || cc1 > PX_CROP - 1)) {
                    double w0[4], w1[4];
                    px_weights(cc0, w0);
                    px_weights(cc1, w1);
                    const int st0 = (int)floor(cc0) - 1, st1 = (int)floor(cc1) - 1;
                    for (int a = 0; a < 4; ++a) {
                        const double *row = coef + (size_t)px_mirror(st0 + a) * PX_CROP;
                        for (int b = 0; b < 4; ++b) tv = tv + (row[px_mirror(st1 + b)] * w0[a]) * w1[b];
                    }
                }
                tv = tv > 0 ? tv + 0.5 : 0.0;
                win[i * PX_WIN + j] = (uint8_t)(tv > 255.0 ? 255.0 : tv);
            }
        /* 5. Pillow's 8-bit resize: horizontal pass, then vertical */
        for (int r = 0; r < PX_WIN; ++r)
            for (int xx = 0; xx < PATCH; ++xx) {
                int32_t acc = 1 << (PX_BITS - 1);
                for (int k = 0; k < bounds[2 * xx + 1]; ++k) acc += (int32_t)win[r * PX_WIN + bounds[2 * xx] + k] * kk[xx * PX_KSIZE + k];
                acc >>= PX_BITS;
                tmp[r * PATCH + xx] = (uint8_t)(acc < 0 ? 0 : (acc > 255 ? 255 : acc));
            }
        for (int yy = 0; yy < PATCH; ++yy)
            for (int xx = 0; xx < PATCH; ++xx) {
                int32_t acc = 1 << (PX_BITS - 1);
                for (int k = 0; k < bounds[2 * yy + 1]; ++k) acc += (int32_t)tmp[(bounds[2 * yy] + k) * PATCH + xx] * kk[yy * PX_KSIZE + k];
                acc >>= PX_BITS;
                out[yy * PATCH + xx] = (uint8_t)(acc < 0 ? 0 : (acc > 255 ? 255 : acc));
            }
    }
    free(coef); free(win); free(tmp);
}

/* CPU-share calibration for bench.py's cpu_baseline leg (oracle/cpu_baseline.py): a fixed amount of dependent integer
 * work.  One call on one thread against T concurrent calls on T threads tells how many cores the process really gets
 * (a container's CPU quota is not visible through sched_getaffinity). */
uint64_t oc_spin(uint64_t iterations) {
    uint64_t x = 0x9e3779b97f4a7c15ull;
    for (uint64_t i = 0; i < iterations; ++i) x = x * 6364136223846793005ull + 1442695040888963407ull + (x >> 29);
    return x;
}
