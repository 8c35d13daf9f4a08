/* Plain-C client of the racecar_hip C-ABI: no Python, no torch, only include/racecar_hip.h.
 *
 *   gcc -O2 -Iinclude examples/c_rollout.c -o c_rollout -Lracing_dreamer_amd/lib -lracecar_hip \
 *       -Wl,-rpath,$PWD/racing_dreamer_amd/lib -lm
 *   ./c_rollout [num_envs] [steps]
 *
 * Builds a small synthetic circuit on the host (a rectangular corridor loop), uploads it, and drives
 * `num_envs` cars with the device-side follow-the-gap agent for `steps` agent steps of 4 sub-steps
 * (dreamer/dream.py:55,211-216).  What a maintainer binding the library from another language would write. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "racecar_hip.h"

#define CHECK(call)                                                              \
    do {                                                                         \
        int _rc = (call);                                                        \
        if (_rc != RC_OK) {                                                      \
            fprintf(stderr, "%s failed (%d): %s\n", #call, _rc, rc_last_error()); \
            return 1;                                                            \
        }                                                                        \
    } while (0)

enum { H = 240, W = 320, PITCH = (W + 31) / 32 };     /* cells of 0.05 m: a 16 m x 12 m map            */
enum { X0 = 30, X1 = 290, Y0 = 30, Y1 = 210, LANE = 40 }; /* corridor: outer box minus inner box, 2 m wide  */

static int drivable(int ix, int iy) {
    const int outer = ix >= X0 && ix < X1 && iy >= Y0 && iy < Y1;
    const int inner = ix >= X0 + LANE && ix < X1 - LANE && iy >= Y0 + LANE && iy < Y1 - LANE;
    return outer && !inner;
}

/* centre line of the corridor, counter-clockwise from the middle of the bottom straight; s in [0, 1) */
static void centre(float s, float *x, float *y, float *heading) {
    const float cx0 = X0 + LANE / 2.0f, cx1 = X1 - LANE / 2.0f, cy0 = Y0 + LANE / 2.0f, cy1 = Y1 - LANE / 2.0f;
    const float lw = cx1 - cx0, lh = cy1 - cy0, total = 2 * (lw + lh);
    float d = s * total + lw / 2;                      /* start in the middle of the bottom straight */
    d = fmodf(d, total);
    if (d < lw) { *x = cx0 + d; *y = cy0; *heading = 0.0f; }
    else if (d < lw + lh) { *x = cx1; *y = cy0 + (d - lw); *heading = 1.5707964f; }
    else if (d < 2 * lw + lh) { *x = cx1 - (d - lw - lh); *y = cy1; *heading = 3.1415927f; }
    else { *x = cx0; *y = cy1 - (d - 2 * lw - lh); *heading = -1.5707964f; }
}

int main(int argc, char **argv) {
    const int num_envs = argc > 1 ? atoi(argv[1]) : 1024;
    const int steps = argc > 2 ? atoi(argv[2]) : 200;
    const float res = 0.05f, ox = -8.0f, oy = -6.0f;

    uint32_t *occ = calloc((size_t)H * PITCH, 4), *drv = calloc((size_t)H * PITCH, 4);
    float *progress = malloc(sizeof(float) * H * W);
    enum { NCL = 2048 };
    float *cl = malloc(sizeof(float) * NCL * 4);
    for (int k = 0; k < NCL; ++k) {
        float x, y, th;
        centre((float)k / NCL, &x, &y, &th);
        cl[4 * k] = ox + x * res; cl[4 * k + 1] = oy + y * res; cl[4 * k + 2] = th; cl[4 * k + 3] = (float)k / NCL;
    }
    for (int iy = 0; iy < H; ++iy)
        for (int ix = 0; ix < W; ++ix) {
            const int d = drivable(ix, iy);
            if (d) drv[iy * PITCH + ix / 32] |= 1u << (ix % 32);
            else occ[iy * PITCH + ix / 32] |= 1u << (ix % 32);
            float best = 1e30f, p = -1.0f;
            if (d)                                     /* progress of the nearest centre-line point */
                for (int k = 0; k < NCL; k += 4) {
                    const float dx = ox + (ix + 0.5f) * res - cl[4 * k], dy = oy + (iy + 0.5f) * res - cl[4 * k + 1];
                    const float q = dx * dx + dy * dy;
                    if (q < best) { best = q; p = cl[4 * k + 3]; }
                }
            progress[iy * W + ix] = p;
        }

    rc_config cfg;
    rc_default_config(&cfg);
    cfg.num_envs = num_envs;
    cfg.cars_per_env = 1;
    cfg.auto_reset = 1;
    printf("racecar_hip ABI %d, %d envs, arena %.1f MB\n", rc_abi_version(), num_envs, rc_arena_bytes(&cfg) / 1e6);
    rc_env *env = NULL;
    CHECK(rc_create(&cfg, &env));
    CHECK(rc_load_track(env, occ, drv, progress, H, W, PITCH, res, ox, oy, cl, NCL));
    if (rc_step(env, NULL, 1) != RC_ERR_NEEDS_RESET) { fprintf(stderr, "expected RC_ERR_NEEDS_RESET\n"); return 1; }
    printf("step before reset: \"%s\"\n", rc_last_error());
    CHECK(rc_reset(env, NULL, RC_RESET_RANDOM, 42));

    float *lidar = malloc(sizeof(float) * (size_t)num_envs * 1080), *reward = malloc(sizeof(float) * num_envs);
    float *ptot = malloc(sizeof(float) * num_envs);
    uint8_t *done = malloc(num_envs);
    double total_reward = 0.0;
    long episodes = 0;
    for (int t = 0; t < steps; ++t) {
        CHECK(rc_follow_the_gap(env, 0.6f, 0.3f));     /* actions from the current scans, on the device */
        CHECK(rc_step(env, NULL, 4));                  /* ActionRepeat(4) */
        CHECK(rc_copy_out(env, RC_F_REWARD, reward, sizeof(float) * num_envs));
        CHECK(rc_copy_out(env, RC_F_DONE, done, (size_t)num_envs));
        for (int e = 0; e < num_envs; ++e) { total_reward += reward[e]; episodes += done[e]; }
    }
    CHECK(rc_copy_out(env, RC_F_LIDAR, lidar, sizeof(float) * (size_t)num_envs * 1080));
    CHECK(rc_copy_out(env, RC_F_PROGRESS_TOTAL, ptot, sizeof(float) * num_envs));
    float lo = 1e9f, hi = -1e9f, pmax = -1e9f;
    for (size_t i = 0; i < (size_t)num_envs * 1080; ++i) { lo = fminf(lo, lidar[i]); hi = fmaxf(hi, lidar[i]); }
    for (int e = 0; e < num_envs; ++e) pmax = fmaxf(pmax, ptot[e]);
    printf("%d agent steps: mean reward per env %.3f, finished episodes %ld, lidar range [%.3f, %.3f] m, "
           "best lap+progress %.3f\n", steps, total_reward / num_envs, episodes, lo, hi, pmax);
    int ok = lo >= 0.0f && hi <= 15.0f && total_reward > 0.0;

    /* The multi-GPU leg from plain C (third argument "gather"): this process is rank 0 of a communicator of one (a job
     * of N processes hands the 128-byte id of rank 0 to the others by file, socket or MPI).  The scan stores the LiDAR
     * row a second time as uint16 into a caller-owned slab; the all-gather sends that half-size record. */
    if (argc > 3 && strcmp(argv[3], "gather") == 0) {
    char id[128];
    void *slab = NULL, *gathered = NULL;
    const size_t cbytes = rc_compact_bytes(&cfg);
    CHECK(rc_comm_unique_id(id, sizeof id));
    CHECK(rc_comm_init(env, id, sizeof id, 0, 1));
    CHECK(rc_device_alloc(env, cbytes, &slab));
    CHECK(rc_device_alloc(env, cbytes, &gathered));
    CHECK(rc_set_compact_slab(env, slab, cbytes));
    CHECK(rc_follow_the_gap(env, 0.6f, 0.3f));
    CHECK(rc_step(env, NULL, 4));
    CHECK(rc_gather_trajectory(env, RC_GATHER_FULL_U16, gathered, cbytes));
    CHECK(rc_gather_wait(env, 1));
    uint16_t q[1080];
    CHECK(rc_copy_from_device(env, gathered, q, sizeof q));                   /* rank 0's record, car 0, uint16 LiDAR */
    CHECK(rc_copy_out(env, RC_F_LIDAR, lidar, sizeof(float) * (size_t)num_envs * 1080));
    int q_ok = 1;
    for (int i = 0; i < 1080; ++i) q_ok &= q[i] == (uint16_t)lrintf(lidar[i] * 4369.0f);   /* rne(v * 65535 / 15) */
    printf("gathered %zu bytes per rank (fp32 record: %zu); uint16 scan of car 0 %s the fp32 one\n", rc_gather_bytes(env, RC_GATHER_FULL_U16),
           rc_gather_bytes(env, RC_GATHER_FULL), q_ok ? "matches" : "DIFFERS FROM");
    ok = ok && q_ok;
    CHECK(rc_set_compact_slab(env, NULL, 0));
    CHECK(rc_device_free(env, slab));
    CHECK(rc_device_free(env, gathered));
    }
    rc_destroy(env);
    free(occ); free(drv); free(progress); free(cl); free(lidar); free(reward); free(ptot); free(done);
    printf(ok ? "OK\n" : "FAILED\n");
    return ok ? 0 : 1;
}
