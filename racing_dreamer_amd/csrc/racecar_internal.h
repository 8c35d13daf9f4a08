// Internal structures shared by the C-ABI layer (racecar_abi.hip) and the kernels
// (racecar_kernels.hip).  Not part of the public interface.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define RC_NSTEP_MAX 16                        // longest n_step_progress window (sub-steps)
#ifndef RC_STAMP_SLOTS
#define RC_STAMP_SLOTS 32                       // uint64 per wave of the instrumented scan (include/racecar_hip.h)
#endif
#define RC_FIRST_BINS 64                       // bins of |dy / dx|: eight per octave over 2^-4 .. 2^4 (outer bins open-ended)
#define RC_FIRST_SHIFT 20                      // slope bits >> 20 = (exponent << 3) | three mantissa bits
#define RC_FIRST_BIAS (123u << 3)              // ... of 2^-4
#define RC_FIRST_PLANES (4 * RC_FIRST_BINS)     // x 2 bytes = 512 bytes per cell.  (32 bins while every round fetched its entries from
                                               // global memory - 64 were 2 % slower then; with the line staged in LDS once per car 64 bins
                                               // are 3 % FASTER: 2.62 instead of 2.76 wave-level trips per round)

struct RcTrackDev {
    const uint32_t *ray_words;   // occupancy | sentinel ring, [h][pitch]
    const uint32_t *drv_words;   // drivable area, [h][pitch]
    const float *progress;       // [h][w], < 0 outside the drivable area
    const float *centerline;     // [n_centerline][4] = x, y, heading, progress
    const float4 *spawn;         // [n_centerline][2] = (x, y, heading, cos), (sin, progress at that cell, checkpoint | safe bin << 8 as int bits, lateral room):
                                 // everything a reset needs from a spawn index in one 32-byte gather (built on the device)
    const float *beams;          // [1080][2] = cos, sin of the beam angle in the sensor frame
    const float *footprint;      // [34][2] body-frame perimeter points
    const uint8_t *free_blocks;  // [blk_h][blk_w]: per (1<<blk_shift)^2-cell block, min over its cells of the
                                 // Chebyshev distance to the nearest occupied/ring cell (0 = block not free)
    const uint8_t *cell_dist;    // [h][cell_pitch]: per cell, chessboard distance to the nearest stop cell (0 = stop, capped 255)
    int32_t cell_pitch;
    const uint16_t *quad_rect;   // [4][h][cell_pitch]: per direction quadrant q = (dy < 0) * 2 + (dx < 0) and cell, a free
                                 // rectangle with that cell at its corner, extending towards the quadrant:
                                 // width | height << 8 in cells (1..255 each), 0 = wall, 0x0100 = sentinel ring.
                                 // Plane q is stored MIRRORED: columns reversed when q & 1, rows when q & 2, so that
                                 // every ray walks its plane towards increasing addresses (racecar_kernels.hip)
    int32_t quad_plane_bytes;    // bytes per quadrant plane
    const uint16_t *first_rect;  // [h][cell_pitch][RC_FIRST_PLANES]: first-trip rectangles by quadrant and slope bin (variant 7)
    const uint32_t *packed_blocks; // [blk_h][blk_w] for 4x4 blocks: bits 0-15 occupancy of the block's cells
                                 // (bit (iy&3)*4 + (ix&3), sentinel ring included), bits 16-23 the value above
    int32_t blk_w, blk_h, blk_shift, blk_bytes, packed_bytes, packed_w;   // packed_w: uint32 per packed row
    int32_t h, w, pitch, n_centerline;
    float org_x, org_y, res, inv_res, tmax;
    float band, band_mh, band2;  // scan variants 6/7: half-width of the zone around a cell boundary in which the other-axis cell
                                 // is counted exactly = (max(w, h) + 2) * 2^-21 cells; band - 0.5; 2 * band
};

struct RcStateDev {              // persistent per-car / per-env simulator state (SoA)
    float *x, *y, *theta, *ct, *st, *v, *delta, *omega, *accel, *progress;
    int32_t *lap, *cp;
    uint8_t *wall, *opp, *wrong, *done, *trunc, *fresh;
    int32_t *steps, *agent_steps;   // per env
    uint32_t *episode;              // per env
    float4 *scan_pose;              // [n_cars] (x, y, cos, sin) once more, packed: the scan fetches a car's state with ONE
                                    // scalar 16-byte load (four separate words cost it a second serial round trip)
    int4 *patch_pose;               // [n_cars] what the lidar_occupancy render needs of a car, as integers: (start cell x, start cell
                                    // y + 1, the pixel step a, b in 16.16) or x = RC_PATCH_SKIP for an all-zero patch (first
                                    // observation of an episode, diverged pose) - written next to scan_pose when the render is on
    float *nstep_hist;              // [n_cars][RC_NSTEP_MAX] total progress at sub-step s in slot s % n_steps; null unless
                                    // some car runs RC_TASK_N_STEP_PROGRESS
    const int32_t *order;           // [n_cars] the order in which the scan's waves take the cars: sorted by track position every
                                    // RC_ORDER_PERIOD observations, so that the waves in flight at one time read one stretch of the
                                    // track's tables (L2); null = car index order (small batches)
};

struct RcOutDev {                // output arena sections (see rc_field)
    float *lidar, *pose, *velocity, *speed, *action, *reward, *discount, *progress_total, *time;
    uint8_t *patch;
    float *progress;
    int32_t *lap, *cp;
    uint8_t *done, *trunc, *wall, *opp, *wrong, *fresh;
    float *accel, *steer;
    uint16_t *lidar_u16;         // optional [n][1080] uint16 copy of the LiDAR row (rc_set_compact_slab), else null
};

struct RcParams {
    RcTrackDev trk;
    RcStateDev st;
    RcOutDev out;
    int32_t num_envs, cars_per_env, n_cars;
    uint32_t first_env;
    int32_t task, laps, terminate_on_collision, remap_actions, time_limit_steps, auto_reset, render_patch;
    int32_t lidar_transform;
    float time_limit, collision_reward;
    float act_lo0, act_lo1, act_hi0, act_hi1;
    int32_t reset_mode;
    uint32_t seed_lo, seed_hi;
    int32_t car_task[4];         // task per car slot (resolved: never -1)
    int32_t n_steps;             // window of RC_TASK_N_STEP_PROGRESS [sub-steps]
    uint32_t *scan_overrun;      // device counter: waves of the BOUNDED scan build that used up a round's trip budget
};

#define RC_PATCH_SKIP 0x7fffffff
// The render's bitmap in LDS with a border of RC_PATCH_PAD zero cells on every side (a tap lies at most 110 sqrt 2 = 155.6 cells
// from the car's cell, + 1 for the row offset of the start cell): a car that stands inside the grid then needs no clamp on any
// tap.  Used when two such images fit one CU's 160 KB (columbia 49 KB, treitlstrasse_v2 56 KB, austria 79 KB; barcelona's
// 169 KB does not: that track keeps the unpadded bitmap and clamps).  Bytes of the padded image, 0 if it is not used.
#define RC_PATCH_PAD 160
static inline size_t rc_patch_padded_bytes(int h, int w) {
    const size_t pitch_words = ((size_t)w + 2 * RC_PATCH_PAD + 31) / 32, rows = (size_t)h + 2 * RC_PATCH_PAD;
    const size_t bytes = (pitch_words * 4 * rows + 15) / 16 * 16;
    return bytes <= 80 * 1024 ? bytes : 0;
}
#define RC_ORDER_BUCKETS 1024       // counting sort of the cars by progress
#define RC_ORDER_REGION 256u         // ranks per region handed to one XCD (rc_order_place_kernel)
#define RC_ORDER_PERIOD 64          // observations between two sorts (cars move centimetres per step)
#define RC_ORDER_MIN_CARS 16384     // track order from here on; below, the whole batch is in flight at once anyway, and on maps whose
#define RC_ORDER_COST_MIN_CARS 1024  // quadrant planes exceed RC_ORDER_COST_MIN_TABLE bytes (columbia: 0.9 MB, no gain; austria: 1.8 MB) the cars are taken
#define RC_ORDER_COST_MIN_TABLE (5u << 18)   // longest first instead (rc_order_cost_key_kernel); 1.25 MB: measured, see EXPERIMENTS I.11
#define RC_GROUP_MAX 8
struct RcGroup {                 // several handles in one launch (rc_step_group)
    const RcParams *params;      // device table, one entry per block
    float *actions[RC_GROUP_MAX];          // dynamics: the block's action buffer
    int32_t wave_start[RC_GROUP_MAX + 1];  // first wave of every block in the launch's grid, and the total
    int32_t n;
};

struct RcLaunchInfo {            // per-handle launch geometry decided at rc_load_track
    int32_t n_cu;
    int32_t ray_blocks, ray_threads;
    int32_t car_threads;         // workgroup size of the one-wave-per-car scan (variant 7): 64 = one wave per workgroup
    int32_t car_split;           // waves sharing one car's 17 rounds of 64 beams (1 for large batches)
    int32_t scan_guarded;        // 1: the scan runs the build whose trip loop is bounded (validation band, RC_DBG_SCAN_BOUNDED)
    unsigned long long *scan_stamps;   // rc_debug_scan_stamps: device buffer of the instrumented scan, else null
    int32_t scan_stamp_waves;
    int32_t patch_variant;       // experiment bits of the lidar_occupancy render (rc_debug_set): 2 = plain instead of non-temporal stores
    size_t lds_bytes;            // occupancy bitmap (also the patch kernel's drivable bitmap)
    size_t lds_bytes_skip;       // bitmap + free-block table (raycast variants 1, 2); 0 if it does not fit
    size_t lds_bytes_packed;     // packed block table only (raycast variant 3); 0 if it does not fit / blocks are 8x8
    int32_t raycast_variant;     // 0 plain, 1 skipping, 2 skipping tuned, 3 tuned + packed table in LDS,
                                 // 4 packed table read from global memory, 5 per-cell distance table from global memory,
                                 // 6 per-cell, per-quadrant free rectangles from global memory, 7 the same with one wave per car
                                 // (identical results)
};

#define RC_GATHER_MAX_FIELDS 24
struct RcSampleWindows {
    const unsigned char *ring;
    size_t slot_bytes, fresh_off, done_off;
    int32_t capacity, oldest, n_start, length, n_windows, n_cars, max_tries;
    uint32_t seed_lo, seed_hi, draw;
    int32_t *slot_idx, *slot_obs_idx, *car_idx, *meta;
    uint32_t *failed;
};
hipError_t rck_sample_windows(const RcSampleWindows &a, hipStream_t s);
hipError_t rck_cost_keys(const float *lidar_dev, int n_cars, float *key_dev, hipStream_t s);
hipError_t rck_sort_cars(const float *progress_dev, int n_cars, uint32_t *counts_dev, int32_t *order_dev, hipStream_t s);
struct RcRandomActions;
hipError_t rck_launch_dynamics_group(const RcGroup &g, int cars_per_env, int repeat, const RcRandomActions &ra, hipStream_t s);
hipError_t rck_launch_raycast_group(const RcGroup &g, int cars_per_env, int split, hipStream_t s);
struct RcBatchRows {             // rc_sample_batch: what turns the row gather into the gather of a whole training batch
    uint32_t obs_mask, reset_mask;                   // by position in the gather's field table
    uint32_t reset_word[RC_GATHER_MAX_FIELDS];
    const int32_t *slot_obs_idx, *meta;
    int32_t length;
};
hipError_t rck_gather_rows(const void *ring, size_t slot_bytes, const int32_t *slot_idx, const int32_t *car_idx, int n_rows,
                           const size_t *src_off, const size_t *dst_off, const uint32_t *bpc, int n_fields, void *out, hipStream_t s,
                           const RcBatchRows *batch = nullptr);
#define RC_P2P_MAX_RANKS 64
#define RC_P2P_TIMEOUT_S 20.0                   // bound of a flag poll (a peer that never posts: an error, not a hung queue)
struct RcP2pPost {               // one store per lane: flag[p] = value (null entries skipped)
    uint32_t *flag[RC_P2P_MAX_RANKS];
    uint32_t value;
    int32_t n;
};
hipError_t rck_p2p_post(const RcP2pPost &post, hipStream_t s);
hipError_t rck_p2p_wait(const uint32_t *flags, int n, int skip, uint32_t value, uint32_t *timeouts, double limit_s, hipStream_t s);

// kernel launchers (racecar_kernels.hip); all asynchronous on `s`
void rck_set_launch_events(hipEvent_t start, hipEvent_t stop);   // attach start / stop timestamps to the NEXT launch of this thread
hipError_t rck_set_lds_limits(size_t lds_bytes);
const char *rck_lab_abi_string();   // sizes of RcParams / RcLaunchInfo + the hash of the headers: what a lab library must have been built against
const char *rck_lab_unavailable();   // nullptr if the lab library (scan variants 0-6, stamps build: racecar_lab.hip) can be used, else why not
hipError_t rck_build_quad_planes(const RcTrackDev &t, uint16_t *quad_rect_dev, hipStream_t s);   // needs ray_words, h, w, pitch, cell_pitch, quad_plane_bytes
hipError_t rck_build_first_table(const RcTrackDev &t, uint16_t *first_rect_dev, hipStream_t s);   // needs ray_words, h, w, pitch, cell_pitch
hipError_t rck_validate_tables(const RcTrackDev &t, float band, hipStream_t s, unsigned long long *n_scans, unsigned *n_overruns);   // bounded scan from every free cell
hipError_t rck_build_spawn_table(const RcTrackDev &t, float4 *spawn_dev, hipStream_t s);   // needs centerline, progress, geometry
struct RcRandomActions { int32_t on; uint32_t seed_lo, seed_hi, step; };   // on != 0: draw the actions in the dynamics kernel
hipError_t rck_launch_dynamics(const RcParams &p, float *actions, int repeat, const RcRandomActions &ra, hipStream_t s);
hipError_t rck_launch_reset(const RcParams &p, const uint8_t *mask_dev, hipStream_t s);
hipError_t rck_launch_set_pose(const RcParams &p, const float *xyyaw_dev, hipStream_t s);
hipError_t rck_launch_raycast(const RcParams &p, const RcLaunchInfo &li, hipStream_t s);
hipError_t rck_launch_patch(const RcParams &p, const RcLaunchInfo &li, hipStream_t s);
#define RC_EXACT_CAR_DOUBLES (220 * 220)         // scratch per car of a chunk: the 220 x 220 binary64 spline coefficients (387 200 B)
#ifndef RC_EXACT_CHUNK_CARS
#define RC_EXACT_CHUNK_CARS 6144                 // cars per chunk of the exact render (2.4 GB of scratch)
#endif
#define RC_EXACT_TABLE_INTS (64 * 15 + 64 * 2)   // Pillow's integer coefficients [64][15] and bounds [64][2]
struct RcExactParams {
    const uint32_t *drv_words;   // drivable bitmap [h][pitch], ring cleared
    int32_t pitch, h, w;
    const float *x, *y, *theta;  // car state
    const uint8_t *fresh;
    int32_t fh, r_top, c0;       // source image height; north-up pixel (R, C) = cell (C - c0, r_top - R)   (rc_set_source_frame)
    double ox, oy, res;
    double *scratch;             // [chunk][220 * 220]
    uint8_t *patch;              // [n_cars][64][64]
    const int32_t *kk;           // [64][15] Pillow's integer coefficients, [64][2] bounds behind them
    int32_t car0, n_cars;
    unsigned long long *check;   // rc_selftest_exact_estimate: 4 counters (racecar_patch_exact.h, CHECK); nullptr in production
};
hipError_t rck_launch_patch_exact(const RcExactParams &p, int chunk_cars, hipStream_t s);   // obs_type lidar_occupancy_reference (racecar_patch_exact.h)
hipError_t rck_launch_ftg(const RcParams &p, float *actions, float motor_straight, float motor_corner, hipStream_t s);
hipError_t rck_launch_ftg_reference(const RcParams &p, float *actions, float *prev_heading, float dt, float *detail, hipStream_t s);
hipError_t rck_launch_selftest_rcp(uint32_t exp_lo, uint32_t exp_hi, unsigned long long *mismatches_dev, hipStream_t s);
hipError_t rck_launch_selftest_sqrt(uint32_t lo_bits, uint32_t hi_bits, unsigned long long *mismatches_dev, hipStream_t s);
hipError_t rck_launch_selftest_div6(int blocks, int threads, int per_lane, unsigned long long *mismatches_dev, hipStream_t s);
hipError_t rck_launch_random_actions(float *actions, int n_cars, uint32_t first_car, uint32_t seed_lo,
                                     uint32_t seed_hi, uint32_t step, hipStream_t s);
