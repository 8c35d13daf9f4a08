/*
 * racecar_hip.h - C ABI of libracecar_hip.so: the MI355X (gfx950) batched F1TENTH racing
 * environment.  Plain C, plain pointers and sizes; no torch / C++ types cross this boundary.
 *
 * What this replaces in the reference (CPS-TUWien/racing_dreamer).  The reference has no
 * native code; its "FFI" for this path is the Python call into the external simulator
 * `racecar_gym` (PyBullet).  Each entry point below cites the reference call site whose
 * role it takes over:
 *
 *   rc_create / rc_load_track ... MultiAgentScenario.from_spec + MultiAgentRaceEnv(scenario)
 *                                 dreamer/wrappers.py:14-15;  SingleAgentScenario.from_spec +
 *                                 ChangingTrackSingleAgentRaceEnv(...)
 *                                 baselines/racing/experiments/sb3/sb_experiment.py:61-63
 *   rc_reset .................... env.reset(mode='grid'|'random'|'random_ball')
 *                                 dreamer/wrappers.py:72,91-92; baselines/racing/environment/common.py:28-29
 *   rc_step ..................... env.step({'A': {'motor', 'steering'}})   dreamer/wrappers.py:62-64
 *                                 with ActionRepeat (wrappers.py:107-116), ReduceActionSpace
 *                                 (wrappers.py:128-134) and TimeLimit (wrappers.py:147-154) folded in
 *   rc_get / rc_copy_out ........ the obs / reward / done / info dicts returned by step()
 *                                 dreamer/wrappers.py:64-69,210-226
 *   RC_F_OCCUPANCY .............. OccupancyMapObs.step                     dreamer/wrappers.py:390-408
 *   rc_trajectory_slab .......... the per-step transition Collect.step records
 *                                 dreamer/wrappers.py:213-219 (+ dreamer/callbacks.py:41-53)
 *   rc_gather_trajectory ........ the concatenation of those records over all envs, which in the reference is
 *                                 one process appending to one episode list (dreamer/wrappers.py:220-226,
 *                                 dreamer/tools.py:235-264 reads them back); here the envs live on several GPUs
 *                                 and the concat is an RCCL all-gather
 *
 * Conventions
 *   - every function returns RC_OK (0) or a negative rc_status; the message of the last
 *     failure on the calling thread is rc_last_error().  No exception crosses the ABI.
 *   - one rc_env = one GPU + one HIP stream.  Calls on one handle must be serialised by the
 *     caller (the reference callers are single threaded); different handles are independent and
 *     may be driven from different host threads: every entry point selects the handle's device
 *     itself (hipSetDevice), whatever the calling thread's current device is.
 *   - the library owns all device buffers for the handle's lifetime unless the caller passes
 *     `external_arena`; rc_get() returns borrowed device pointers.
 *   - all work is stream-ordered on the handle's stream; rc_sync() waits for it.
 *   - car index c = env * cars_per_env + agent.  Arrays are SoA, one section per field.
 */
#ifndef RACECAR_HIP_H
#define RACECAR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RC_ABI_VERSION 3      /* 3 (round 6): RC_OBS_LIDAR_OCCUPANCY_REFERENCE + rc_set_source_frame; starts never touch a wall on narrow maps
                               * 2 (round 4): rc_build_id, reset laws of SURVEY H6, outbound ordering of the peer-copy gather */
#define RC_N_BEAMS 1080
#define RC_PATCH 64
#define RC_MAX_CARS 4

typedef struct rc_env rc_env;

typedef enum rc_status {
    RC_OK = 0,
    RC_ERR_INVALID = -1,      /* bad argument / call order            */
    RC_ERR_HIP = -2,          /* a HIP runtime call failed            */
    RC_ERR_NO_TRACK = -3,     /* rc_load_track has not been called    */
    RC_ERR_NEEDS_RESET = -4,  /* step before reset ("Must reset environment.", wrappers.py:148) */
    RC_ERR_NOMEM = -5,
    RC_ERR_COMM = -6          /* RCCL could not be loaded or a collective call failed */
} rc_status;

enum { RC_TASK_MAX_PROGRESS = 0, RC_TASK_MAX_SPEED = 1,             /* scenario yml task_name; tasks.py:4-22 */
       RC_TASK_N_STEP_PROGRESS = 2 };  /* secondary agents, baselines/scenarios/max_progress/columbia.yml:17-18: reward =
                                          100 x total progress gained over the last n_steps sub-steps; never done */
#define RC_NSTEP_MAX 16
enum { RC_RESET_GRID = 0, RC_RESET_RANDOM = 1, RC_RESET_RANDOM_BALL = 2 };  /* dream.py:105-108,120 */
enum { RC_OBS_LIDAR = 0, RC_OBS_LIDAR_OCCUPANCY = 1,                  /* dream.py obs_type */
       RC_OBS_LIDAR_OCCUPANCY_REFERENCE = 2 };   /* lidar_occupancy computed EXACTLY as the reference's OccupancyMapObs.step does
                                                  * (dreamer/wrappers.py:396-406: to_pixel, 220 x 220 crop, cubic-spline rotation,
                                                  * centre crop, antialiased bicubic resize) instead of by the one-tap sampler of
                                                  * RC_OBS_LIDAR_OCCUPANCY: bit-identical to the reference's patches, ~500 x the render's cost (0.5 us per car);
                                                  * needs rc_set_source_frame */
/* what RC_F_LIDAR holds: metres, or the caller-side scaling fused into the scan's store */
enum { RC_LIDAR_METRES = 0,
       RC_LIDAR_DREAMER = 1,      /* range / 15 - 0.5              tools.preprocess, dreamer/tools.py:274          */
       RC_LIDAR_UNIT = 2 };       /* (range - 0) * (1 / (15 - 0))  NormalizeObservations, single_agent.py:92-99   */

/* Output / state fields, for rc_get() and rc_copy_out().  n = num_envs * cars_per_env. */
typedef enum rc_field {
    /* --- trajectory record, contiguous in this order (rc_trajectory_slab) --- */
    RC_F_LIDAR = 0,          /* float32 [n, 1080]  ranges [m], beam 0 at +135 deg, clockwise  */
    RC_F_POSE = 1,           /* float32 [n, 6]     x, y, 0, 0, 0, yaw                          */
    RC_F_VELOCITY = 2,       /* float32 [n, 6]     v, 0, 0, 0, 0, yaw rate                     */
    RC_F_SPEED = 3,          /* float32 [n]        |v|   (RaceCarWrapper.step, wrappers.py:66) */
    RC_F_ACTION = 4,         /* float32 [n, 2]     the action passed to rc_step                */
    RC_F_REWARD = 5,         /* float32 [n]        summed over the repeated sub-steps          */
    RC_F_DISCOUNT = 6,       /* float32 [n]        1 - done               (wrappers.py:217)    */
    RC_F_PROGRESS_TOTAL = 7, /* float32 [n]        lap + progress - 1     (wrappers.py:218)    */
    RC_F_TIME = 8,           /* float32 [n]        simulated seconds      (wrappers.py:219)    */
    RC_F_OCCUPANCY = 9,      /* uint8   [n, 64, 64] lidar_occupancy, 1 = drivable (only if enabled) */
    /* --- info / flags --- */
    RC_F_PROGRESS = 10,      /* float32 [n]  norm. distance from start in [0, 1]               */
    RC_F_LAP = 11,           /* int32   [n]  first lap is 1                                    */
    RC_F_CHECKPOINT = 12,    /* int32   [n]                                                    */
    RC_F_DONE = 13,          /* uint8   [n]                                                    */
    RC_F_TRUNCATED = 14,     /* uint8   [n]  done because of time_limit_steps                  */
    RC_F_WALL_COLLISION = 15,     /* uint8 [n]                                                 */
    RC_F_OPPONENT_COLLISION = 16, /* uint8 [n]                                                 */
    RC_F_WRONG_WAY = 17,     /* uint8   [n]                                                    */
    RC_F_FRESH = 18,         /* uint8   [n]  1 if the observation is the first of a new episode */
    RC_F_ACCELERATION = 19,  /* float32 [n]  longitudinal acceleration                          */
    RC_F_STEERING_ANGLE = 20,/* float32 [n]  front wheel angle [rad]                            */
    RC_F_ACTION_IN = 21,     /* float32 [n, 2] device-side action input buffer (writable)       */
    RC_F_COUNT = 22
} rc_field;

/* Kernels, for rc_kernel_time(). */
enum { RC_K_DYNAMICS = 0, RC_K_RAYCAST = 1, RC_K_PATCH = 2, RC_K_RESET = 3, RC_K_ACTIONS = 4, RC_K_FTG = 5, RC_K_COUNT = 6 };

typedef struct rc_config {
    uint32_t struct_size;          /* = sizeof(rc_config), for ABI evolution                   */
    int32_t  device;               /* HIP device ordinal                                       */
    int32_t  num_envs;             /* envs held by this handle (this GPU's shard)              */
    int32_t  cars_per_env;         /* 1..RC_MAX_CARS                                           */
    int64_t  first_env;            /* global index of env 0: RNG streams are keyed by the global
                                      env id, so results do not depend on the sharding         */
    int32_t  obs_type;             /* RC_OBS_LIDAR | RC_OBS_LIDAR_OCCUPANCY                    */
    int32_t  task;                 /* RC_TASK_*                                                */
    int32_t  laps;                 /* scenario params, dreamer/scenarios/max_progress/columbia.yml:10 */
    float    time_limit;           /*   seconds of simulated time                              */
    int32_t  terminate_on_collision;
    float    collision_reward;
    int32_t  remap_actions;        /* 1: a' = (a + 1) / 2 * (high - low) + low (wrappers.py:128-130) */
    float    action_low[2];        /*   (motor, steering), dream.py:138                        */
    float    action_high[2];
    int32_t  time_limit_steps;     /* TimeLimit duration in rc_step calls; 0 = off (wrappers.py:137-158) */
    int32_t  auto_reset;           /* 1: finished envs are reset inside rc_step (batched rollouts) */
    int32_t  lidar_transform;      /* RC_LIDAR_*: scaling applied when the scan is stored                  */
    void    *external_arena;       /* optional caller-owned device memory for the output arena */
    size_t   external_arena_bytes; /*   must be >= rc_arena_bytes(cfg)                         */
    void    *stream;               /* optional hipStream_t to run on; NULL = library creates one */
    int32_t  car_task[RC_MAX_CARS];/* task of car slot a (agents A, B, C, D of a scenario yml); -1 = `task`          */
    int32_t  n_steps;              /* RC_TASK_N_STEP_PROGRESS window in sub-steps, 1..RC_NSTEP_MAX (yml n_steps: 10) */
    /* Several handles filling ONE arena - one handle per track, so that a batch can mix tracks by blocks of envs (SURVEY.md
     * 8e "per-env track"; BASELINE configs[4] on one GPU): the arena (external_arena) is laid out for arena_total_cars cars
     * and this handle's cars are cars arena_first_car ... of it; every output section then holds the cars of all handles in
     * order.  0 / 0 = the arena is this handle's alone.  rc_trajectory_slab, the compact record and the gathers belong to the
     * arena's owner then, not to a slice handle.  Give each handle first_env = the global index of its first env. */
    int32_t  arena_total_cars;
    int32_t  arena_first_car;
} rc_config;

/* Fill `cfg` with the defaults of the reference's max_progress scenario. */
void rc_default_config(rc_config *cfg);

/* Bytes of device memory the output arena needs for this configuration (for arena_total_cars cars if that is set). */
size_t rc_arena_bytes(const rc_config *cfg);
/* Where a field's section starts in that arena and how many bytes a car takes in it (0 if the field is not enabled):
 * car c's data at section_offset + c * bytes_per_car. */
int rc_field_layout(const rc_config *cfg, int32_t field, size_t *section_offset, size_t *bytes_per_car);

int rc_create(const rc_config *cfg, rc_env **out);
void rc_destroy(rc_env *env);

/*
 * Upload one compiled track (host pointers).  Bitmaps are uint32 [h][pitch], bit i of word j =
 * cell ix = 32*j + i, row iy = 0 is the southern-most.  progress is float32 [h][w]
 * (norm_distance_from_start, generate-costmap.py:220-222; < 0 outside the drivable area),
 * centerline float32 [n][4] = x, y, heading, progress (spawn table for rc_reset).
 */
int rc_load_track(rc_env *env, const uint32_t *occ_words, const uint32_t *drivable_words,
                  const float *progress, int32_t h, int32_t w, int32_t pitch,
                  float resolution, float origin_x, float origin_y,
                  const float *centerline, int32_t n_centerline);

/*
 * Where the track's grid lies in the source image it was cropped from - what the reference's GridMap.to_pixel indexes
 * (dreamer/wrappers.py:396: row = int(H - (y - oy) / res), col = int((x - ox) / res), north-up, the whole image).  Needed by
 * RC_OBS_LIDAR_OCCUPANCY_REFERENCE only, before the first reset: full_height = H; a north-up pixel (R, C) of the image is cell
 * (gx, gy) = (C - col0, row_top - R) of the grid rc_load_track was given; (origin_x, origin_y) = world position of the image's
 * lower left corner; resolution in metres per cell - all three as binary64, as the reference computes with them.
 */
int rc_set_source_frame(rc_env *env, int32_t full_height, int32_t row_top, int32_t col0, double origin_x, double origin_y,
                        double resolution);

/* Reset the envs selected by the host mask (uint8 [num_envs], NULL = all) and produce their
 * first observation. */
int rc_reset(rc_env *env, const uint8_t *mask_or_null, int32_t mode, uint64_t seed);

/* One agent step = up to `repeat` simulator sub-steps of dt = 0.01 s, then the observation.
 * `actions_dev` is device memory float32 [n, 2] = (motor, steering); NULL = use the buffer
 * behind RC_F_ACTION_IN (e.g. after rc_fill_random_actions).  motor >= 0 accelerates, < 0 brakes; a POSITIVE steering command
 * turns RIGHT - towards higher beam indices (beam 0 is at +135 deg on the left) - and +-1 is a front-wheel angle of 0.19 rad:
 * the convention under which the reference's own trained agents (ros_agent/checkpoints) drive (tests/test_golden_policy.py). */
int rc_step(rc_env *env, const float *actions_dev, int32_t repeat);
/* Same, actions in host memory (copied with the stream). */
int rc_step_host(rc_env *env, const float *actions_host, int32_t repeat);

/* Teleport: overwrite every car's pose from host memory, float32 [n, 3] = x, y, yaw (|yaw| <= pi), keep the
 * rest of the state, and recompute the observation (LiDAR, patch).  The analogue of setting the base pose of
 * the vehicle body in the reference's simulator; used by evaluation tooling and by the raycast parity tests. */
int rc_set_pose(rc_env *env, const float *xyyaw_host);

/* Fill RC_F_ACTION_IN with U(-1,1)^2 from Philox4x32-10 keyed by (seed, step, global car id). */
int rc_fill_random_actions(rc_env *env, uint64_t seed, uint32_t step);
/* rc_fill_random_actions(seed, step) followed by rc_step(NULL, repeat) as ONE pass: the dynamics kernel draws the
 * same actions itself (and leaves them in RC_F_ACTION_IN).  The step of a synthetic random-action rollout - the
 * reference's default prefill policy is random actions too (dreamer/dream.py:207-210) - without a
 * launch of its own for the action generator.  Results are identical to the two-call form. */
int rc_step_random(rc_env *env, uint64_t seed, uint32_t step, int32_t repeat);

/* rc_step / rc_step_random of SEVERAL handles as one launch per kernel - the handles of a batch that mixes tracks (one handle
 * per track, each filling its block of cars of ONE arena: rc_config.arena_total_cars / arena_first_car; SURVEY.md 8e "per-env
 * track_id selecting a device-resident grid", BASELINE configs[4]'s track mix).  Every wave of the launch works from the
 * parameters of the block it lies in, so the chip sees one dynamics and one scan kernel over all cars instead of one small pair
 * per track.  The handles share a device, a stream and cars_per_env (at most 8 of them); `actions_dev` is float32 [total cars, 2]
 * in arena order, NULL = every handle's RC_F_ACTION_IN.  Results are those of stepping the handles one by one. */
int rc_step_group(rc_env **envs, int32_t n, const float *actions_dev, int32_t repeat);
int rc_step_random_group(rc_env **envs, int32_t n, uint64_t seed, uint32_t step, int32_t repeat);

/* Batched follow-the-gap agent on the device (the prefill / baseline agent of dreamer/dream.py:211-216, whose
 * host form is agents.gap_follower.GapFollower): from the current LiDAR scan of every car, clip to 6 m,
 * 5-beam smoothing over the forward 202.5 deg, safety bubble of +-60 beams around the closest return, point the wheels at
 * the centre of the widest run of beams whose smoothed range exceeds 2 m (full lock beyond 0.19 rad).  Writes (motor, steering) into RC_F_ACTION_IN: steering in [-1, 1],
 * motor = motor_corner if |steering| > 0.35 else motor_straight (values in the caller's action convention). */
int rc_follow_the_gap(rc_env *env, float motor_straight, float motor_corner);

/* The REFERENCE's follow-the-gap law on the device (ros_agent/agents/follow_the_gap/src/agent.py:128-193: forward arc
 * of +-90 deg clipped at the look-ahead distance, disparities found with 10-degree median / maximum filters and extended
 * by the vehicle's half-width, heading = mean angle of the beams at or above the 83.3rd percentile; :200-234: steering =
 * 1.4 heading - 0.1 d(heading)/dt clipped to +-24 deg, speed 6 m/s less up to 30 % with the steering angle, at most 4/5
 * of the free distance below 5 m, at least 1.5 m/s).  Writes (motor, steering) into RC_F_ACTION_IN - the node's speed over
 * the car's top speed, and the command that puts the front wheels at the node's steering angle (positive command = right; full
 * lock beyond the car's 0.19 rad), in the caller's action convention (the remap of
 * rc_config is inverted when it is on).  dt = seconds per agent step (the derivative term; none on an episode's first
 * command).  detail_dev: optional device float32 [n, 4] = heading [rad], free distance [m], steering angle [rad], speed
 * [m/s].  The generic bubble / widest-gap agent above stays available as rc_follow_the_gap. */
int rc_follow_the_gap_reference(rc_env *env, float dt, float *detail_dev);

int rc_get(rc_env *env, int32_t field, void **dev_ptr, size_t *bytes);
int rc_copy_out(rc_env *env, int32_t field, void *host_dst, size_t bytes);
/* The trajectory record of the last step as one contiguous device slab (fields LIDAR..TIME,
 * plus OCCUPANCY when enabled): the source buffer of the multi-GPU all-gather. */
int rc_trajectory_slab(rc_env *env, void **dev_ptr, size_t *bytes);

/* Device memory for clients that do not link the HIP runtime themselves (a plain-C learner process): allocation on the
 * handle's device, release, and a stream-ordered copy of caller-chosen device bytes to the host.  What they hand out
 * is ordinary device memory: usable as compact slab, gather destination or arena. */
int rc_device_alloc(rc_env *env, size_t bytes, void **dev_ptr);
int rc_device_free(rc_env *env, void *dev_ptr);
int rc_copy_from_device(rc_env *env, const void *dev_src, void *host_dst, size_t bytes);

/* ---- Half-size record and multi-GPU gather -------------------------------------------------------
 * The record of a step is 4 396 B per car, 4 320 of them the fp32 LiDAR row.  rc_set_compact_slab makes the scan
 * store a second copy of the row as uint16 - q = rne((v + off) * scale) with (off, scale) = (0, 65535/15) for
 * RC_LIDAR_METRES, (0.5, 65535) for RC_LIDAR_DREAMER, (0, 65535) for RC_LIDAR_UNIT: 0.23 mm per count, below the
 * 0.05 m map cell by two orders of magnitude - into a caller-owned device buffer, followed by a copy of the arena's
 * POSE..TIME sections (76 B per car): 2 236 B per car, the payload of RC_GATHER_FULL_U16.  Layout of the buffer:
 * uint16 [n][1080], padding to 64 B, then the POSE..TIME sections exactly as they lie in the arena
 * (rc_compact_layout gives the offsets).  NULL switches it off.  Alternate two buffers to overlap a gather with
 * the next step. */
size_t rc_compact_bytes(const rc_config *cfg);
int rc_set_compact_slab(rc_env *env, void *slab, size_t bytes);
int rc_compact_layout(rc_env *env, size_t *lidar_u16_bytes, size_t *summary_offset, size_t *summary_bytes);

/* The communicator: one rank per handle (= per GPU).  Rank 0 calls rc_comm_unique_id and hands the 128 bytes to the
 * other ranks by any means (file, socket, MPI, torch.distributed store); every rank then calls rc_comm_init.  RCCL is
 * bound at run time: the first of librccl.so.1 / librccl.so / /opt/rocm/lib that is already loaded in the process or
 * can be loaded, or the path given to rc_comm_library before the first use. */
int rc_comm_library(const char *path);
int rc_comm_unique_id(void *out_128_bytes, size_t bytes);
int rc_comm_init(rc_env *env, const void *unique_id, size_t bytes, int32_t rank, int32_t world);
int rc_comm_count(rc_env *env, int32_t *ranks);      /* ncclCommCount of the handle's communicator */

/* All-gather of the last step's record over the communicator: dev_dst receives world x rc_gather_bytes(mode) bytes,
 * rank r's record at r * rc_gather_bytes(mode).  RC_GATHER_FULL = the rc_trajectory_slab bytes (fp32 LiDAR),
 * RC_GATHER_FULL_U16 = the compact slab, RC_GATHER_SUMMARY = POSE..TIME only (the scans stay on their GPU).
 * Asynchronous: queued behind the work already on the handle's stream, runs on a stream of its own so that the next
 * steps overlap it - the caller must not let them overwrite the source (rc_set_arena / rc_set_compact_slab to the
 * other buffer of a pair) nor reuse dev_dst before rc_gather_wait.  rc_gather_wait orders the handle's stream behind
 * the collective; with host_sync != 0 it also blocks the host until the gathered bytes are there. */
enum { RC_GATHER_FULL = 0, RC_GATHER_FULL_U16 = 1, RC_GATHER_SUMMARY = 2 };
size_t rc_gather_bytes(rc_env *env, int32_t mode);
int rc_gather_trajectory(rc_env *env, int32_t mode, void *dev_dst, size_t dst_bytes);
int rc_gather_wait(rc_env *env, int32_t host_sync);

/* ---- The same all-gather as DIRECT PEER COPIES (on the xGMI full mesh each rank's record should cross
 * each link once - N - 1 concurrent copies into the peers' buffers - where a ring passes it on N - 1 times).  No RCCL:
 * hipIpc memory handles, one copy stream per peer, sequence flags in uncached device memory.
 *   rc_p2p_setup    allocates this rank's destination - two slots of `world` entries, each entry sized for the LARGEST
 *                   payload (rc_gather_bytes(RC_GATHER_FULL), rounded up to 256: rc_p2p_stride) - and flag block, and
 *                   writes an RC_P2P_EXPORT_BYTES blob; the caller hands every rank's blob to every rank (file, socket,
 *                   MPI, torch.distributed - the library does not care), rank r's at offset r * RC_P2P_EXPORT_BYTES.
 *                   Called again with another mode it only switches the payload (same buffers, same blob, sequence
 *                   numbers run on; every rank switches at the same gather);
 *   rc_p2p_connect  opens the peers' buffers (a no-op once connected);
 *   rc_gather_trajectory_p2p   sends the last step's record (the source of `mode`, see rc_gather_trajectory; the caller
 *                   double-buffers it the same way) into slot k & 1 of every rank, k = 0, 1, ... counting the calls:
 *                   queued behind the work on the handle's stream, runs on streams of its own;
 *   rc_gather_p2p_wait         orders the handle's stream (host_sync != 0: and the host) behind the COMPLETION of the LAST
 *                   issued gather - every rank's record has arrived here AND this rank's outbound copies have read their
 *                   source to the end - and returns the slot (*bytes = world x stride): rank r's record
 *                   - rc_gather_bytes(mode) bytes of it - at r * stride, stride = *bytes / world.  The slot
 *                   stays valid until the call that issues the gather after next; a peer that does not take part within
 *                   RC_P2P_TIMEOUT_S (20 s) makes the host-synchronising wait return RC_ERR_COMM instead of blocking
 *                   (reported once, the counter then starts again; the slots of that gather and the one before are not
 *                   valid on any rank - tear the transport down and set it up again).
 *                   THE RULE FOR THE SOURCE: gather k reads the record in place, asynchronously.  Call
 *                   rc_gather_p2p_wait(env, 0, ...) BEFORE issuing gather k + 1 - the handle's stream then waits for gather k,
 *                   so the step after next, which rewrites gather k's source in a double-buffered pair, runs behind it.
 *                   Without that call nothing orders a later step behind the outbound copies (records may be torn);
 *   rc_p2p_slot     the slot of the last issued gather (back = 0) or of the one before it (back = 1), without waiting;
 *   rc_p2p_disconnect          waits for this rank's copies and unmaps the peers' buffers; rc_p2p_teardown frees this rank's
 *                   own.  Exported memory must not be freed while a peer still maps it: EVERY rank disconnects, the caller
 *                   synchronises the ranks (a barrier of its own), THEN the ranks tear down (rc_destroy tears down too).
 * Every rank must issue the same sequence of gathers.  Ranks may share a GPU (functional tests) or sit on one each. */
#define RC_P2P_EXPORT_BYTES 256
int rc_p2p_setup(rc_env *env, int32_t mode, int32_t rank, int32_t world, void *export_out, size_t bytes);
int rc_p2p_connect(rc_env *env, const void *exports_world_x_256, size_t bytes);
int rc_gather_trajectory_p2p(rc_env *env);
int rc_gather_p2p_wait(rc_env *env, int32_t host_sync, void **gathered_dev, size_t *gathered_bytes);
int rc_p2p_slot(rc_env *env, int32_t back, void **gathered_dev, size_t *gathered_bytes);
int rc_p2p_disconnect(rc_env *env);
int rc_p2p_teardown(rc_env *env);

/* Re-point the output fields (everything rc_get returns except RC_F_ACTION_IN, which stays where it is) at
 * another device buffer of at least rc_arena_bytes(), 64-byte aligned; NULL = back to the handle's own arena.
 * Stream-ordered: the next rc_reset / rc_step / rc_set_pose writes there.  This is how a device-resident
 * trajectory ring is filled without copies - the step after Collect.step in the reference
 * (dreamer/wrappers.py:213-219, dreamer/tools.py:235-264): one arena per time slot, rotate before each step.
 * Lifetime: the arena a step wrote must stay allocated until the NEXT observation has been produced (small batches take
 * the cars longest-scan-first and read the previous rows for that); after rc_set_arena(env, NULL, 0) nothing of a lent
 * arena is read again, so that is the call to make before freeing one. */
int rc_set_arena(rc_env *env, void *arena, size_t bytes);

/* Rows out of a ring of arenas - the window gather of a replay sampler (the reference reads fixed-length windows out of
 * episode files: dreamer/tools.py:235-264).  ring_base + k * slot_bytes is arena k of a ring filled through rc_set_arena;
 * output row r takes the record of car car_idx[r] in slot slot_idx[r] (device int32 arrays), for every field of
 * field_mask (bit f = rc_field f): section f of the output holds n_rows records of that field back to back, sections in
 * field order, each starting on a 64-byte boundary (rc_gather_rows_bytes = the total).  Queued on the handle's stream.
 * ring_base and slot_bytes must be multiples of 64 (every slot is an arena as rc_set_arena takes it). */
size_t rc_gather_rows_bytes(rc_env *env, uint32_t field_mask, int32_t n_rows);
int rc_gather_rows(rc_env *env, const void *ring_base, size_t slot_bytes, const int32_t *slot_idx_dev, const int32_t *car_idx_dev,
                   int32_t n_rows, uint32_t field_mask, void *out_dev, size_t out_bytes);

/* Window starts for that gather, drawn on the device (the reference draws a random index into an episode file and takes
 * `length` steps from there: dreamer/tools.py:250-262).  The ring holds `count` consecutive records (`oldest` = slot of the
 * oldest, capacity slots in all); n_windows windows of `length` records of one car each, (first record, car) uniform -
 * Philox4x32-10 keyed by `seed`, counter (window, try, draw) - among the windows that stay inside one episode: no fresh record
 * strictly inside, a fresh last record only if it is the terminal transition (done).  Device int32 outputs: slot_idx /
 * slot_obs_idx / car_idx [n_windows * length] = the rows for rc_gather_rows (slot_obs_idx: a terminal row reads the slot before
 * it - for the observation fields), meta [n_windows * 4] = (ring age of the first record, car, terminal, starts an episode);
 * *failed_dev is incremented for every window that found no such start in max_tries draws.  Queued on the handle's stream. */
int rc_sample_windows(rc_env *env, const void *ring_base, size_t slot_bytes, int32_t capacity, int32_t oldest, int32_t count,
                      int32_t length, int32_t n_windows, uint64_t seed, uint32_t draw, int32_t max_tries, int32_t *slot_idx_dev,
                      int32_t *slot_obs_idx_dev, int32_t *car_idx_dev, int32_t *meta_dev, uint32_t *failed_dev);

/* One training batch in ONE call and ONE buffer: rc_sample_windows + the row gather of both kinds (observation fields
 * through slot_obs_idx, the others through slot_idx) + the reference's reset row (first row of a window that starts an episode:
 * action 0, reward 0, discount 1, time 0, progress_total -1; dreamer/wrappers.py:221-226; reset_rows = 0: records as stored).
 * Layout of out_dev (64-byte aligned): the sections of field_mask's fields in field order, each n_windows * length records,
 * each 64-byte aligned; meta int32 [n_windows][4] at *meta_offset; a 64-byte block whose first uint32 counts the windows that
 * found no episode-internal start.  Those *payload_bytes are what a sharded replay store sends (replay.ShardedReplay:
 * one collective per batch); the sampler's row indices follow as scratch, rc_sample_batch_bytes = the total to allocate.
 * A memset and two launches on the handle's stream. */
size_t rc_sample_batch_bytes(rc_env *env, uint32_t field_mask, int32_t n_windows, int32_t length, size_t *payload_bytes, size_t *meta_offset);
int rc_sample_batch(rc_env *env, const void *ring_base, size_t slot_bytes, int32_t capacity, int32_t oldest, int32_t count, int32_t length,
                    int32_t n_windows, uint64_t seed, uint32_t draw, int32_t max_tries, uint32_t field_mask, int32_t reset_rows,
                    void *out_dev, size_t out_bytes);

int rc_sync(rc_env *env);
void *rc_stream(rc_env *env);      /* the hipStream_t the handle launches on */

/* Per-kernel timing with HIP events recorded on the handle's stream.  enabled: 0 = off, 1 = every kernel,
 * otherwise a bit mask (1 << RC_K_*) of the kernels to time (fewer events in a timed region). */
int rc_set_profiling(rc_env *env, int32_t enabled);
int rc_kernel_time(rc_env *env, int32_t kernel, double *total_ms, uint64_t *launches);
int rc_reset_kernel_times(rc_env *env);

/* Raycast implementation selector, 0..7 (all variants return identical results; 7, the default, is the
 * fastest: per-cell, per-quadrant free rectangles, one wave per car; 0 is the cell-by-cell reference traversal). */
int rc_set_raycast_variant(rc_env *env, int32_t variant);
/* The symbol of the scan kernel the next rc_step launches, as rocprofv3 lists it (without namespace and argument list),
 * e.g. "rc_raycast_car_kernel<1, false, false>": template arguments = cars per env, next round prepared under the first
 * request (small batches), bounded trip loop. */
int rc_scan_kernel_name(rc_env *env, char *out, size_t bytes);

/* Experiment / validation knobs of the scan - NOT part of the product interface; every knob is 0 in production and
 * the library reads nothing from the process environment.  RAY_THREADS / RAY_SPLIT / RAY_WG_PER_CU: launch geometry
 * sweeps (tools/knob_sweep.sh); BAND_LOG2 in [-40, -10]: width of the exact-count zone of the scan as
 * max(w, h) * 2^value cells instead of 2^-21 - tests/test_gpu_parity.py narrows it to show that its corner-aimed rays
 * detect a band below the rounding bound.  Takes effect immediately (also after rc_load_track). */
enum { RC_DBG_RAY_THREADS = 0, RC_DBG_RAY_SPLIT = 1, RC_DBG_RAY_WG_PER_CU = 2, RC_DBG_BAND_LOG2 = 3,
       RC_DBG_PATCH_VARIANT = 4,    /* lidar_occupancy render experiment: bit 1 = plain instead of non-temporal stores */
       RC_DBG_SCAN_BOUNDED = 5,     /* != 0: the scan runs the build whose trip loop carries a trip budget (see below) */
       RC_DBG_SCAN_ORDER = 6,   /* 0 = production (the scan takes the cars in track order, sorted every 64 observations), 1 = car index order, k > 1 = sorted every k - 1 observations */
       RC_DBG_EXACT_CHUNK = 7,  /* k > 0: the exact render (RC_OBS_LIDAR_OCCUPANCY_REFERENCE) works on k cars at a time instead of 6 144
                                 * (at most the size its scratch was allocated for) */
    RC_DBG_COUNT = 8 };
int rc_debug_set(rc_env *env, int32_t knob, int32_t value);

/* The default scan's trip loop is unbounded in the production build (its termination is a property of the tables and
 * of the exact-count band, derived in racecar_kernels.hip; a bound costs 4 % of the scan).  The BOUNDED build of the same
 * kernel - a wave-level budget of w + h + 2 trips per round; a ray that uses it up reads "no return" - runs (a) inside
 * rc_load_track over every cell a sensor can stand in (a track whose tables make any ray overrun is refused), (b) while a
 * validation band is set, (c) under RC_DBG_SCAN_BOUNDED.  rc_scan_overruns reports how many waves of this handle's
 * bounded scans have used up a budget so far (0 unless a band was mis-set or a table is corrupt). */
int rc_scan_overruns(rc_env *env, uint64_t *count);

/* In-kernel time stamps of the default scan (analysis only, like the knobs above): when `stamps` is non-NULL the next
 * scans run an instrumented build of the one-wave-per-car kernel (1 car per env only) whose first `n_waves` waves write
 * RC_STAMP_SLOTS uint64 each to the DEVICE buffer `stamps` - shader-clock values (s_memtime) at fixed points of the wave's
 * life plus two counters; slot meanings: tools/scan_stamps.py.  NULL switches back to the production kernel. */
#define RC_STAMP_SLOTS 32
int rc_debug_scan_stamps(rc_env *env, uint64_t *stamps, int32_t n_waves);

/* Host-only: the beam (cos, sin) and footprint tables the kernels use (float32 [1080][2], [34][2]). */
void rc_spec_tables(float *beams_1080x2, float *footprint_34x2);

/* Device self-test of an arithmetic property the raycast kernel relies on: v_rcp_f32 + one FMA Newton step is
 * the correctly rounded (IEEE) reciprocal.  Checks every fp32 of both signs with biased exponent in
 * [27, 227] (2^-100 .. 2^100, 3.37e9 values) on `device`; *n_mismatch must come back 0. */
int rc_selftest_reciprocal(int32_t device, uint64_t *n_checked, uint64_t *n_mismatch);

/* The same for the square root of the reference follow-the-gap agent's arccos (ros_agent/agents/follow_the_gap/src/agent.py:171:
 * np.arccos; spec: oracle.acos32): v_sqrt_f32 alone is good to 1 ulp, the kernel's fix-up (two fused residuals) makes it the
 * correctly rounded root the spec's np.sqrt is.  Checks every positive binary32 from 2^-60 to 2^10 (587 M values) and zero
 * against the double-precision root rounded once; *n_mismatch must come back 0. */
int rc_selftest_sqrt(int32_t device, uint64_t *n_checked, uint64_t *n_mismatch);

/* The exact render (RC_OBS_LIDAR_OCCUPANCY_REFERENCE) divides the spline weights by 6 without a division instruction (a quotient
 * estimate, its exact remainder by one fused multiply-add, one correction: correctly rounded by Markstein's theorem).  This
 * compares that with the device's own binary64 division over 2^32 operands from 2^-160 to 8, either sign; n_mismatch must be 0. */
int rc_selftest_div6(int32_t device, uint64_t *n_checked, uint64_t *n_mismatch);

/* The exact render decides a pixel of the rotated window by a binary32 estimate of its 16-tap sum wherever that estimate lies more
 * than 1e-3 from a rounding boundary, and by the library's binary64 sum elsewhere (racecar_patch_exact.h, PX_BAND: the estimate's
 * error is bounded by 1.1e-4).  This renders the env's cars once more with BOTH computed for every pixel: out[0] = pixels inside
 * the array, out[1] = those the band sent to the binary64 sum, out[2] = those the estimate alone would have got wrong (0, or the
 * bound does not hold), out[3] = the largest |estimate - binary64 sum| seen, as the bits of a binary32.  The patches are written
 * as by an observation.  tests/test_gpu_api.py. */
int rc_selftest_exact_estimate(rc_env *env, uint64_t out[4]);

const char *rc_last_error(void);
int rc_abi_version(void);
/* Hash (32 hex digits) of the compiler flags and of the contents of every source and header this library was built
 * from - what racing_dreamer_amd/build.py compares with the tree beside it to decide whether to compile. */
const char *rc_build_id(void);

#ifdef __cplusplus
}
#endif
#endif /* RACECAR_HIP_H */
