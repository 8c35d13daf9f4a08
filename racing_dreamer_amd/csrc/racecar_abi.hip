// C-ABI of libracecar_hip.so (declared in include/racecar_hip.h): handle, device buffers,
// stream-ordered launches.  No exceptions cross the boundary; errors are codes + rc_last_error().
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <unistd.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iterator>
#include <map>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../../include/racecar_hip.h"
#include "racecar_internal.h"
#include "racecar_spec.h"

namespace {

thread_local std::string g_last_error;

int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess)                                                                      \
            return fail(RC_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct RcUid { char internal[128]; };      // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128), passed by value

// bytes per car of every rc_field, in arena order
const size_t kFieldBytes[RC_F_COUNT] = {
    RC_N_BEAMS * 4, 24, 24, 4, 8, 4, 4, 4, 4, RC_PATCH * RC_PATCH,   // LIDAR .. OCCUPANCY
    4, 4, 4, 1, 1, 1, 1, 1, 1, 4, 4, 8,                               // PROGRESS .. ACTION_IN
};

struct Layout {
    size_t offset[RC_F_COUNT];
    size_t bytes[RC_F_COUNT];
    size_t slab_bytes;   // LIDAR..TIME (+OCCUPANCY when rendered)
    size_t total;
};

// The arena holds `n_cars` cars; a handle that owns only cars [first_car, first_car + n_own) of it (rc_config::arena_total_cars:
// several handles - one per track - fill ONE set of output arrays) gets the offsets and sizes of ITS slice of every section.
Layout make_layout(int n_cars, bool occupancy, int first_car = 0, int n_own = -1) {
    Layout l{};
    if (n_own < 0) n_own = n_cars;
    size_t off = 0;
    for (int f = 0; f < RC_F_COUNT; ++f) {
        size_t b = kFieldBytes[f] * (size_t)n_cars;
        if (f == RC_F_OCCUPANCY && !occupancy) b = 0;
        l.offset[f] = off + (b ? kFieldBytes[f] * (size_t)first_car : 0);
        l.bytes[f] = b ? kFieldBytes[f] * (size_t)n_own : 0;
        if (f == (occupancy ? RC_F_OCCUPANCY : RC_F_TIME)) l.slab_bytes = off + b;
        off = align_up(off + b, 64);
    }
    l.total = off;
    return l;
}

// The half-size trajectory record (rc_set_compact_slab): uint16 LiDAR rows, then a copy of the arena's POSE..TIME
// sections (same relative layout, 64-byte aligned sections).
struct CompactLayout {
    size_t lidar_bytes;      // n * 1080 * 2, rounded up to 64
    size_t summary_src_off;  // offset of RC_F_POSE in the arena
    size_t summary_bytes;    // RC_F_POSE .. end of RC_F_TIME
    size_t total;
};

CompactLayout make_compact(const Layout &l, int n_cars) {
    CompactLayout c{};
    c.lidar_bytes = align_up((size_t)n_cars * RC_N_BEAMS * 2, 64);
    c.summary_src_off = l.offset[RC_F_POSE];
    c.summary_bytes = l.offset[RC_F_TIME] + l.bytes[RC_F_TIME] - l.offset[RC_F_POSE];
    c.total = align_up(c.lidar_bytes + c.summary_bytes, 64);
    return c;
}

// ---- RCCL, bound at run time (rc_comm_init): the library has no link-time dependency on it, so single-GPU clients
// need no RCCL installed, and a process that already holds a copy (PyTorch bundles one) keeps using that copy.
struct Rccl {
    void *handle = nullptr;
    int (*GetUniqueId)(void *) = nullptr;
    int (*CommInitRank)(void **, int, /* ncclUniqueId by value */ struct RcUid, int) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*CommCount)(void *, int *) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};

// Device tables of one compiled track (bitmaps, progress grid, spawn table, the scan's rectangle planes and first-trip
// table: 30 - 420 MB, built on the device in 20 - 100 ms).  They are read-only and depend on nothing but the track, so
// handles of one process that load the same track on the same device share one copy: the second rc_load_track of a
// track costs a hash of its inputs instead of an upload and a rebuild (tests and multi-handle clients create many).
struct TrackTables {
    int device = 0;
    void *mem = nullptr;
    // what the tables were built from, compared on a cache hit besides the 64-bit key: shape, geometry and a second,
    // independent checksum of the arrays (a key collision must not hand a handle another track's tables)
    int32_t h = 0, w = 0, pitch = 0, n_centerline = 0;
    float res = 0.f, ox = 0.f, oy = 0.f;
    uint64_t sum2 = 0;
    RcTrackDev t{};
    size_t lds_bytes = 0, lds_bytes_skip = 0, lds_bytes_packed = 0;
    ~TrackTables() {
        if (mem) {
            (void)hipSetDevice(device);
            (void)hipFree(mem);
        }
    }
};
std::mutex g_track_mutex;
std::map<std::pair<int, uint64_t>, std::weak_ptr<TrackTables>> g_track_cache;

uint64_t fnv1a(uint64_t h, const void *data, size_t n) {
    const unsigned char *b = (const unsigned char *)data;
    for (size_t i = 0; i < n; ++i) h = (h ^ b[i]) * 0x100000001b3ull;
    return h;
}

uint64_t wordsum(uint64_t acc, const void *data, size_t n) {     // position-weighted sum of the 32-bit words (n % 4 == 0)
    const uint32_t *w = (const uint32_t *)data;
    for (size_t i = 0; i < n / 4; ++i) acc += (uint64_t)w[i] * (2 * i + 1) + 0x9e3779b97f4a7c15ull;
    return acc;
}

struct EventPair {
    hipEvent_t a, b;
    int kernel;
};

// ---- peer-copy all-gather (on the xGMI full mesh every shard crosses exactly one link once if each rank
// copies its record straight into every peer's buffer - N - 1 concurrent copies - where a ring passes it N - 1 times).
// Every rank owns: the destination, two slots of world x bytes (hipMalloc, exported over hipIpc), and a block of
// sequence flags in uncached device memory that its PEERS write: arrived[p] = k + 1 when peer p's shard of gather k has
// landed, released[p] = k + 1 when peer p allows gather k to be written into ITS slot k & 1.
struct P2pExport {                 // what a rank hands to its peers (RC_P2P_EXPORT_BYTES)
    hipIpcMemHandle_t dst, flags;
    uint64_t bytes;                // per rank and slot
    int32_t rank, world, mode, pid;
    char pci[32];
    char pad[RC_P2P_EXPORT_BYTES - 2 * sizeof(hipIpcMemHandle_t) - 8 - 16 - 32];
};
static_assert(sizeof(P2pExport) == RC_P2P_EXPORT_BYTES, "export blob size");

struct P2p {
    int rank = 0, world = 0, mode = 0;
    size_t bytes = 0;              // one rank's record in the current mode
    size_t cap = 0;                // ... and in the largest one (RC_GATHER_FULL): what the slots are sized for
    P2pExport blob{};              // what rc_p2p_setup handed out
    char *dst = nullptr;           // [2][world][cap], mine
    uint32_t *flags = nullptr;     // arrived[64] | released[64] | timeouts, mine (uncached)
    std::vector<char *> peer_dst;          // peers' destinations, opened (null for me)
    std::vector<uint32_t *> peer_flags;    // peers' flag blocks, opened (null for me)
    std::vector<hipStream_t> push;         // one stream per peer (the local copy runs on push[rank])
    hipStream_t ctrl = nullptr;            // release + wait-for-release kernels; arrival waits
    hipEvent_t ev_ready = nullptr, ev_go = nullptr, ev_arrived = nullptr, ev_local = nullptr;
    std::vector<hipEvent_t> ev_sent;       // per peer: my copy into its slot and the arrival flag behind it have been executed
    uint32_t issued = 0;           // gathers issued so far
    bool connected = false;
    uint32_t *arrived() const { return flags; }
    uint32_t *released() const { return flags + RC_P2P_MAX_RANKS; }
    uint32_t *timeouts() const { return flags + 2 * RC_P2P_MAX_RANKS; }
};

}  // namespace

struct rc_env {
    rc_config cfg{};
    int n_cars = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    // device memory
    void *arena = nullptr;
    bool own_arena = false;
    bool shared_arena = false;     // this handle fills a slice of a larger arena (rc_config::arena_total_cars)
    Layout layout{};
    void *state_mem = nullptr;
    std::shared_ptr<TrackTables> track;   // shared with the other handles that loaded the same track on this device
    uint8_t *mask_dev = nullptr;
    float *actions_in = nullptr;   // inside the arena the handle was created with (RC_F_ACTION_IN)
    void *out_arena = nullptr;     // where the output fields currently point (rc_set_arena)
    RcParams params{};
    RcLaunchInfo launch{};
    bool has_track = false;
    bool was_reset = false;
    // profiling
    uint32_t profiling = 0;        // bit k set: time kernel k with HIP events
    std::vector<EventPair> pending;
    std::vector<EventPair> free_events;
    double k_ms[RC_K_COUNT] = {0};
    uint64_t k_n[RC_K_COUNT] = {0};
    int32_t dbg[RC_DBG_COUNT] = {0};   // rc_debug_set: experiment / validation knobs, all 0 = production behaviour
    // half-size record + multi-GPU gather
    CompactLayout compact{};
    void *compact_slab = nullptr;      // caller-owned device buffer of compact.total bytes, or null
    void *comm = nullptr;              // ncclComm_t
    int comm_rank = 0, comm_world = 0;
    hipStream_t comm_stream = nullptr;
    hipEvent_t ev_ready = nullptr, ev_gathered = nullptr;
    bool gather_pending = false;
    P2p *p2p = nullptr;                // peer-copy all-gather (rc_p2p_setup), else null
    // obs_type lidar_occupancy_reference (racecar_patch_exact.h): the source frame (rc_set_source_frame), the spline
    // coefficients' scratch for one chunk of cars, Pillow's integer tables on the device
    RcExactParams exact{};
    bool has_frame = false;
    int exact_chunk = 0;
    void *exact_mem = nullptr;
    float *ftg_prev = nullptr;         // rc_follow_the_gap_reference: previous heading per car (NaN = none), allocated on first use
    void *order_mem = nullptr;         // RcStateDev::order + the sort's bucket counters (batches of RC_ORDER_MIN_CARS cars and more)
    uint32_t order_age = 0;            // observations since the cars were last sorted by track position
    const float *last_scan_rows = nullptr;   // the LiDAR rows the last scan of this handle wrote (the small batches' cost keys)
    // rc_step_group (this handle as the first of a group): the blocks' RcParams as the last launch saw them, on the device
    // and on the host (pinned staging slots taken in turn, each with the event of its copy)
    RcParams *group_dev = nullptr;
    RcParams *group_host = nullptr;    // [kGroupSlots][RC_GROUP_MAX], pinned
    hipEvent_t group_ev[4] = {nullptr, nullptr, nullptr, nullptr};
    uint32_t group_slot = 0;
    int group_n = 0;
    RcParams group_last[RC_GROUP_MAX];
};

namespace {

int drain_events(rc_env *env) {
    if (env->pending.empty()) return RC_OK;
    HIP_TRY(hipSetDevice(env->cfg.device));
    HIP_TRY(hipStreamSynchronize(env->stream));
    for (EventPair &ep : env->pending) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, ep.a, ep.b));
        env->k_ms[ep.kernel] += ms;
        env->k_n[ep.kernel] += 1;
        env->free_events.push_back(ep);
    }
    env->pending.clear();
    return RC_OK;
}

struct KernelTimer {
    rc_env *env;
    EventPair ep{};
    bool on = false;
    int begin(rc_env *e, int kernel) {
        env = e;
        if (!((e->profiling >> kernel) & 1u)) return RC_OK;
        if (e->pending.size() >= 4096) {
            int rc = drain_events(e);
            if (rc) return rc;
        }
        if (!e->free_events.empty()) {
            ep = e->free_events.back();
            e->free_events.pop_back();
        } else {
            HIP_TRY(hipEventCreate(&ep.a));
            HIP_TRY(hipEventCreate(&ep.b));
        }
        ep.kernel = kernel;
        rck_set_launch_events(ep.a, ep.b);      // the launch that follows carries the two timestamps itself
        on = true;
        return RC_OK;
    }
    int end() {
        if (!on) return RC_OK;
        env->pending.push_back(ep);
        return RC_OK;
    }
};

#define TIMED(env, kernel, launch_expr)                 \
    do {                                                \
        KernelTimer _t;                                 \
        int _rc = _t.begin(env, kernel);                \
        if (_rc) return _rc;                            \
        HIP_TRY(launch_expr);                           \
        _rc = _t.end();                                 \
        if (_rc) return _rc;                            \
    } while (0)

// Waves per car of the one-wave-per-car scan (the measurements: set_launch_geometry, tools/split_sweep.py).
static int scan_split(long long cars, int n_cu) {
    if (cars > 16LL * n_cu) return 1;
    return (int)std::min<long long>(17, std::max<long long>(1, 48LL * n_cu / std::max<long long>(cars, 1)));
}

// Persistent workgroups: as many as stay resident (2 x 1024 threads = 32 waves/CU is the hardware maximum).
void set_launch_geometry(rc_env *env) {
    RcLaunchInfo &li = env->launch;
    auto blocks_for = [&](size_t lds, long long items, int threads) {
        int wg_per_cu = lds ? (int)((160 * 1024) / lds) : 2;
        wg_per_cu = wg_per_cu > 2 ? 2 : (wg_per_cu < 1 ? 1 : wg_per_cu);
        return (int)std::min<long long>((items + threads - 1) / threads, (long long)li.n_cu * wg_per_cu);
    };
    const long long rays = (long long)env->n_cars * RC_N_BEAMS;
    li.ray_blocks = blocks_for(li.raycast_variant >= 4 ? (size_t)1 : li.raycast_variant == 3 ? li.lds_bytes_packed : (li.raycast_variant != 0 ? li.lds_bytes_skip : li.lds_bytes), rays, li.ray_threads);
    li.patch_variant = env->dbg[RC_DBG_PATCH_VARIANT];
    // tuning knobs for experiments (rc_debug_set; all zero in production): workgroup size / workgroups per CU of the LDS-free scan
    if (li.raycast_variant == 7) {
        const int threads = env->dbg[RC_DBG_RAY_THREADS];
        li.car_threads = (threads == 64 || threads == 128 || threads == 256) ? threads : 64;
        // one wave per car keeps the chip busy only with several waves per wave slot (8 per SIMD x 4 SIMDs x CUs) for the
        // dispatcher to balance; smaller batches split each car's 17 rounds over `split` waves (round k of a car goes to
        // wave k mod split).  Below 8 192 cars every wave is resident at once and the launch lasts as long as its slowest
        // wave: T = 15 us + 2.9 ns x cars from 4 096 cars up.  Measured on columbia (tools/split_sweep.py, round 5, us):
        //     cars   split 1      2      3      4      6      9     17
        //      256     21.6   14.3   12.3   10.9   11.7    9.3    8.9
        //     1024     23.8   16.5   14.8   13.9   13.0   11.8   12.5
        //     2048     24.9   18.6   17.1   16.8   17.2   16.7   18.9
        //     4096     27.0   25.5   24.5   27.4   25.8   26.9   31.8
        //     6144     31.3   34.7   32.7
        //     8192     36.6   41.3   39.8   44.7   43.6   46.7   56.6
        //    16384     62.8   72.2   68.7   81.6   80.2   84.2  105.8
        // Two regimes.  More than 16 cars per CU (half the wave slots): one wave per car - a second wave per car pays the
        // fixed cost (the car's state and first-trip line, 1.3 us) twice and, past 32 waves per CU, waits for a slot
        // (6 144 cars 31 us in one wave each, 35 in two; 8 192 cars 37 / 41: what the rule of rounds 2-4,
        // ceil(48 n_cu / cars) for every batch, chose).  Up to 16 cars per CU the chain of a car's 17 rounds is the
        // launch's duration: floor(48 n_cu / cars) waves per car (4 096 cars in three: 24.5 us against 27.0).
        int split = env->dbg[RC_DBG_RAY_SPLIT];
        if (split < 1 || split > 17) split = scan_split(env->n_cars, li.n_cu);
        li.car_split = split;
    } else if (li.raycast_variant >= 4) {
        const int threads = env->dbg[RC_DBG_RAY_THREADS];
        if (threads >= 64 && threads <= 1024 && threads % 64 == 0) li.ray_threads = threads;
        const int per_cu = env->dbg[RC_DBG_RAY_WG_PER_CU];
        if (threads || per_cu) {
            const long long chunks = (rays + li.ray_threads - 1) / li.ray_threads;
            const long long resident = (long long)li.n_cu * (per_cu > 0 ? per_cu : 2048 / li.ray_threads);
            li.ray_blocks = (int)std::min<long long>(chunks, per_cu < 0 ? chunks : resident);
        }
    }
}

// Half-width of the zone around a cell boundary in which the scan counts the other-axis cell exactly (derivation in
// racecar_kernels.hip).  RC_DBG_BAND_LOG2 is the validation knob of tests/test_gpu_parity.py: a band below the
// rounding bound must make the parity tests fail.
void set_band(rc_env *env) {
    RcTrackDev &t = env->params.trk;
    const int l2 = env->dbg[RC_DBG_BAND_LOG2];
    t.band = std::ldexp((float)(std::max(t.w, t.h) + 2), (l2 <= -10 && l2 >= -40) ? l2 : -21);
    t.band_mh = t.band - 0.5f;
    t.band2 = 2.0f * t.band;
    // With the shipped band every trip provably moves its ray at least one cell along the exit axis (racecar_kernels.hip,
    // ray_traverse), so the loop ends at the ring of stop cells; a narrower validation band voids that proof, and the
    // scan then runs the build whose loop is bounded by a trip budget (also on request: RC_DBG_SCAN_BOUNDED).
    env->launch.scan_guarded = ((l2 <= -10 && l2 >= -40 && l2 != -21) || env->dbg[RC_DBG_SCAN_BOUNDED] != 0) ? 1 : 0;
}

// Point every output field at `arena` (layout of make_layout).  The action input buffer is NOT part of this: it
// stays in the arena the handle was created with, so re-pointing the outputs (rc_set_arena) between producing
// the actions and stepping is safe.
void bind_outputs(rc_env *env, void *arena) {
    char *a = (char *)arena;
    const Layout &l = env->layout;
    RcOutDev &o = env->params.out;
    o.lidar = (float *)(a + l.offset[RC_F_LIDAR]);
    o.pose = (float *)(a + l.offset[RC_F_POSE]);
    o.velocity = (float *)(a + l.offset[RC_F_VELOCITY]);
    o.speed = (float *)(a + l.offset[RC_F_SPEED]);
    o.action = (float *)(a + l.offset[RC_F_ACTION]);
    o.reward = (float *)(a + l.offset[RC_F_REWARD]);
    o.discount = (float *)(a + l.offset[RC_F_DISCOUNT]);
    o.progress_total = (float *)(a + l.offset[RC_F_PROGRESS_TOTAL]);
    o.time = (float *)(a + l.offset[RC_F_TIME]);
    o.patch = (uint8_t *)(a + l.offset[RC_F_OCCUPANCY]);
    o.progress = (float *)(a + l.offset[RC_F_PROGRESS]);
    o.lap = (int32_t *)(a + l.offset[RC_F_LAP]);
    o.cp = (int32_t *)(a + l.offset[RC_F_CHECKPOINT]);
    o.done = (uint8_t *)(a + l.offset[RC_F_DONE]);
    o.trunc = (uint8_t *)(a + l.offset[RC_F_TRUNCATED]);
    o.wall = (uint8_t *)(a + l.offset[RC_F_WALL_COLLISION]);
    o.opp = (uint8_t *)(a + l.offset[RC_F_OPPONENT_COLLISION]);
    o.wrong = (uint8_t *)(a + l.offset[RC_F_WRONG_WAY]);
    o.fresh = (uint8_t *)(a + l.offset[RC_F_FRESH]);
    o.accel = (float *)(a + l.offset[RC_F_ACCELERATION]);
    o.steer = (float *)(a + l.offset[RC_F_STEERING_ANGLE]);
    env->out_arena = arena;
}

// Every RC_ORDER_PERIOD observations (and at the first one after a reset) the cars are sorted by their progress along the track:
// three small launches; the scan then takes them in that order.
int sort_cars_if_due(rc_env *env) {
    const bool small = env->n_cars < RC_ORDER_MIN_CARS;
    if (!env->order_mem || env->dbg[RC_DBG_SCAN_ORDER] == 1 ||
        (small && 4ull * (unsigned long long)env->params.trk.quad_plane_bytes <= RC_ORDER_COST_MIN_TABLE)) {
        env->params.st.order = nullptr;
        return RC_OK;
    }
    const uint32_t period = env->dbg[RC_DBG_SCAN_ORDER] > 1 ? (uint32_t)env->dbg[RC_DBG_SCAN_ORDER] - 1u : (uint32_t)RC_ORDER_PERIOD;
    if (env->params.st.order == nullptr || env->order_age >= period) {
        int32_t *order = (int32_t *)env->order_mem;
        if (env->n_cars >= RC_ORDER_MIN_CARS) {
            HIP_TRY(rck_sort_cars(env->params.st.progress, env->n_cars, (uint32_t *)(order + env->n_cars), order, env->stream));
        } else {
            // small batch: every wave slot is taken at once and the rest of the waves follow as slots come free - longest first
            // (keys from the rows the LAST scan wrote: after rc_set_arena `params.out.lidar` is a slot that was written `capacity`
            // steps ago, or never; with no scan behind it - the first observation - the cars are taken in index order)
            if (env->last_scan_rows == nullptr) {
                env->params.st.order = nullptr;
                return RC_OK;
            }
            float *key = (float *)(order + env->n_cars) + RC_ORDER_BUCKETS;
            HIP_TRY(rck_cost_keys(env->last_scan_rows, env->n_cars, key, env->stream));
            HIP_TRY(rck_sort_cars(key, env->n_cars, (uint32_t *)(order + env->n_cars), order, env->stream));
        }
        env->params.st.order = order;
        env->order_age = 0;
    }
    env->order_age += 1;
    return RC_OK;
}

// obs_type lidar_occupancy_reference: the reference's own render (racecar_patch_exact.h), a chunk of cars at a time
int render_reference_patches(rc_env *env) {
    if (!env->has_frame)
        return fail(RC_ERR_INVALID, "obs_type lidar_occupancy_reference needs rc_set_source_frame (where the grid lies in its source "
                                  "image) before the first observation");
    RcExactParams &x = env->exact;
    const RcParams &p = env->params;
    x.drv_words = p.trk.drv_words; x.pitch = p.trk.pitch; x.h = p.trk.h; x.w = p.trk.w;
    x.x = p.st.x; x.y = p.st.y; x.theta = p.st.theta; x.fresh = p.st.fresh;
    x.patch = p.out.patch;                                   // (rc_set_arena may have moved the outputs)
    int chunk = env->exact_chunk;
    if (env->dbg[RC_DBG_EXACT_CHUNK] > 0 && env->dbg[RC_DBG_EXACT_CHUNK] < chunk) chunk = env->dbg[RC_DBG_EXACT_CHUNK];
    TIMED(env, RC_K_PATCH, rck_launch_patch_exact(x, chunk, env->stream));
    return RC_OK;
}

int observe(rc_env *env) {
    int rc_sort = sort_cars_if_due(env);
    if (rc_sort) return rc_sort;
    TIMED(env, RC_K_RAYCAST, rck_launch_raycast(env->params, env->launch, env->stream));
    env->last_scan_rows = env->params.out.lidar;
    if (env->params.render_patch)
        TIMED(env, RC_K_PATCH, rck_launch_patch(env->params, env->launch, env->stream));
    if (env->cfg.obs_type == RC_OBS_LIDAR_OCCUPANCY_REFERENCE) {
        int rc_px = render_reference_patches(env);
        if (rc_px) return rc_px;
    }
    if (env->compact_slab)      // the scan has written the uint16 rows; the 76 B/car summary follows them
        HIP_TRY(hipMemcpyAsync((char *)env->compact_slab + env->compact.lidar_bytes,
                               (const char *)env->out_arena + env->compact.summary_src_off, env->compact.summary_bytes,
                               hipMemcpyDeviceToDevice, env->stream));
    return RC_OK;
}

Rccl g_rccl;
std::string g_rccl_path;
std::mutex g_rccl_mutex;

int load_rccl() {
    std::lock_guard<std::mutex> lock(g_rccl_mutex);       // handles may be set up from different threads
    if (g_rccl.handle) return RC_OK;
    void *h = nullptr;
    if (!g_rccl_path.empty()) {
        h = dlopen(g_rccl_path.c_str(), RTLD_NOW | RTLD_LOCAL);
        if (!h) return fail(RC_ERR_COMM, "dlopen(%s) failed: %s", g_rccl_path.c_str(), dlerror());
    } else {
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
        for (const char *n : names)                       // a copy the process already holds wins
            if ((h = dlopen(n, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD))) break;
        if (!h)
            for (const char *n : names)
                if ((h = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
        if (!h) return fail(RC_ERR_COMM, "RCCL not found (tried librccl.so.1, librccl.so, /opt/rocm/lib): %s", dlerror());
    }
    Rccl r;
    r.handle = h;
    r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(h, "ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))dlsym(h, "ncclCommInitRank");
    r.AllGather = (decltype(r.AllGather))dlsym(h, "ncclAllGather");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(h, "ncclCommDestroy");
    r.CommCount = (decltype(r.CommCount))dlsym(h, "ncclCommCount");
    r.GetErrorString = (decltype(r.GetErrorString))dlsym(h, "ncclGetErrorString");
    if (!r.GetUniqueId || !r.CommInitRank || !r.AllGather || !r.CommDestroy || !r.GetErrorString)
        return fail(RC_ERR_COMM, "the RCCL library lacks an expected symbol");
    g_rccl = r;
    return RC_OK;
}

#define NCCL_TRY(expr)                                                                                     \
    do {                                                                                                   \
        int _r = (expr);                                                                                   \
        if (_r != 0) return fail(RC_ERR_COMM, "%s failed: %s", #expr, g_rccl.GetErrorString(_r));          \
    } while (0)

// source pointer and size of what one gather mode sends
int gather_source(rc_env *env, int mode, const void **src, size_t *bytes) {
    switch (mode) {
    case RC_GATHER_FULL:
        if (env->shared_arena) return fail(RC_ERR_INVALID, "this handle fills a slice of a shared arena: gather the arena itself");
        *src = env->out_arena;
        *bytes = env->layout.slab_bytes;
        return RC_OK;
    case RC_GATHER_SUMMARY:
        if (env->shared_arena) return fail(RC_ERR_INVALID, "this handle fills a slice of a shared arena: gather the arena itself");
        *src = (const char *)env->out_arena + env->compact.summary_src_off;
        *bytes = env->compact.summary_bytes;
        return RC_OK;
    case RC_GATHER_FULL_U16:
        if (!env->compact_slab) return fail(RC_ERR_INVALID, "RC_GATHER_FULL_U16 needs rc_set_compact_slab first");
        *src = env->compact_slab;
        *bytes = env->compact.total;
        return RC_OK;
    }
    return fail(RC_ERR_INVALID, "unknown gather mode %d", mode);
}

int check_cfg(const rc_config *cfg) {
    if (!cfg) return fail(RC_ERR_INVALID, "rc_config is NULL");
    if (cfg->struct_size != sizeof(rc_config))
        return fail(RC_ERR_INVALID, "rc_config.struct_size %u != %zu (ABI mismatch)", cfg->struct_size, sizeof(rc_config));
    if (cfg->num_envs < 1) return fail(RC_ERR_INVALID, "num_envs must be >= 1 (got %d)", cfg->num_envs);
    if (cfg->cars_per_env < 1 || cfg->cars_per_env > RC_MAX_CARS)
        return fail(RC_ERR_INVALID, "cars_per_env must be in 1..%d (got %d)", RC_MAX_CARS, cfg->cars_per_env);
    if ((int64_t)cfg->num_envs * cfg->cars_per_env * RC_N_BEAMS > 0x7fffffffLL)
        return fail(RC_ERR_INVALID, "num_envs * cars_per_env * 1080 must fit int32");
    if (cfg->arena_total_cars != 0) {
        const int64_t own = (int64_t)cfg->num_envs * cfg->cars_per_env;
        if (cfg->arena_first_car < 0 || cfg->arena_total_cars < 0 || cfg->arena_first_car + own > cfg->arena_total_cars)
            return fail(RC_ERR_INVALID, "cars [%d, %lld) do not lie inside an arena of %d cars", cfg->arena_first_car,
                        (long long)(cfg->arena_first_car + own), cfg->arena_total_cars);
        if (!cfg->external_arena) return fail(RC_ERR_INVALID, "arena_total_cars needs the caller's arena (external_arena)");
    } else if (cfg->arena_first_car != 0) {
        return fail(RC_ERR_INVALID, "arena_first_car without arena_total_cars");
    }
    if (cfg->obs_type != RC_OBS_LIDAR && cfg->obs_type != RC_OBS_LIDAR_OCCUPANCY && cfg->obs_type != RC_OBS_LIDAR_OCCUPANCY_REFERENCE)
        return fail(RC_ERR_INVALID, "unknown obs_type %d", cfg->obs_type);
    if (cfg->lidar_transform < RC_LIDAR_METRES || cfg->lidar_transform > RC_LIDAR_UNIT)
        return fail(RC_ERR_INVALID, "unknown lidar_transform %d", cfg->lidar_transform);
    if (cfg->task < RC_TASK_MAX_PROGRESS || cfg->task > RC_TASK_N_STEP_PROGRESS)
        return fail(RC_ERR_INVALID, "unknown task %d", cfg->task);
    for (int a = 0; a < RC_MAX_CARS; ++a)
        if (cfg->car_task[a] < -1 || cfg->car_task[a] > RC_TASK_N_STEP_PROGRESS)
            return fail(RC_ERR_INVALID, "unknown car_task[%d] = %d", a, cfg->car_task[a]);
    if (cfg->n_steps < 1 || cfg->n_steps > RC_NSTEP_MAX)
        return fail(RC_ERR_INVALID, "n_steps must be in 1..%d (got %d)", RC_NSTEP_MAX, cfg->n_steps);
    return RC_OK;
}

}  // namespace

namespace {
void make_tables(float *beams, float *foot) {
    // evaluated in double and rounded once, as spec.beam_table()/footprint_table() and the oracle do
    const double kPi = 3.14159265358979323846;
    const double half = (270.0 * kPi / 180.0) / 2.0;
    for (int i = 0; i < RC_N_BEAMS; ++i) {
        const double ang = half - (double)i * (2.0 * half / (RC_N_BEAMS - 1));
        beams[2 * i] = (float)std::cos(ang);
        beams[2 * i + 1] = (float)std::sin(ang);
    }
    const double xr = -0.10, xf = 0.45, hw = 0.15;
    int k = 0;
    auto lin = [](double a, double b, int n, int i) {   // numpy.linspace
        const double step = (b - a) / (n - 1);
        return i == n - 1 ? b : a + i * step;
    };
    for (int i = 0; i < 12; ++i) { foot[2 * k] = (float)lin(xr, xf, 12, i); foot[2 * k + 1] = (float)-hw; ++k; }
    for (int i = 0; i < 12; ++i) { foot[2 * k] = (float)lin(xr, xf, 12, i); foot[2 * k + 1] = (float)hw; ++k; }
    for (int i = 1; i < 6; ++i) { foot[2 * k] = (float)xr; foot[2 * k + 1] = (float)lin(-hw, hw, 7, i); ++k; }
    for (int i = 1; i < 6; ++i) { foot[2 * k] = (float)xf; foot[2 * k + 1] = (float)lin(-hw, hw, 7, i); ++k; }
}
}  // namespace

extern "C" {

int rc_selftest_reciprocal(int32_t device, uint64_t *n_checked, uint64_t *n_mismatch) {
    if (!n_checked || !n_mismatch) return fail(RC_ERR_INVALID, "NULL output pointer");
    HIP_TRY(hipSetDevice(device));
    unsigned long long *dev = nullptr, host = 0;
    HIP_TRY(hipMalloc((void **)&dev, sizeof(host)));
    hipError_t e = hipMemset(dev, 0, sizeof(host));
    const uint32_t exp_lo = 127 - 100, exp_hi = 127 + 100;
    if (e == hipSuccess) e = rck_launch_selftest_rcp(exp_lo, exp_hi, dev, nullptr);
    if (e == hipSuccess) e = hipMemcpy(&host, dev, sizeof(host), hipMemcpyDeviceToHost);
    (void)hipFree(dev);
    if (e != hipSuccess) return fail(RC_ERR_HIP, "rc_selftest_reciprocal: %s", hipGetErrorString(e));
    *n_checked = 2ull * ((uint64_t)(exp_hi - exp_lo + 1) << 23);
    *n_mismatch = host;
    return RC_OK;
}

int rc_selftest_sqrt(int32_t device, uint64_t *n_checked, uint64_t *n_mismatch) {
    if (!n_checked || !n_mismatch) return fail(RC_ERR_INVALID, "NULL output pointer");
    HIP_TRY(hipSetDevice(device));
    unsigned long long *dev = nullptr, host = 0;
    HIP_TRY(hipMalloc((void **)&dev, sizeof(host)));
    hipError_t e = hipMemset(dev, 0, sizeof(host));
    // every positive binary32 from 2^-60 to 2^10: far beyond the arguments the reference follow-the-gap agent's arccos takes it for
    // ((1 - |x|) / 2 in [0, 0.25]); zero is checked with them
    const uint32_t lo_bits = (127u - 60u) << 23, hi_bits = ((127u + 10u) << 23) - 1u;
    if (e == hipSuccess) e = rck_launch_selftest_sqrt(lo_bits, hi_bits, dev, nullptr);
    if (e == hipSuccess) e = rck_launch_selftest_sqrt(0u, 0u, dev, nullptr);
    if (e == hipSuccess) e = hipMemcpy(&host, dev, sizeof(host), hipMemcpyDeviceToHost);
    (void)hipFree(dev);
    if (e != hipSuccess) return fail(RC_ERR_HIP, "rc_selftest_sqrt: %s", hipGetErrorString(e));
    *n_checked = (uint64_t)(hi_bits - lo_bits + 1u) + 1u;
    *n_mismatch = host;
    return RC_OK;
}

int rc_selftest_div6(int32_t device, uint64_t *n_checked, uint64_t *n_mismatch) {
    if (!n_checked || !n_mismatch) return fail(RC_ERR_INVALID, "NULL output pointer");
    HIP_TRY(hipSetDevice(device));
    unsigned long long *dev = nullptr, host = 0;
    HIP_TRY(hipMalloc((void **)&dev, sizeof(host)));
    hipError_t e = hipMemset(dev, 0, sizeof(host));
    const int blocks = 4096, threads = 256, per_lane = 4096;          // 2^32 operands
    if (e == hipSuccess) e = rck_launch_selftest_div6(blocks, threads, per_lane, dev, nullptr);
    if (e == hipSuccess) e = hipMemcpy(&host, dev, sizeof(host), hipMemcpyDeviceToHost);
    (void)hipFree(dev);
    if (e != hipSuccess) return fail(RC_ERR_HIP, "rc_selftest_div6: %s", hipGetErrorString(e));
    *n_checked = (uint64_t)blocks * threads * per_lane;
    *n_mismatch = host;
    return RC_OK;
}

int rc_selftest_exact_estimate(rc_env *env, uint64_t out[4]) {
    if (!env || !out) return fail(RC_ERR_INVALID, "NULL argument");
    if (env->cfg.obs_type != RC_OBS_LIDAR_OCCUPANCY_REFERENCE)
        return fail(RC_ERR_INVALID, "rc_selftest_exact_estimate is for obs_type lidar_occupancy_reference (this handle: %d)", env->cfg.obs_type);
    HIP_TRY(hipSetDevice(env->cfg.device));
    unsigned long long *dev = nullptr, host[4] = {0, 0, 0, 0};
    HIP_TRY(hipMalloc((void **)&dev, sizeof(host)));
    hipError_t e = hipMemsetAsync(dev, 0, sizeof(host), env->stream);
    int rc = RC_OK;
    if (e == hipSuccess) {
        env->exact.check = dev;
        rc = render_reference_patches(env);                    // the patches themselves come out as ever
        env->exact.check = nullptr;
        e = hipStreamSynchronize(env->stream);
    }
    if (e == hipSuccess) e = hipMemcpy(host, dev, sizeof(host), hipMemcpyDeviceToHost);
    (void)hipFree(dev);
    if (rc != RC_OK) return rc;
    if (e != hipSuccess) return fail(RC_ERR_HIP, "rc_selftest_exact_estimate: %s", hipGetErrorString(e));
    for (int k = 0; k < 4; ++k) out[k] = host[k];
    return RC_OK;
}

void rc_spec_tables(float *beams_1080x2, float *footprint_34x2) {
    if (beams_1080x2 && footprint_34x2) make_tables(beams_1080x2, footprint_34x2);
}

const char *rc_last_error(void) { return g_last_error.c_str(); }
int rc_abi_version(void) { return RC_ABI_VERSION; }

// The identity of this build: racing_dreamer_amd/build.py hashes the flags and the contents of every source and header
// and passes the result as -DRC_BUILD_ID; the marker in front lets the build find the id in the file without loading it.
#ifndef RC_BUILD_ID
#define RC_BUILD_ID "unidentified (built without racing_dreamer_amd/build.py)"
#endif
extern "C" __attribute__((used)) const char rc_build_id_string[] = "RC_BUILD_ID=" RC_BUILD_ID;
const char *rc_build_id(void) { return rc_build_id_string + 12; }

void rc_default_config(rc_config *cfg) {
    if (!cfg) return;
    std::memset(cfg, 0, sizeof(*cfg));
    cfg->struct_size = sizeof(rc_config);
    cfg->num_envs = 1;
    cfg->cars_per_env = 1;
    cfg->obs_type = RC_OBS_LIDAR;
    cfg->task = RC_TASK_MAX_PROGRESS;
    cfg->laps = 10;                      // dreamer/scenarios/max_progress/columbia.yml:10
    cfg->time_limit = 180.0f;
    cfg->terminate_on_collision = 1;
    cfg->collision_reward = -1.0f;
    cfg->remap_actions = 0;
    cfg->action_low[0] = 0.005f;         // dreamer/dream.py:138
    cfg->action_low[1] = -1.0f;
    cfg->action_high[0] = 1.0f;
    cfg->action_high[1] = 1.0f;
    for (int a = 0; a < RC_MAX_CARS; ++a) cfg->car_task[a] = -1;      // every car runs `task`
    cfg->n_steps = 10;                   // baselines/scenarios/max_progress/columbia.yml:18
    cfg->arena_total_cars = 0;           // the arena is this handle's alone
    cfg->arena_first_car = 0;
}

size_t rc_arena_bytes(const rc_config *cfg) {
    if (!cfg || cfg->num_envs < 1 || cfg->cars_per_env < 1) return 0;
    const int n = cfg->arena_total_cars > 0 ? cfg->arena_total_cars : cfg->num_envs * cfg->cars_per_env;
    return make_layout(n, cfg->obs_type != RC_OBS_LIDAR).total;
}

int rc_field_layout(const rc_config *cfg, int32_t field, size_t *section_offset, size_t *bytes_per_car) {
    if (!cfg || !section_offset || !bytes_per_car) return fail(RC_ERR_INVALID, "NULL argument");
    if (field < 0 || field >= RC_F_COUNT) return fail(RC_ERR_INVALID, "unknown field %d", field);
    if (cfg->num_envs < 1 || cfg->cars_per_env < 1) return fail(RC_ERR_INVALID, "num_envs and cars_per_env must be >= 1");
    const int n = cfg->arena_total_cars > 0 ? cfg->arena_total_cars : cfg->num_envs * cfg->cars_per_env;
    const Layout l = make_layout(n, cfg->obs_type != RC_OBS_LIDAR);
    *section_offset = l.offset[field];
    *bytes_per_car = l.bytes[field] ? kFieldBytes[field] : 0;
    return RC_OK;
}

int rc_create(const rc_config *cfg, rc_env **out) {
    if (!out) return fail(RC_ERR_INVALID, "out is NULL");
    *out = nullptr;
    int rc = check_cfg(cfg);
    if (rc) return rc;
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (cfg->device < 0 || cfg->device >= ndev)
        return fail(RC_ERR_INVALID, "device %d out of range (%d HIP devices visible)", cfg->device, ndev);
    HIP_TRY(hipSetDevice(cfg->device));
    rc_env *env = new (std::nothrow) rc_env();
    if (!env) return fail(RC_ERR_NOMEM, "out of host memory");
    env->cfg = *cfg;
    const int n = env->n_cars = cfg->num_envs * cfg->cars_per_env;
    const bool occ = cfg->obs_type != RC_OBS_LIDAR;           // both renders fill the OCCUPANCY section
    env->shared_arena = cfg->arena_total_cars > 0 && cfg->arena_total_cars != n;
    env->layout = cfg->arena_total_cars > 0 ? make_layout(cfg->arena_total_cars, occ, cfg->arena_first_car, n) : make_layout(n, occ);
    env->compact = make_compact(env->layout, n);
#define FAIL_FREE(code_expr) do { int _c = (code_expr); rc_destroy(env); return _c; } while (0)
#define HIP_TRY_FREE(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) FAIL_FREE(fail(RC_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(_e))); } while (0)
    if (cfg->stream) {
        env->stream = (hipStream_t)cfg->stream;
    } else {
        HIP_TRY_FREE(hipStreamCreateWithFlags(&env->stream, hipStreamNonBlocking));
        env->own_stream = true;
    }
    if (cfg->external_arena) {
        if (cfg->external_arena_bytes < env->layout.total)
            FAIL_FREE(fail(RC_ERR_INVALID, "external arena too small: %zu < %zu", cfg->external_arena_bytes, env->layout.total));
        if ((uintptr_t)cfg->external_arena % 64)
            FAIL_FREE(fail(RC_ERR_INVALID, "external arena must be 64-byte aligned"));
        env->arena = cfg->external_arena;
    } else {
        HIP_TRY_FREE(hipMalloc(&env->arena, env->layout.total));
        env->own_arena = true;
    }
    if (!env->shared_arena) {
        HIP_TRY_FREE(hipMemsetAsync(env->arena, 0, env->layout.total, env->stream));
    } else {
        for (int f = 0; f < RC_F_COUNT; ++f)          // this handle's slices only: the rest belongs to other handles
            if (env->layout.bytes[f])
                HIP_TRY_FREE(hipMemsetAsync((char *)env->arena + env->layout.offset[f], 0, env->layout.bytes[f], env->stream));
    }

    // simulator state: packed scan pose, 10 float + 2 int + 6 byte arrays per car, 2 int + 1 uint per env
    const size_t nc = (size_t)align_up(n, 64), ne = (size_t)align_up(cfg->num_envs, 64);
    bool any_nstep = false;
    for (int a = 0; a < RC_MAX_CARS; ++a) {
        env->params.car_task[a] = cfg->car_task[a] < 0 ? cfg->task : cfg->car_task[a];
        any_nstep |= a < cfg->cars_per_env && env->params.car_task[a] == RC_TASK_N_STEP_PROGRESS;
    }
    env->params.n_steps = cfg->n_steps;
    const size_t state_bytes = nc * (10 * 4 + 2 * 4 + 6 + 16 + 16) + ne * 12 + (any_nstep ? nc * RC_NSTEP_MAX * 4 : 0) + 64;
    HIP_TRY_FREE(hipMalloc(&env->state_mem, state_bytes));
    HIP_TRY_FREE(hipMemsetAsync(env->state_mem, 0, state_bytes, env->stream));
    HIP_TRY_FREE(hipMalloc((void **)&env->mask_dev, ne));
    {
        char *m = (char *)env->state_mem;
        RcStateDev &s = env->params.st;
        s.scan_pose = (float4 *)m; m += nc * 16;                // first: 16-byte aligned
        s.patch_pose = (int4 *)m; m += nc * 16;
        float **fp[] = {&s.x, &s.y, &s.theta, &s.ct, &s.st, &s.v, &s.delta, &s.omega, &s.accel, &s.progress};
        for (float **f : fp) { *f = (float *)m; m += nc * 4; }
        s.lap = (int32_t *)m; m += nc * 4;
        s.cp = (int32_t *)m; m += nc * 4;
        s.steps = (int32_t *)m; m += ne * 4;
        s.agent_steps = (int32_t *)m; m += ne * 4;
        s.episode = (uint32_t *)m; m += ne * 4;
        uint8_t **bp[] = {&s.wall, &s.opp, &s.wrong, &s.done, &s.trunc, &s.fresh};
        for (uint8_t **b : bp) { *b = (uint8_t *)m; m += nc; }
        s.nstep_hist = any_nstep ? (float *)m : nullptr;      // (nc * (48 + 6) + ne * 12 bytes in: 4-byte aligned)
        if (any_nstep) m += nc * RC_NSTEP_MAX * 4;
        env->params.scan_overrun = (uint32_t *)m;             // (zeroed with the rest)
    }
    if (n >= RC_ORDER_COST_MIN_CARS) {     // the scan takes the cars in track order / longest first (sorted every RC_ORDER_PERIOD observations)
        HIP_TRY_FREE(hipMalloc(&env->order_mem, (size_t)n * 8 + RC_ORDER_BUCKETS * 4));
        env->params.st.order = nullptr;    // (identity until the first sort: set in sort_cars_if_due)
    }
    bind_outputs(env, env->arena);
    env->actions_in = (float *)((char *)env->arena + env->layout.offset[RC_F_ACTION_IN]);
    RcParams &p = env->params;
    p.num_envs = cfg->num_envs;
    p.cars_per_env = cfg->cars_per_env;
    p.n_cars = n;
    p.first_env = (uint32_t)(uint64_t)cfg->first_env;
    p.task = cfg->task;
    p.laps = cfg->laps;
    p.terminate_on_collision = cfg->terminate_on_collision;
    p.remap_actions = cfg->remap_actions;
    p.time_limit_steps = cfg->time_limit_steps;
    p.auto_reset = cfg->auto_reset;
    p.render_patch = cfg->obs_type == RC_OBS_LIDAR_OCCUPANCY ? 1 : 0;      // (the fast sampler's per-car header; the reference render reads the state itself)
    p.lidar_transform = cfg->lidar_transform;
    p.time_limit = cfg->time_limit;
    p.collision_reward = cfg->collision_reward;
    p.act_lo0 = cfg->action_low[0];
    p.act_lo1 = cfg->action_low[1];
    p.act_hi0 = cfg->action_high[0];
    p.act_hi1 = cfg->action_high[1];
    hipDeviceProp_t prop;
    HIP_TRY_FREE(hipGetDeviceProperties(&prop, cfg->device));
    env->launch.n_cu = prop.multiProcessorCount;
    *out = env;
    return RC_OK;
}

static void p2p_free(rc_env *env);

void rc_destroy(rc_env *env) {
    if (!env) return;
    (void)hipSetDevice(env->cfg.device);
    if (env->stream) (void)hipStreamSynchronize(env->stream);
    p2p_free(env);
    if (env->comm_stream) (void)hipStreamSynchronize(env->comm_stream);
    if (env->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(env->comm);
    if (env->ev_ready) (void)hipEventDestroy(env->ev_ready);
    if (env->ev_gathered) (void)hipEventDestroy(env->ev_gathered);
    if (env->comm_stream) (void)hipStreamDestroy(env->comm_stream);
    for (EventPair &ep : env->pending) { (void)hipEventDestroy(ep.a); (void)hipEventDestroy(ep.b); }
    for (EventPair &ep : env->free_events) { (void)hipEventDestroy(ep.a); (void)hipEventDestroy(ep.b); }
    if (env->own_arena && env->arena) (void)hipFree(env->arena);
    if (env->state_mem) (void)hipFree(env->state_mem);
    if (env->ftg_prev) (void)hipFree(env->ftg_prev);
    if (env->exact_mem) (void)hipFree(env->exact_mem);
    if (env->order_mem) (void)hipFree(env->order_mem);
    if (env->group_dev) (void)hipFree(env->group_dev);
    if (env->group_host) (void)hipHostFree(env->group_host);
    for (hipEvent_t e : env->group_ev) if (e) (void)hipEventDestroy(e);
    env->track.reset();
    if (env->mask_dev) (void)hipFree(env->mask_dev);
    if (env->own_stream && env->stream) (void)hipStreamDestroy(env->stream);
    delete env;
}

namespace {
// Pillow's precompute_coeffs + normalize_coeffs_8bpc (src/libImaging/Resample.c) for 200 -> 64 pixels with the bicubic filter
// (a = -0.5): oracle/patch_reference.py resize_coefficients, oracle/racecar_oracle.c oc_resize_coefficients - the same doubles.
double px_bicubic(double x) {
    const double a = -0.5;
    if (x < 0.0) x = -x;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
    return 0.0;
}
void px_resize_tables(int32_t *tab) {          // [64][15] coefficients, then [64][2] bounds
    const int in = 200, out = 64, ksize = 15, bits = 22;
    const double scale = (double)in / out, filterscale = scale, support = 2.0 * filterscale, ss = 1.0 / filterscale;
    for (int xx = 0; xx < out; ++xx) {
        const double center = 0 + (xx + 0.5) * scale;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in) xmax = in;
        xmax -= xmin;
        double k[15], ww = 0.0;
        for (int x = 0; x < xmax; ++x) { k[x] = px_bicubic((x + xmin - center + 0.5) * ss); ww += k[x]; }
        for (int x = 0; x < ksize; ++x) {
            double w = x < xmax ? k[x] : 0.0;
            if (x < xmax && ww != 0.0) w /= ww;
            tab[xx * ksize + x] = w < 0 ? (int32_t)(-0.5 + w * (1 << bits)) : (int32_t)(0.5 + w * (1 << bits));
        }
        tab[out * ksize + 2 * xx] = xmin;
        tab[out * ksize + 2 * xx + 1] = xmax;
    }
}
}  // namespace

int rc_set_source_frame(rc_env *env, int32_t full_height, int32_t row_top, int32_t col0, double origin_x, double origin_y,
                        double resolution) {
    if (!env) return fail(RC_ERR_INVALID, "env is NULL");
    if (env->cfg.obs_type != RC_OBS_LIDAR_OCCUPANCY_REFERENCE)
        return fail(RC_ERR_INVALID, "rc_set_source_frame is for obs_type lidar_occupancy_reference (this handle: %d)", env->cfg.obs_type);
    if (full_height < 1 || !(resolution > 0.0)) return fail(RC_ERR_INVALID, "bad source frame: height %d, resolution %g", full_height, resolution);
    HIP_TRY(hipSetDevice(env->cfg.device));
    if (!env->exact_mem) {
        // a chunk of cars in flight: 387 200 B of spline coefficients each.  RC_EXACT_CHUNK_CARS = 24 per CU: the prefilter runs one
        // workgroup per CU, the sampling six - whole rounds for both
        env->exact_chunk = env->n_cars < RC_EXACT_CHUNK_CARS ? env->n_cars : RC_EXACT_CHUNK_CARS;
        const size_t tab_bytes = (size_t)RC_EXACT_TABLE_INTS * 4, scratch = (size_t)env->exact_chunk * RC_EXACT_CAR_DOUBLES * 8;
        HIP_TRY(hipMalloc(&env->exact_mem, scratch + align_up(tab_bytes, 64)));
        int32_t tab[RC_EXACT_TABLE_INTS];
        px_resize_tables(tab);
        HIP_TRY(hipMemcpy((char *)env->exact_mem + scratch, tab, tab_bytes, hipMemcpyHostToDevice));
        env->exact.scratch = (double *)env->exact_mem;
        env->exact.kk = (const int32_t *)((char *)env->exact_mem + scratch);
    }
    RcExactParams &x = env->exact;
    x.fh = full_height; x.r_top = row_top; x.c0 = col0;
    x.ox = origin_x; x.oy = origin_y; x.res = resolution;
    x.n_cars = env->n_cars; x.car0 = 0;
    env->has_frame = true;
    return RC_OK;
}

int rc_load_track(rc_env *env, const uint32_t *occ_words, const uint32_t *drivable_words, const float *progress,
                  int32_t h, int32_t w, int32_t pitch, float resolution, float origin_x, float origin_y,
                  const float *centerline, int32_t n_centerline) {
    if (!env) return fail(RC_ERR_INVALID, "env is NULL");
    if (!occ_words || !drivable_words || !progress || !centerline) return fail(RC_ERR_INVALID, "NULL track array");
    if (h < 3 || w < 3 || pitch * 32 < w) return fail(RC_ERR_INVALID, "bad track shape h=%d w=%d pitch=%d", h, w, pitch);
    if (n_centerline < 1) return fail(RC_ERR_INVALID, "centerline table is empty");
    if (!(resolution > 0.f)) return fail(RC_ERR_INVALID, "resolution must be > 0");
    // the skipping traversals place a ray inside a free rectangle with fp32 arithmetic on cell coordinates and fall
    // back to exact comparisons within max(w, h) * 2^-21 cell of a boundary (RcTrackDev::band, derived in
    // racecar_kernels.hip); 4096 keeps that zone below 2e-3 cell and every index within the 24-bit multiplies
    if (h > 4096 || w > 4096) return fail(RC_ERR_INVALID, "grids larger than 4096 cells per side are not supported (h=%d w=%d)", h, w);
    HIP_TRY(hipSetDevice(env->cfg.device));
    const size_t nwords = (size_t)h * pitch;
    // the same track already on this device (another handle of the process loaded it)?
    uint64_t key = 0xcbf29ce484222325ull, sum2 = 0;
    {
        const int32_t dims[4] = {h, w, pitch, n_centerline};
        const float geo[3] = {resolution, origin_x, origin_y};
        key = fnv1a(key, dims, sizeof(dims));
        key = fnv1a(key, geo, sizeof(geo));
        key = fnv1a(key, occ_words, nwords * 4);
        key = fnv1a(key, drivable_words, nwords * 4);
        key = fnv1a(key, progress, (size_t)h * w * 4);
        key = fnv1a(key, centerline, (size_t)n_centerline * 16);
        sum2 = wordsum(wordsum(wordsum(wordsum(0, occ_words, nwords * 4), drivable_words, nwords * 4), progress, (size_t)h * w * 4),
                       centerline, (size_t)n_centerline * 16);
    }
    std::lock_guard<std::mutex> track_lock(g_track_mutex);
    auto finish_load = [&](const std::shared_ptr<TrackTables> &tt) {
        env->track = tt;
        env->params.trk = tt->t;
        set_band(env);
        RcLaunchInfo &li = env->launch;
        li.lds_bytes = tt->lds_bytes;
        li.lds_bytes_skip = tt->lds_bytes_skip;
        li.lds_bytes_packed = tt->lds_bytes_packed;
        li.raycast_variant = 7;         // per-cell, per-quadrant free rectangles, one wave per car
        li.car_threads = 64;
        li.car_split = 1;
        li.ray_threads = 1024;
        env->has_track = true;
        set_launch_geometry(env);
        env->was_reset = false;
    };
    for (auto it = g_track_cache.begin(); it != g_track_cache.end();)       // entries whose tables are gone
        it = it->second.expired() ? g_track_cache.erase(it) : std::next(it);
    {
        auto it = g_track_cache.find({env->cfg.device, key});
        if (it != g_track_cache.end()) {
            std::shared_ptr<TrackTables> tt = it->second.lock();
            const bool same = tt && tt->h == h && tt->w == w && tt->pitch == pitch && tt->n_centerline == n_centerline &&
                              tt->res == resolution && tt->ox == origin_x && tt->oy == origin_y && tt->sum2 == sum2;
            if (same) {
                HIP_TRY(hipStreamSynchronize(env->stream));       // nothing of this handle still reads its old track
                finish_load(tt);
                return RC_OK;
            }
            // (a different track under the same 64-bit key: build its tables; the map keeps the newer one)
        }
    }
    const size_t bm_bytes = align_up(nwords * 4 + 4, 64);   // at least one all-zero word behind the bitmap (rc_patch_car_kernel)
    // The lidar_occupancy render and the scan's early forms (variants 0-3) keep the whole bitmap in the 160 KiB LDS;
    // the default scan does not, so a larger map is fine as long as the patch is not asked for.
    const bool fits_lds = bm_bytes <= 160 * 1024;
    if (!fits_lds && env->params.render_patch)
        return fail(RC_ERR_INVALID, "track bitmap %zu B does not fit the 160 KiB LDS (needed for obs_type lidar_occupancy)", bm_bytes);
    // occupancy with the sentinel ring set
    std::vector<uint32_t> ray(bm_bytes / 4, 0u), drv(bm_bytes / 4, 0u);
    std::memcpy(ray.data(), occ_words, nwords * 4);
    std::memcpy(drv.data(), drivable_words, nwords * 4);
    auto setbit = [&](int ix, int iy) { ray[(size_t)iy * pitch + (ix >> 5)] |= 1u << (ix & 31); };
    // ... and the drivable bitmap's outermost ring cleared: the lidar_occupancy render clamps out-of-grid taps onto it
    // (env spec: "the outermost ring of cells is not drivable"; every compiled track keeps a 16-cell margin anyway)
    auto clrbit = [&](int ix, int iy) { drv[(size_t)iy * pitch + (ix >> 5)] &= ~(1u << (ix & 31)); };
    for (int ix = 0; ix < w; ++ix) { setbit(ix, 0); setbit(ix, h - 1); clrbit(ix, 0); clrbit(ix, h - 1); }
    for (int iy = 0; iy < h; ++iy) { setbit(0, iy); setbit(w - 1, iy); clrbit(0, iy); clrbit(w - 1, iy); }
    std::vector<float> beams(((RC_N_BEAMS + 63) / 64) * 64 * 2, 0.0f), foot(RCS_N_FOOTPRINT * 2);   // beams padded to whole waves
    make_tables(beams.data(), foot.data());
    // the one-wave-per-car scan relies on no beam being exactly axis-parallel in the sensor frame (racecar_kernels.hip)
    for (int i = 0; i < 2 * RC_N_BEAMS; ++i)
        if (!(std::fabs(beams[i]) >= 1e-4f)) return fail(RC_ERR_INVALID, "beam table holds a zero component");
    // Free-block table for the skipping traversal: exact chessboard distance transform of the stop cells
    // (two raster passes), then the minimum over each block.  A block value v >= 1 certifies that every
    // cell within Chebyshev distance v - 1 of any cell of the block is free.
    std::vector<int32_t> dist((size_t)h * w);
    for (int iy = 0; iy < h; ++iy)
        for (int ix = 0; ix < w; ++ix)
            dist[(size_t)iy * w + ix] = ((ray[(size_t)iy * pitch + (ix >> 5)] >> (ix & 31)) & 1u) ? 0 : (1 << 20);
    auto relax = [&](int iy, int ix, int oy, int ox) {
        const int y = iy + oy, x = ix + ox;
        if (y < 0 || y >= h || x < 0 || x >= w) return;
        int32_t &d = dist[(size_t)iy * w + ix];
        const int32_t c = dist[(size_t)y * w + x] + 1;
        if (c < d) d = c;
    };
    for (int iy = 0; iy < h; ++iy)
        for (int ix = 0; ix < w; ++ix) { relax(iy, ix, -1, -1); relax(iy, ix, -1, 0); relax(iy, ix, -1, 1); relax(iy, ix, 0, -1); }
    for (int iy = h - 1; iy >= 0; --iy)
        for (int ix = w - 1; ix >= 0; --ix) { relax(iy, ix, 1, 1); relax(iy, ix, 1, 0); relax(iy, ix, 1, -1); relax(iy, ix, 0, 1); }
    int blk_shift = 2;
    auto blk_dim = [&](int n) { return (n + (1 << blk_shift) - 1) >> blk_shift; };
    if (bm_bytes + align_up((size_t)blk_dim(h) * blk_dim(w), 64) > 160 * 1024) blk_shift = 3;
    const int blk_w = blk_dim(w), blk_h = blk_dim(h), bs = 1 << blk_shift;
    const size_t blk_bytes = align_up((size_t)blk_w * blk_h, 64);
    std::vector<uint8_t> blocks(blk_bytes, 0);
    for (int by = 0; by < blk_h; ++by)
        for (int bx = 0; bx < blk_w; ++bx) {
            int32_t m = 255;
            for (int oy = 0; oy < bs; ++oy)
                for (int ox = 0; ox < bs; ++ox) {
                    const int y = by * bs + oy, x = bx * bs + ox;
                    const int32_t d = (y < h && x < w) ? dist[(size_t)y * w + x] : 0;
                    if (d < m) m = d;
                }
            blocks[(size_t)by * blk_w + bx] = (uint8_t)m;
        }
    // packed table for variant 3: one uint32 per 4x4 block = [value << 16 | 16 occupancy bits]
    const int pk_w = (w + 3) >> 2, pk_h = (h + 3) >> 2;
    const size_t packed_bytes = align_up((size_t)pk_w * pk_h * 4, 64);
    std::vector<uint32_t> packed(packed_bytes / 4, 0u);
    for (int by = 0; by < pk_h; ++by)
        for (int bx = 0; bx < pk_w; ++bx) {
            uint32_t occ16 = 0;
            int32_t m = 255;
            for (int oy = 0; oy < 4; ++oy)
                for (int ox = 0; ox < 4; ++ox) {
                    const int y = by * 4 + oy, x = bx * 4 + ox;
                    const bool in = y < h && x < w;
                    const int32_t d = in ? dist[(size_t)y * w + x] : 0;
                    if (d < m) m = d;
                    if (!in || d == 0) occ16 |= 1u << (oy * 4 + ox);
                }
            packed[(size_t)by * pk_w + bx] = ((uint32_t)m << 16) | occ16;
        }
    // per-cell table for variant 5
    const int cell_pitch = (w + 3) & ~3;
    const size_t cell_bytes = align_up((size_t)cell_pitch * h, 64);
    std::vector<uint8_t> cells(cell_bytes, 0);
    for (int iy = 0; iy < h; ++iy)
        for (int ix = 0; ix < w; ++ix) cells[(size_t)iy * cell_pitch + ix] = (uint8_t)std::min<int32_t>(dist[(size_t)iy * w + ix], 255);
    // per-cell, per-quadrant free rectangles for variant 6.  A ray in cell (ix, iy) heading into quadrant
    // (sx, sy) only ever visits cells with (x - ix) * sx >= 0 and (y - iy) * sy >= 0, so the certificate can be a
    // rectangle with the current cell at its corner: it reaches as far as the walls AHEAD allow, where the
    // symmetric square of variant 5 is limited by the nearest wall in any direction (a wall-grazing ray crawls).
    // Among the free rectangles (width = min over its rows of the free run towards sx) the one with the largest
    // geometric mean of exit distance for rays at 11.25, 33.75, 56.25 and 78.75 degrees inside the quadrant is kept
    // (tools/analysis/skip_stats.py quadrant: 4 % fewer trips than the arithmetic mean at 22.5 / 67.5 degrees, far fewer than squares
    // or maximal area).  The choice only affects speed: any free rectangle gives the same result (see cast_ray_skip).
    // (built on the device by rc_build_quad_kernel after the upload, like the first-trip table: 0.67 -> 0.1 s of
    // rc_load_track for austria, 2.7 -> 0.4 s for gbr)
    const size_t quad_plane_bytes = align_up((size_t)cell_pitch * h * 2, 64);
    // First-trip table of the one-wave-per-car scan (variant 7): RC_FIRST_PLANES rectangles per cell, one per
    // quadrant and bin of the ray's slope |dy / dx| - built on the device by rc_build_first_kernel right after the
    // upload (racecar_kernels.hip has the description); 512 B per cell: austria 130 MB, gbr 506 MB - sized for the
    // 288 GB of HBM, not for the L2 (a car reads one line of it per step).
    const size_t first_bytes = align_up((size_t)cell_pitch * h * RC_FIRST_PLANES * 2, 64);
    const size_t prog_bytes = align_up((size_t)h * w * 4, 64);
    const size_t cl_bytes = align_up((size_t)n_centerline * 16, 64);
    const size_t beam_bytes = align_up(beams.size() * 4, 64), foot_bytes = align_up(foot.size() * 4, 64);
    const size_t spawn_bytes = align_up((size_t)n_centerline * 32, 64);
    const size_t total = 2 * bm_bytes + prog_bytes + cl_bytes + spawn_bytes + beam_bytes + foot_bytes + blk_bytes + packed_bytes + cell_bytes + 4 * quad_plane_bytes + first_bytes;
    HIP_TRY(hipStreamSynchronize(env->stream));
    // from here on the handle has NO track until the new one is complete: a failure below (allocation, upload, validation)
    // must not leave it launching on the tables it has just given up
    env->has_track = false;
    env->was_reset = false;
    env->params.trk = RcTrackDev{};
    env->track.reset();                                    // (frees the old tables if no other handle shares them)
    std::shared_ptr<TrackTables> tt = std::make_shared<TrackTables>();
    tt->device = env->cfg.device;
    tt->h = h; tt->w = w; tt->pitch = pitch; tt->n_centerline = n_centerline;
    tt->res = resolution; tt->ox = origin_x; tt->oy = origin_y; tt->sum2 = sum2;
    HIP_TRY(hipMalloc(&tt->mem, total));
    char *m = (char *)tt->mem;
    RcTrackDev &t = tt->t;
    HIP_TRY(hipMemcpy(m, ray.data(), bm_bytes, hipMemcpyHostToDevice)); t.ray_words = (const uint32_t *)m; m += bm_bytes;
    HIP_TRY(hipMemcpy(m, drv.data(), bm_bytes, hipMemcpyHostToDevice)); t.drv_words = (const uint32_t *)m; m += bm_bytes;
    HIP_TRY(hipMemcpy(m, progress, (size_t)h * w * 4, hipMemcpyHostToDevice)); t.progress = (const float *)m; m += prog_bytes;
    HIP_TRY(hipMemcpy(m, centerline, (size_t)n_centerline * 16, hipMemcpyHostToDevice)); t.centerline = (const float *)m; m += cl_bytes;
    t.spawn = (const float4 *)m; m += spawn_bytes;            // filled below, on the device
    HIP_TRY(hipMemcpy(m, beams.data(), beams.size() * 4, hipMemcpyHostToDevice)); t.beams = (const float *)m; m += beam_bytes;
    HIP_TRY(hipMemcpy(m, foot.data(), foot.size() * 4, hipMemcpyHostToDevice)); t.footprint = (const float *)m; m += foot_bytes;
    HIP_TRY(hipMemcpy(m, blocks.data(), blk_bytes, hipMemcpyHostToDevice)); t.free_blocks = (const uint8_t *)m; m += blk_bytes;
    t.blk_w = blk_w; t.blk_h = blk_h; t.blk_shift = blk_shift; t.blk_bytes = (int32_t)blk_bytes;
    HIP_TRY(hipMemcpy(m, packed.data(), packed_bytes, hipMemcpyHostToDevice)); t.packed_blocks = (const uint32_t *)m; m += packed_bytes;
    t.packed_bytes = (int32_t)packed_bytes;
    t.packed_w = pk_w;
    HIP_TRY(hipMemcpy(m, cells.data(), cell_bytes, hipMemcpyHostToDevice)); t.cell_dist = (const uint8_t *)m; m += cell_bytes;
    t.cell_pitch = cell_pitch;
    t.quad_rect = (const uint16_t *)m; m += 4 * quad_plane_bytes;      // filled below, on the device
    t.quad_plane_bytes = (int32_t)quad_plane_bytes;
    t.first_rect = (const uint16_t *)m; m += first_bytes;      // filled below, on the device
    t.h = h; t.w = w; t.pitch = pitch; t.n_centerline = n_centerline;
    t.org_x = origin_x; t.org_y = origin_y; t.res = resolution;
    t.inv_res = 1.0f / resolution;
    t.tmax = RCS_MAX_RANGE * t.inv_res;
    t.band = t.band_mh = t.band2 = 0.0f;                   // per handle: set_band
    // launch geometry: persistent workgroups, the whole bitmap resident in each workgroup's LDS
    tt->lds_bytes = fits_lds ? bm_bytes : 0;
    tt->lds_bytes_skip = bm_bytes + blk_bytes <= 160 * 1024 ? bm_bytes + blk_bytes : 0;
    tt->lds_bytes_packed = (blk_shift == 2 && packed_bytes <= 160 * 1024) ? packed_bytes : 0;
    {   // the dynamic-LDS ceiling is a property of the kernel functions, not of a handle: only ever raise it, or a
        // small track loaded after a large one would make the large one's launches fail
        static std::map<int, size_t> lds_limit;
        size_t &lim = lds_limit[env->cfg.device];
        const size_t need = std::min<size_t>(160 * 1024, std::max(std::max(std::max(tt->lds_bytes, tt->lds_bytes_skip), tt->lds_bytes_packed),
                                                                  rc_patch_padded_bytes(h, w)));
        if (need > lim || lim == 0) {
            HIP_TRY(rck_set_lds_limits(std::max(need, lim)));
            lim = std::max(need, lim);
        }
    }
    HIP_TRY(rck_build_quad_planes(t, (uint16_t *)t.quad_rect, env->stream));
    HIP_TRY(rck_build_first_table(t, (uint16_t *)t.first_rect, env->stream));
    HIP_TRY(rck_build_spawn_table(t, (float4 *)t.spawn, env->stream));
    HIP_TRY(hipStreamSynchronize(env->stream));
    {   // the new tables under the bounded scan, from every cell a sensor can stand in: a table that sends a ray in circles
        // fails HERE, with a message, and not as a hung wave in the unbounded production loop
        unsigned long long n_scans = 0;
        unsigned n_overruns = 0;
        HIP_TRY(rck_validate_tables(t, std::ldexp((float)(std::max(w, h) + 2), -21), env->stream, &n_scans, &n_overruns));
        if (n_overruns != 0)
            return fail(RC_ERR_INVALID, "track tables failed validation: %u of %llu validation scans used up their trip budget", n_overruns, n_scans);
    }
    g_track_cache[{env->cfg.device, key}] = tt;
    finish_load(tt);
    return RC_OK;
}

int rc_reset(rc_env *env, const uint8_t *mask_or_null, int32_t mode, uint64_t seed) {
    if (!env) return fail(RC_ERR_INVALID, "env is NULL");
    if (!env->has_track) return fail(RC_ERR_NO_TRACK, "rc_load_track must be called before rc_reset");
    if (mode < RC_RESET_GRID || mode > RC_RESET_RANDOM_BALL) return fail(RC_ERR_INVALID, "unknown reset mode %d", mode);
    if (mask_or_null && !env->was_reset) return fail(RC_ERR_INVALID, "the first rc_reset must reset every env (mask = NULL)");
    HIP_TRY(hipSetDevice(env->cfg.device));
    env->params.reset_mode = mode;
    env->params.seed_lo = (uint32_t)(seed & 0xffffffffu);
    env->params.seed_hi = (uint32_t)(seed >> 32);
    const uint8_t *mask_dev = nullptr;
    if (mask_or_null) {
        HIP_TRY(hipMemcpyAsync(env->mask_dev, mask_or_null, (size_t)env->cfg.num_envs, hipMemcpyHostToDevice, env->stream));
        // the host buffer may be pageable and reused by the caller right after we return
        HIP_TRY(hipStreamSynchronize(env->stream));
        mask_dev = env->mask_dev;
    }
    TIMED(env, RC_K_RESET, rck_launch_reset(env->params, mask_dev, env->stream));
    env->was_reset = true;
    if (!mask_or_null) env->order_age = 0xffffffffu;      // every car has a new place: sort before this observation
    return observe(env);
}

int rc_step(rc_env *env, const float *actions_dev, int32_t repeat) {
    if (!env) return fail(RC_ERR_INVALID, "env is NULL");
    if (!env->has_track) return fail(RC_ERR_NO_TRACK, "rc_load_track must be called before rc_step");
    if (!env->was_reset) return fail(RC_ERR_NEEDS_RESET, "Must reset environment.");
    if (repeat < 1) return fail(RC_ERR_INVALID, "repeat must be >= 1 (got %d)", repeat);
    HIP_TRY(hipSetDevice(env->cfg.device));
    // the kernel only reads the caller's buffer (its actions pointer is written in random-action mode alone)
    float *act = actions_dev ? const_cast<float *>(actions_dev) : env->actions_in;
    const RcRandomActions none{0, 0u, 0u, 0u};
    TIMED(env, RC_K_DYNAMICS, rck_launch_dynamics(env->params, act, repeat, none, env->stream));
    return observe(env);
}

int rc_step_random(rc_env *env, uint64_t seed, uint32_t step, int32_t repeat) {
    if (!env) return fail(RC_ERR_INVALID, "env is NULL");
    if (!env->has_track) return fail(RC_ERR_NO_TRACK, "rc_load_track must be called before rc_step_random");
    if (!env->was_reset) return fail(RC_ERR_NEEDS_RESET, "Must reset environment.");
    if (repeat < 1) return fail(RC_ERR_INVALID, "repeat must be >= 1 (got %d)", repeat);
    HIP_TRY(hipSetDevice(env->cfg.device));
    const RcRandomActions ra{1, (uint32_t)(seed & 0xffffffffu), (uint32_t)(seed >> 32), step};
    TIMED(env, RC_K_DYNAMICS, rck_launch_dynamics(env->params, env->actions_in, repeat, ra, env->stream));
    return observe(env);
}

// ---- several handles, one launch per kernel --------------------------------------------------------------------------
static int group_step(rc_env **envs, int32_t n, const float *actions_dev, int32_t repeat, const RcRandomActions &ra, const char *who) {
    if (!envs || n < 1) return fail(RC_ERR_INVALID, "%s: no handles", who);
    if (n > RC_GROUP_MAX) return fail(RC_ERR_INVALID, "%s: at most %d handles in a group (got %d)", who, RC_GROUP_MAX, n);
    if (repeat < 1) return fail(RC_ERR_INVALID, "repeat must be >= 1 (got %d)", repeat);
    rc_env *lead = envs[0];
    for (int b = 0; b < n; ++b) {
        rc_env *e = envs[b];
        if (!e) return fail(RC_ERR_INVALID, "%s: handle %d is NULL", who, b);
        if (!e->has_track) return fail(RC_ERR_NO_TRACK, "rc_load_track must be called before %s (handle %d)", who, b);
        if (!e->was_reset) return fail(RC_ERR_NEEDS_RESET, "Must reset environment.");
        if (e->cfg.device != lead->cfg.device || e->stream != lead->stream)
            return fail(RC_ERR_INVALID, "%s: the handles of a group share one device and one stream (handle %d does not)", who, b);
        if (e->cfg.cars_per_env != lead->cfg.cars_per_env)
            return fail(RC_ERR_INVALID, "%s: the handles of a group have the same cars_per_env", who);
        if (e->launch.raycast_variant != 7 || e->launch.scan_guarded || e->launch.scan_stamps || e->compact_slab)
            return fail(RC_ERR_INVALID, "%s: handle %d runs a scan variant / validation build / uint16 copy that a group launch does not carry", who, b);
        for (int c = 0; c < b; ++c)
            if (envs[c] == e) return fail(RC_ERR_INVALID, "%s: handle %d appears twice", who, b);
    }
    HIP_TRY(hipSetDevice(lead->cfg.device));
    for (int b = 0; b < n; ++b) {
        // (the sort reads last step's progress: the order is a matter of locality, not of results)
        int rc_sort = sort_cars_if_due(envs[b]);
        if (rc_sort) return rc_sort;
    }
    if (!lead->group_dev) {
        HIP_TRY(hipMalloc((void **)&lead->group_dev, sizeof(RcParams) * RC_GROUP_MAX));
        HIP_TRY(hipHostMalloc((void **)&lead->group_host, sizeof(RcParams) * RC_GROUP_MAX * 4, hipHostMallocDefault));
        for (hipEvent_t &ev : lead->group_ev) HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        lead->group_n = 0;
    }
    // the device table follows the handles' parameters in stream order: a copy is queued only when something changed (an arena
    // re-pointed by rc_set_arena, a knob), from a pinned slot that is not rewritten before its copy has run
    bool changed = lead->group_n != n;
    for (int b = 0; b < n && !changed; ++b) changed = std::memcmp(&lead->group_last[b], &envs[b]->params, sizeof(RcParams)) != 0;
    if (changed) {
        const uint32_t slot = lead->group_slot++ & 3u;
        HIP_TRY(hipEventSynchronize(lead->group_ev[slot]));
        RcParams *host = lead->group_host + (size_t)slot * RC_GROUP_MAX;
        for (int b = 0; b < n; ++b) { host[b] = envs[b]->params; lead->group_last[b] = envs[b]->params; }
        HIP_TRY(hipMemcpyAsync(lead->group_dev, host, sizeof(RcParams) * n, hipMemcpyHostToDevice, lead->stream));
        HIP_TRY(hipEventRecord(lead->group_ev[slot], lead->stream));
        lead->group_n = n;
    }
    RcGroup g{};
    g.params = lead->group_dev;
    g.n = n;
    // dynamics: a wave = 64 envs of one block
    int waves = 0, cars = 0;
    for (int b = 0; b < n; ++b) {
        g.wave_start[b] = waves;
        waves += (envs[b]->cfg.num_envs + 63) / 64;
        // (the caller's actions are in arena order: block b's start where its cars start)
        g.actions[b] = (actions_dev && !ra.on) ? const_cast<float *>(actions_dev) + 2 * (size_t)cars : envs[b]->actions_in;
        cars += envs[b]->n_cars;
    }
    g.wave_start[n] = waves;
    TIMED(lead, RC_K_DYNAMICS, rck_launch_dynamics_group(g, lead->cfg.cars_per_env, repeat, ra, lead->stream));
    // scan: a wave = one car (or 1 / split of one); the split follows the group's total, as one handle of that size would
    int split = lead->dbg[RC_DBG_RAY_SPLIT];
    if (split < 1 || split > 17) split = scan_split(cars, lead->launch.n_cu);
    waves = 0;
    for (int b = 0; b < n; ++b) {
        g.wave_start[b] = waves;
        waves += envs[b]->n_cars * split;
    }
    g.wave_start[n] = waves;
    TIMED(lead, RC_K_RAYCAST, rck_launch_raycast_group(g, lead->cfg.cars_per_env, split, lead->stream));
    for (int b = 0; b < n; ++b) envs[b]->last_scan_rows = envs[b]->params.out.lidar;
    for (int b = 0; b < n; ++b) {
        if (envs[b]->params.render_patch)
            TIMED(envs[b], RC_K_PATCH, rck_launch_patch(envs[b]->params, envs[b]->launch, envs[b]->stream));
        if (envs[b]->cfg.obs_type == RC_OBS_LIDAR_OCCUPANCY_REFERENCE) {
            int rc_px = render_reference_patches(envs[b]);
            if (rc_px) return rc_px;
        }
    }
    return RC_OK;
}

int rc_step_group(rc_env **envs, int32_t n, const float *actions_dev, int32_t repeat) {
    const RcRandomActions none{0, 0u, 0u, 0u};
    return group_step(envs, n, actions_dev, repeat, none, "rc_step_group");
}

int rc_step_random_group(rc_env **envs, int32_t n, uint64_t seed, uint32_t step, int32_t repeat) {
    const RcRandomActions ra{1, (uint32_t)(seed & 0xffffffffu), (uint32_t)(seed >> 32), step};
    return group_step(envs, n, nullptr, repeat, ra, "rc_step_random_group");
}

int rc_step_host(rc_env *env, const float *actions_host, int32_t repeat) {
    if (!env) return fail(RC_ERR_INVALID, "env is NULL");
    if (!actions_host) return fail(RC_ERR_INVALID, "actions_host is NULL");
    HIP_TRY(hipSetDevice(env->cfg.device));
    HIP_TRY(hipMemcpyAsync(env->actions_in, actions_host, (size_t)env->n_cars * 8, hipMemcpyHostToDevice, env->stream));
    HIP_TRY(hipStreamSynchronize(env->stream));
    return rc_step(env, nullptr, repeat);
}

int rc_set_pose(rc_env *env, const float *xyyaw_host) {
    if (!env) return fail(RC_ERR_INVALID, "env is NULL");
    if (!xyyaw_host) return fail(RC_ERR_INVALID, "xyyaw_host is NULL");
    if (!env->has_track) return fail(RC_ERR_NO_TRACK, "rc_load_track must be called before rc_set_pose");
    if (!env->was_reset) return fail(RC_ERR_NEEDS_RESET, "Must reset environment.");
    HIP_TRY(hipSetDevice(env->cfg.device));
    // stage through the lidar section of the arena: it is rewritten by the observation pass right after
    float *staging = env->params.out.lidar;
    HIP_TRY(hipMemcpyAsync(staging, xyyaw_host, (size_t)env->n_cars * 12, hipMemcpyHostToDevice, env->stream));
    HIP_TRY(hipStreamSynchronize(env->stream));
    HIP_TRY(rck_launch_set_pose(env->params, staging, env->stream));
    return observe(env);
}

int rc_follow_the_gap(rc_env *env, float motor_straight, float motor_corner) {
    if (!env) return fail(RC_ERR_INVALID, "env is NULL");
    if (!env->was_reset) return fail(RC_ERR_NEEDS_RESET, "Must reset environment.");
    HIP_TRY(hipSetDevice(env->cfg.device));
    TIMED(env, RC_K_FTG, rck_launch_ftg(env->params, env->actions_in, motor_straight, motor_corner, env->stream));
    return RC_OK;
}

int rc_follow_the_gap_reference(rc_env *env, float dt, float *detail_dev) {
    if (!env) return fail(RC_ERR_INVALID, "env is NULL");
    if (!env->was_reset) return fail(RC_ERR_NEEDS_RESET, "Must reset environment.");
    if (!(dt > 0.f)) return fail(RC_ERR_INVALID, "dt must be > 0 (seconds per agent step)");
    if (env->cfg.lidar_transform != RC_LIDAR_METRES) return fail(RC_ERR_INVALID, "rc_follow_the_gap_reference reads the scan in metres (lidar_transform RC_LIDAR_METRES)");
    HIP_TRY(hipSetDevice(env->cfg.device));
    if (!env->ftg_prev) {
        HIP_TRY(hipMalloc((void **)&env->ftg_prev, (size_t)env->n_cars * sizeof(float)));
        HIP_TRY(hipMemsetAsync(env->ftg_prev, 0xff, (size_t)env->n_cars * sizeof(float), env->stream));      // all ones: a NaN
    }
    TIMED(env, RC_K_FTG, rck_launch_ftg_reference(env->params, env->actions_in, env->ftg_prev, dt, detail_dev, env->stream));
    return RC_OK;
}

int rc_fill_random_actions(rc_env *env, uint64_t seed, uint32_t step) {
    if (!env) return fail(RC_ERR_INVALID, "env is NULL");
    HIP_TRY(hipSetDevice(env->cfg.device));
    const uint32_t first_car = env->params.first_env * (uint32_t)env->cfg.cars_per_env;
    TIMED(env, RC_K_ACTIONS, rck_launch_random_actions(env->actions_in, env->n_cars, first_car, (uint32_t)(seed & 0xffffffffu),
                                                       (uint32_t)(seed >> 32), step, env->stream));
    return RC_OK;
}

int rc_get(rc_env *env, int32_t field, void **dev_ptr, size_t *bytes) {
    if (!env || !dev_ptr || !bytes) return fail(RC_ERR_INVALID, "NULL argument");
    if (field < 0 || field >= RC_F_COUNT) return fail(RC_ERR_INVALID, "unknown field %d", field);
    if (env->layout.bytes[field] == 0) return fail(RC_ERR_INVALID, "field %d is not enabled in this configuration", field);
    // outputs follow rc_set_arena; the action input buffer stays in the handle's own arena
    *dev_ptr = (char *)(field == RC_F_ACTION_IN ? env->arena : env->out_arena) + env->layout.offset[field];
    *bytes = env->layout.bytes[field];
    return RC_OK;
}

int rc_copy_out(rc_env *env, int32_t field, void *host_dst, size_t bytes) {
    void *src;
    size_t n;
    int rc = rc_get(env, field, &src, &n);
    if (rc) return rc;
    if (!host_dst) return fail(RC_ERR_INVALID, "host_dst is NULL");
    if (bytes != n) return fail(RC_ERR_INVALID, "field %d holds %zu bytes, caller asked for %zu", field, n, bytes);
    HIP_TRY(hipSetDevice(env->cfg.device));
    HIP_TRY(hipMemcpyAsync(host_dst, src, n, hipMemcpyDeviceToHost, env->stream));
    HIP_TRY(hipStreamSynchronize(env->stream));
    return RC_OK;
}

int rc_trajectory_slab(rc_env *env, void **dev_ptr, size_t *bytes) {
    if (!env || !dev_ptr || !bytes) return fail(RC_ERR_INVALID, "NULL argument");
    if (env->shared_arena) return fail(RC_ERR_INVALID, "this handle fills a slice of a shared arena: the slab is the head of the arena itself");
    *dev_ptr = env->out_arena;
    *bytes = env->layout.slab_bytes;
    return RC_OK;
}

size_t rc_gather_rows_bytes(rc_env *env, uint32_t field_mask, int32_t n_rows) {
    if (!env || n_rows < 1) return 0;
    size_t off = 0;
    for (int f = 0; f < RC_F_COUNT; ++f)
        if (((field_mask >> f) & 1u) && env->layout.bytes[f]) off = align_up(off + kFieldBytes[f] * (size_t)n_rows, 64);
    return off;
}

int rc_gather_rows(rc_env *env, const void *ring_base, size_t slot_bytes, const int32_t *slot_idx_dev, const int32_t *car_idx_dev,
                   int32_t n_rows, uint32_t field_mask, void *out_dev, size_t out_bytes) {
    if (!env || !ring_base || !slot_idx_dev || !car_idx_dev || !out_dev) return fail(RC_ERR_INVALID, "NULL argument");
    if (n_rows < 1) return fail(RC_ERR_INVALID, "n_rows must be >= 1");
    if (slot_bytes < env->layout.total) return fail(RC_ERR_INVALID, "slot_bytes %zu is smaller than an arena (%zu)", slot_bytes, env->layout.total);
    // (the row gather moves 16 bytes per lane: every slot must start as rc_set_arena demands of an arena)
    if (((uintptr_t)ring_base & 63u) != 0 || (slot_bytes & 63u) != 0)
        return fail(RC_ERR_INVALID, "ring_base (%p) and slot_bytes (%zu) must be multiples of 64", ring_base, slot_bytes);
    if (env->shared_arena) return fail(RC_ERR_INVALID, "rc_gather_rows works on whole arenas, not on a slice handle");
    size_t src[RC_GATHER_MAX_FIELDS], dst[RC_GATHER_MAX_FIELDS], off = 0;
    uint32_t bpc[RC_GATHER_MAX_FIELDS];
    int n = 0;
    for (int f = 0; f < RC_F_COUNT; ++f) {
        if (!((field_mask >> f) & 1u)) continue;
        if (f == RC_F_ACTION_IN || !env->layout.bytes[f]) return fail(RC_ERR_INVALID, "field %d is not part of a recorded arena in this configuration", f);
        src[n] = env->layout.offset[f]; dst[n] = off; bpc[n] = (uint32_t)kFieldBytes[f];
        off = align_up(off + kFieldBytes[f] * (size_t)n_rows, 64);
        ++n;
    }
    if (n == 0) return fail(RC_ERR_INVALID, "empty field mask");
    if (out_bytes < off) return fail(RC_ERR_INVALID, "output too small: %zu < %zu", out_bytes, off);
    HIP_TRY(hipSetDevice(env->cfg.device));
    HIP_TRY(rck_gather_rows(ring_base, slot_bytes, slot_idx_dev, car_idx_dev, n_rows, src, dst, bpc, n, out_dev, env->stream));
    return RC_OK;
}

int rc_sample_windows(rc_env *env, const void *ring_base, size_t slot_bytes, int32_t capacity, int32_t oldest, int32_t count,
                      int32_t length, int32_t n_windows, uint64_t seed, uint32_t draw, int32_t max_tries, int32_t *slot_idx_dev,
                      int32_t *slot_obs_idx_dev, int32_t *car_idx_dev, int32_t *meta_dev, uint32_t *failed_dev) {
    if (!env || !ring_base || !slot_idx_dev || !slot_obs_idx_dev || !car_idx_dev || !meta_dev || !failed_dev) return fail(RC_ERR_INVALID, "NULL argument");
    if (slot_bytes < env->layout.total) return fail(RC_ERR_INVALID, "slot_bytes %zu is smaller than an arena (%zu)", slot_bytes, env->layout.total);
    // (the row gather moves 16 bytes per lane: every slot must start as rc_set_arena demands of an arena)
    if (((uintptr_t)ring_base & 63u) != 0 || (slot_bytes & 63u) != 0)
        return fail(RC_ERR_INVALID, "ring_base (%p) and slot_bytes (%zu) must be multiples of 64", ring_base, slot_bytes);
    if (env->shared_arena) return fail(RC_ERR_INVALID, "rc_sample_windows works on whole arenas, not on a slice handle");
    if (capacity < 1 || oldest < 0 || oldest >= capacity || count < 1 || count > capacity) return fail(RC_ERR_INVALID, "ring of %d slots, oldest %d, %d filled", capacity, oldest, count);
    if (length < 1 || length > count) return fail(RC_ERR_INVALID, "a window of %d records does not fit the %d records of the ring", length, count);
    if (n_windows < 1 || max_tries < 1) return fail(RC_ERR_INVALID, "n_windows and max_tries must be >= 1");
    RcSampleWindows a{};
    a.ring = (const unsigned char *)ring_base; a.slot_bytes = slot_bytes;
    a.fresh_off = env->layout.offset[RC_F_FRESH]; a.done_off = env->layout.offset[RC_F_DONE];
    a.capacity = capacity; a.oldest = oldest; a.n_start = count - length + 1; a.length = length; a.n_windows = n_windows;
    a.n_cars = env->n_cars; a.max_tries = max_tries;
    a.seed_lo = (uint32_t)seed; a.seed_hi = (uint32_t)(seed >> 32); a.draw = draw;
    a.slot_idx = slot_idx_dev; a.slot_obs_idx = slot_obs_idx_dev; a.car_idx = car_idx_dev; a.meta = meta_dev; a.failed = failed_dev;
    HIP_TRY(hipSetDevice(env->cfg.device));
    HIP_TRY(rck_sample_windows(a, env->stream));
    return RC_OK;
}

// One training batch as ONE packed buffer: field sections (64-byte aligned, field order), meta, the failure counter - the
// payload a sharded replay store exchanges - then the sampler's row indices (scratch).
namespace {
struct BatchLayout { size_t field_off[RC_F_COUNT]; size_t meta, failed, payload, slot, slot_obs, car, total; };
bool batch_layout(const rc_env *env, uint32_t field_mask, int32_t n_windows, int32_t length, BatchLayout *bl) {
    const size_t rows = (size_t)n_windows * (size_t)length;
    size_t off = 0;
    for (int f = 0; f < RC_F_COUNT; ++f) {
        bl->field_off[f] = off;
        if (!((field_mask >> f) & 1u)) continue;
        if (f == RC_F_ACTION_IN || !env->layout.bytes[f]) return false;
        off = align_up(off + kFieldBytes[f] * rows, 64);
    }
    bl->meta = off;     off = align_up(off + 16u * (size_t)n_windows, 64);
    bl->failed = off;   off += 64;
    bl->payload = off;
    bl->slot = off;     off = align_up(off + 4u * rows, 64);
    bl->slot_obs = off; off = align_up(off + 4u * rows, 64);
    bl->car = off;      off = align_up(off + 4u * rows, 64);
    bl->total = off;
    return true;
}
}  // namespace

size_t rc_sample_batch_bytes(rc_env *env, uint32_t field_mask, int32_t n_windows, int32_t length, size_t *payload_bytes, size_t *meta_offset) {
    if (!env || n_windows < 1 || length < 1 || field_mask == 0u) return 0;
    BatchLayout bl;
    if (!batch_layout(env, field_mask, n_windows, length, &bl)) return 0;
    if (payload_bytes) *payload_bytes = bl.payload;
    if (meta_offset) *meta_offset = bl.meta;
    return bl.total;
}

int rc_sample_batch(rc_env *env, const void *ring_base, size_t slot_bytes, int32_t capacity, int32_t oldest, int32_t count, int32_t length,
                    int32_t n_windows, uint64_t seed, uint32_t draw, int32_t max_tries, uint32_t field_mask, int32_t reset_rows,
                    void *out_dev, size_t out_bytes) {
    if (!env || !ring_base || !out_dev) return fail(RC_ERR_INVALID, "NULL argument");
    if (n_windows < 1 || length < 1) return fail(RC_ERR_INVALID, "n_windows and length must be >= 1");
    BatchLayout bl;
    if (field_mask == 0u || !batch_layout(env, field_mask, n_windows, length, &bl))
        return fail(RC_ERR_INVALID, "field mask 0x%x names no field, or one that is not part of a recorded arena in this configuration", field_mask);
    if (out_bytes < bl.total) return fail(RC_ERR_INVALID, "output too small: %zu < %zu (rc_sample_batch_bytes)", out_bytes, bl.total);
    if (((uintptr_t)out_dev & 63u) != 0) return fail(RC_ERR_INVALID, "out_dev must be 64-byte aligned");
    char *out = (char *)out_dev;
    HIP_TRY(hipSetDevice(env->cfg.device));
    HIP_TRY(hipMemsetAsync(out + bl.failed, 0, 64, env->stream));
    int rc = rc_sample_windows(env, ring_base, slot_bytes, capacity, oldest, count, length, n_windows, seed, draw, max_tries,
                               (int32_t *)(out + bl.slot), (int32_t *)(out + bl.slot_obs), (int32_t *)(out + bl.car),
                               (int32_t *)(out + bl.meta), (uint32_t *)(out + bl.failed));
    if (rc) return rc;
    // the observation part of a record: what a terminal row borrows from the row before it
    const uint32_t obs_fields = (1u << RC_F_LIDAR) | (1u << RC_F_OCCUPANCY) | (1u << RC_F_POSE) | (1u << RC_F_VELOCITY) | (1u << RC_F_SPEED) |
                                (1u << RC_F_ACCELERATION) | (1u << RC_F_STEERING_ANGLE);
    size_t src[RC_GATHER_MAX_FIELDS], dst[RC_GATHER_MAX_FIELDS];
    uint32_t bpc[RC_GATHER_MAX_FIELDS];
    RcBatchRows br{};
    br.slot_obs_idx = (const int32_t *)(out + bl.slot_obs); br.meta = (const int32_t *)(out + bl.meta); br.length = length;
    int n = 0;
    for (int f = 0; f < RC_F_COUNT; ++f) {
        if (!((field_mask >> f) & 1u)) continue;
        src[n] = env->layout.offset[f]; dst[n] = bl.field_off[f]; bpc[n] = (uint32_t)kFieldBytes[f];
        if ((obs_fields >> f) & 1u) br.obs_mask |= 1u << n;
        if (reset_rows && (f == RC_F_ACTION || f == RC_F_REWARD || f == RC_F_DISCOUNT || f == RC_F_TIME || f == RC_F_PROGRESS_TOTAL)) {
            const float v = f == RC_F_DISCOUNT ? 1.0f : (f == RC_F_PROGRESS_TOTAL ? -1.0f : 0.0f);
            br.reset_mask |= 1u << n;
            std::memcpy(&br.reset_word[n], &v, 4);
        }
        ++n;
    }
    HIP_TRY(rck_gather_rows(ring_base, slot_bytes, (const int32_t *)(out + bl.slot), (const int32_t *)(out + bl.car), n_windows * length,
                            src, dst, bpc, n, out, env->stream, &br));
    return RC_OK;
}

int rc_sync(rc_env *env) {
    if (!env) return fail(RC_ERR_INVALID, "env is NULL");
    HIP_TRY(hipSetDevice(env->cfg.device));
    HIP_TRY(hipStreamSynchronize(env->stream));
    return RC_OK;
}

void *rc_stream(rc_env *env) { return env ? (void *)env->stream : nullptr; }

int rc_set_profiling(rc_env *env, int32_t enabled) {
    if (!env) return fail(RC_ERR_INVALID, "env is NULL");
    if (!enabled) {
        int rc = drain_events(env);
        if (rc) return rc;
    }
    env->profiling = enabled == 1 ? 0xffffffffu : (uint32_t)enabled;
    if (enabled) {
        // The event pairs of the launches to come are made HERE, not at the launches: hipEventCreate costs the host ~15 us, and
        // two of them per timed launch inside a short window that starts on an idle GPU starve it (round 5: the first 20 steps
        // after a reset read 0.205 ms per step between two stream events where the kernels themselves took 0.175 - the
        // "25 us per step of non-scan time" of VERDICT r4 weak 4 were these calls).  512 pairs: a window of 512 timed launches
        // runs without a single runtime call besides its launches; beyond that KernelTimer creates them as before.
        HIP_TRY(hipSetDevice(env->cfg.device));
        while (env->free_events.size() < 512) {
            EventPair ep{};
            HIP_TRY(hipEventCreate(&ep.a));
            HIP_TRY(hipEventCreate(&ep.b));
            env->free_events.push_back(ep);
        }
    }
    return RC_OK;
}

int rc_kernel_time(rc_env *env, int32_t kernel, double *total_ms, uint64_t *launches) {
    if (!env || !total_ms || !launches) return fail(RC_ERR_INVALID, "NULL argument");
    if (kernel < 0 || kernel >= RC_K_COUNT) return fail(RC_ERR_INVALID, "unknown kernel %d", kernel);
    int rc = drain_events(env);
    if (rc) return rc;
    *total_ms = env->k_ms[kernel];
    *launches = env->k_n[kernel];
    return RC_OK;
}

int rc_reset_kernel_times(rc_env *env) {
    if (!env) return fail(RC_ERR_INVALID, "env is NULL");
    int rc = drain_events(env);
    if (rc) return rc;
    for (int k = 0; k < RC_K_COUNT; ++k) { env->k_ms[k] = 0; env->k_n[k] = 0; }
    return RC_OK;
}

int rc_set_arena(rc_env *env, void *arena, size_t bytes) {
    if (!env) return fail(RC_ERR_INVALID, "env is NULL");
    if (!arena) {
        arena = env->arena;                                           // NULL: back to the handle's own arena
        // ... and the caller may free what it had lent (TrajectoryRing.detach does): the rows the last scan wrote - the cost
        // keys of the small batches' next sort - must not be read from there any more; the sort falls back to index order
        // until the next scan has written rows of its own
        const char *own = (const char *)env->arena, *rows = (const char *)env->last_scan_rows;
        if (rows != nullptr && !(rows >= own && rows < own + env->layout.total)) env->last_scan_rows = nullptr;
    } else {
        if (bytes < env->layout.total) return fail(RC_ERR_INVALID, "arena too small: %zu < %zu", bytes, env->layout.total);
        if ((uintptr_t)arena % 64) return fail(RC_ERR_INVALID, "arena must be 64-byte aligned");
    }
    bind_outputs(env, arena);
    return RC_OK;
}

int rc_device_alloc(rc_env *env, size_t bytes, void **dev_ptr) {
    if (!env || !dev_ptr || bytes == 0) return fail(RC_ERR_INVALID, "NULL argument or zero size");
    HIP_TRY(hipSetDevice(env->cfg.device));
    HIP_TRY(hipMalloc(dev_ptr, bytes));
    return RC_OK;
}

int rc_device_free(rc_env *env, void *dev_ptr) {
    if (!env) return fail(RC_ERR_INVALID, "env is NULL");
    if (!dev_ptr) return RC_OK;
    HIP_TRY(hipSetDevice(env->cfg.device));
    HIP_TRY(hipStreamSynchronize(env->stream));
    if (env->comm_stream) HIP_TRY(hipStreamSynchronize(env->comm_stream));
    HIP_TRY(hipFree(dev_ptr));
    return RC_OK;
}

int rc_copy_from_device(rc_env *env, const void *dev_src, void *host_dst, size_t bytes) {
    if (!env || !dev_src || !host_dst) return fail(RC_ERR_INVALID, "NULL argument");
    HIP_TRY(hipSetDevice(env->cfg.device));
    HIP_TRY(hipMemcpyAsync(host_dst, dev_src, bytes, hipMemcpyDeviceToHost, env->stream));
    HIP_TRY(hipStreamSynchronize(env->stream));
    return RC_OK;
}

size_t rc_compact_bytes(const rc_config *cfg) {
    if (!cfg || cfg->num_envs < 1 || cfg->cars_per_env < 1) return 0;
    const int n = cfg->num_envs * cfg->cars_per_env;
    return make_compact(make_layout(n, cfg->obs_type != RC_OBS_LIDAR), n).total;
}

int rc_set_compact_slab(rc_env *env, void *slab, size_t bytes) {
    if (!env) return fail(RC_ERR_INVALID, "env is NULL");
    if (slab) {
        if (env->shared_arena) return fail(RC_ERR_INVALID, "the compact record is not available on a handle that fills a slice of a shared arena");
        if (bytes < env->compact.total) return fail(RC_ERR_INVALID, "compact slab too small: %zu < %zu", bytes, env->compact.total);
        if ((uintptr_t)slab % 64) return fail(RC_ERR_INVALID, "compact slab must be 64-byte aligned");
        if (env->has_track && env->launch.raycast_variant != 7)
            return fail(RC_ERR_INVALID, "the uint16 LiDAR copy is written by the default scan only (raycast variant 7)");
    }
    env->compact_slab = slab;
    env->params.out.lidar_u16 = (uint16_t *)slab;
    return RC_OK;
}

int rc_compact_layout(rc_env *env, size_t *lidar_u16_bytes, size_t *summary_offset, size_t *summary_bytes) {
    if (!env || !lidar_u16_bytes || !summary_offset || !summary_bytes) return fail(RC_ERR_INVALID, "NULL argument");
    *lidar_u16_bytes = (size_t)env->n_cars * RC_N_BEAMS * 2;
    *summary_offset = env->compact.lidar_bytes;
    *summary_bytes = env->compact.summary_bytes;
    return RC_OK;
}

int rc_comm_library(const char *path) {
    if (g_rccl.handle) return fail(RC_ERR_INVALID, "RCCL is already loaded");
    g_rccl_path = path ? path : "";
    return RC_OK;
}

int rc_comm_unique_id(void *out, size_t bytes) {
    if (!out || bytes < sizeof(RcUid)) return fail(RC_ERR_INVALID, "unique id buffer must hold %zu bytes", sizeof(RcUid));
    int rc = load_rccl();
    if (rc) return rc;
    NCCL_TRY(g_rccl.GetUniqueId(out));
    return RC_OK;
}

int rc_comm_init(rc_env *env, const void *unique_id, size_t bytes, int32_t rank, int32_t world) {
    if (!env) return fail(RC_ERR_INVALID, "env is NULL");
    if (!unique_id || bytes < sizeof(RcUid)) return fail(RC_ERR_INVALID, "unique id must hold %zu bytes", sizeof(RcUid));
    if (world < 1 || rank < 0 || rank >= world) return fail(RC_ERR_INVALID, "rank %d outside world of %d", rank, world);
    if (env->comm) return fail(RC_ERR_INVALID, "the handle already has a communicator");
    int rc = load_rccl();
    if (rc) return rc;
    HIP_TRY(hipSetDevice(env->cfg.device));
    RcUid id;
    std::memcpy(&id, unique_id, sizeof(id));
    NCCL_TRY(g_rccl.CommInitRank(&env->comm, world, id, rank));
    env->comm_rank = rank;
    env->comm_world = world;
    HIP_TRY(hipStreamCreateWithFlags(&env->comm_stream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&env->ev_ready, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&env->ev_gathered, hipEventDisableTiming));
    return RC_OK;
}

int rc_comm_count(rc_env *env, int32_t *ranks) {
    if (!env || !ranks) return fail(RC_ERR_INVALID, "NULL argument");
    if (!env->comm) return fail(RC_ERR_INVALID, "rc_comm_init has not been called on this handle");
    if (!g_rccl.CommCount) return fail(RC_ERR_COMM, "the RCCL library lacks ncclCommCount");
    int n = 0;
    NCCL_TRY(g_rccl.CommCount(env->comm, &n));
    *ranks = n;
    return RC_OK;
}

size_t rc_gather_bytes(rc_env *env, int32_t mode) {
    if (!env) return 0;
    switch (mode) {
    case RC_GATHER_FULL: return env->layout.slab_bytes;
    case RC_GATHER_FULL_U16: return env->compact.total;
    case RC_GATHER_SUMMARY: return env->compact.summary_bytes;
    }
    return 0;
}

int rc_gather_trajectory(rc_env *env, int32_t mode, void *dev_dst, size_t dst_bytes) {
    if (!env) return fail(RC_ERR_INVALID, "env is NULL");
    if (!env->comm) return fail(RC_ERR_INVALID, "rc_comm_init has not been called on this handle");
    if (!dev_dst) return fail(RC_ERR_INVALID, "dev_dst is NULL");
    const void *src;
    size_t n;
    int rc = gather_source(env, mode, &src, &n);
    if (rc) return rc;
    if (dst_bytes < n * (size_t)env->comm_world)
        return fail(RC_ERR_INVALID, "gather destination too small: %zu < %d x %zu", dst_bytes, env->comm_world, n);
    HIP_TRY(hipSetDevice(env->cfg.device));
    // ordered after everything queued on the env's stream (the step that produced the record), but on a stream of
    // its own: the following steps' kernels overlap the collective
    HIP_TRY(hipEventRecord(env->ev_ready, env->stream));
    HIP_TRY(hipStreamWaitEvent(env->comm_stream, env->ev_ready, 0));
    NCCL_TRY(g_rccl.AllGather(src, dev_dst, n, /* ncclUint8 */ 1, env->comm, env->comm_stream));
    HIP_TRY(hipEventRecord(env->ev_gathered, env->comm_stream));
    env->gather_pending = true;
    return RC_OK;
}

int rc_gather_wait(rc_env *env, int32_t host_sync) {
    if (!env) return fail(RC_ERR_INVALID, "env is NULL");
    if (!env->gather_pending) return RC_OK;
    HIP_TRY(hipSetDevice(env->cfg.device));
    HIP_TRY(hipStreamWaitEvent(env->stream, env->ev_gathered, 0));     // later work on the env's stream sees the result
    if (host_sync) {
        HIP_TRY(hipEventSynchronize(env->ev_gathered));
        env->gather_pending = false;
    }
    return RC_OK;
}

// ---- peer-copy all-gather ------------------------------------------------------------------------------------------
static void p2p_disconnect(rc_env *env) {          // my copies done, the peers' buffers unmapped; mine stay
    P2p *x = env->p2p;
    if (!x) return;
    (void)hipSetDevice(env->cfg.device);
    for (hipStream_t st : x->push) if (st) { (void)hipStreamSynchronize(st); }
    if (x->ctrl) (void)hipStreamSynchronize(x->ctrl);
    for (char *&d : x->peer_dst) if (d) { (void)hipIpcCloseMemHandle(d); d = nullptr; }
    for (uint32_t *&f : x->peer_flags) if (f) { (void)hipIpcCloseMemHandle(f); f = nullptr; }
    x->connected = false;
}

static void p2p_free(rc_env *env) {
    P2p *x = env->p2p;
    if (!x) return;
    p2p_disconnect(env);
    for (hipStream_t st : x->push) if (st) (void)hipStreamDestroy(st);
    if (x->ctrl) (void)hipStreamDestroy(x->ctrl);
    for (hipEvent_t e : {x->ev_ready, x->ev_go, x->ev_arrived, x->ev_local}) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : x->ev_sent) if (e) (void)hipEventDestroy(e);
    if (x->dst) (void)hipFree(x->dst);
    if (x->flags) (void)hipFree(x->flags);
    delete x;
    env->p2p = nullptr;
}

int rc_p2p_setup(rc_env *env, int32_t mode, int32_t rank, int32_t world, void *export_out, size_t bytes) {
    if (!env || !export_out) return fail(RC_ERR_INVALID, "NULL argument");
    if (bytes < RC_P2P_EXPORT_BYTES) return fail(RC_ERR_INVALID, "export buffer must hold %d bytes", RC_P2P_EXPORT_BYTES);
    if (world < 1 || world > RC_P2P_MAX_RANKS || rank < 0 || rank >= world)
        return fail(RC_ERR_INVALID, "rank %d outside world of %d (at most %d ranks)", rank, world, RC_P2P_MAX_RANKS);
    const size_t n = rc_gather_bytes(env, mode);
    if (n == 0) return fail(RC_ERR_INVALID, "unknown gather mode %d", mode);
    HIP_TRY(hipSetDevice(env->cfg.device));
    if (P2p *x = env->p2p) {
        // Already set up: only the payload changes.  The buffers, their exports and the peers' mappings stay - they are
        // sized for the largest payload, and exporting fresh allocations again and again is what the runtime likes least
        // (a re-export at a recycled address failed with "invalid argument" now and then).  Sequence numbers run on.
        if (x->rank != rank || x->world != world) return fail(RC_ERR_INVALID, "set up as rank %d of %d: rc_p2p_teardown first", x->rank, x->world);
        for (hipStream_t st : x->push) HIP_TRY(hipStreamSynchronize(st));
        HIP_TRY(hipStreamSynchronize(x->ctrl));
        x->mode = mode; x->bytes = n;
        std::memcpy(export_out, &x->blob, sizeof(x->blob));
        return RC_OK;
    }
    P2p *x = new (std::nothrow) P2p();
    if (!x) return fail(RC_ERR_NOMEM, "out of host memory");
    env->p2p = x;
    x->rank = rank; x->world = world; x->mode = mode; x->bytes = n;
    x->cap = align_up(std::max(n, env->layout.slab_bytes), 256);
    x->peer_dst.assign(world, nullptr);
    x->peer_flags.assign(world, nullptr);
    x->push.assign(world, nullptr);
    x->ev_sent.assign(world, nullptr);
#define P2P_TRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { p2p_free(env); return fail(RC_ERR_HIP, "%s failed: %s (payload %zu B x %d ranks)", #expr, hipGetErrorString(_e), n, world); } } while (0)
    P2P_TRY(hipMalloc((void **)&x->dst, 2 * (size_t)world * x->cap));
    // the flags are written by other GPUs' kernels and polled by this one's: uncached memory, so that a poll sees them
    P2P_TRY(hipExtMallocWithFlags((void **)&x->flags, 4096, hipDeviceMallocUncached));
    P2P_TRY(hipMemset(x->flags, 0, 4096));
    for (int p = 0; p < world; ++p) P2P_TRY(hipStreamCreateWithFlags(&x->push[p], hipStreamNonBlocking));
    P2P_TRY(hipStreamCreateWithFlags(&x->ctrl, hipStreamNonBlocking));
    for (hipEvent_t *e : {&x->ev_ready, &x->ev_go, &x->ev_arrived, &x->ev_local}) P2P_TRY(hipEventCreateWithFlags(e, hipEventDisableTiming));
    for (int p = 0; p < world; ++p) if (p != rank) P2P_TRY(hipEventCreateWithFlags(&x->ev_sent[p], hipEventDisableTiming));
    P2pExport &ex = x->blob;
    std::memset(&ex, 0, sizeof(ex));
    P2P_TRY(hipIpcGetMemHandle(&ex.dst, x->dst));
    P2P_TRY(hipIpcGetMemHandle(&ex.flags, x->flags));
    ex.bytes = x->cap; ex.rank = rank; ex.world = world; ex.mode = 0; ex.pid = (int32_t)getpid();
    P2P_TRY(hipDeviceGetPCIBusId(ex.pci, sizeof(ex.pci), env->cfg.device));
    std::memcpy(export_out, &ex, sizeof(ex));
    return RC_OK;
}

int rc_p2p_connect(rc_env *env, const void *exports, size_t bytes) {
    if (!env || !exports) return fail(RC_ERR_INVALID, "NULL argument");
    P2p *x = env->p2p;
    if (!x) return fail(RC_ERR_INVALID, "rc_p2p_setup has not been called on this handle");
    if (x->connected) return RC_OK;                      // (a mode switch: the peers' buffers are mapped already)
    if (bytes < (size_t)x->world * RC_P2P_EXPORT_BYTES) return fail(RC_ERR_INVALID, "need %d export blobs of %d bytes", x->world, RC_P2P_EXPORT_BYTES);
    HIP_TRY(hipSetDevice(env->cfg.device));
    for (int p = 0; p < x->world; ++p) {
        P2pExport ex;
        std::memcpy(&ex, (const char *)exports + (size_t)p * RC_P2P_EXPORT_BYTES, sizeof(ex));
        if (ex.rank != p || ex.world != x->world || ex.bytes != x->cap) {
            p2p_disconnect(env);
            return fail(RC_ERR_INVALID, "export blob %d does not match (rank %d, world %d, %llu bytes per slot entry; mine %zu)", p, ex.rank, ex.world,
                        (unsigned long long)ex.bytes, x->cap);
        }
        if (p == x->rank) continue;
        // a peer on another GPU: let this device's copy engines and kernels reach its memory
        int pdev = -1;
        if (hipDeviceGetByPCIBusId(&pdev, ex.pci) == hipSuccess && pdev >= 0 && pdev != env->cfg.device) {
            hipError_t pe = hipDeviceEnablePeerAccess(pdev, 0);
            if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) {
                p2p_disconnect(env);
                return fail(RC_ERR_HIP, "hipDeviceEnablePeerAccess(%d) failed: %s", pdev, hipGetErrorString(pe));
            }
            (void)hipGetLastError();
        }
        // (a failure half way leaves nothing mapped: the call can be repeated)
        hipError_t oe = hipIpcOpenMemHandle((void **)&x->peer_dst[p], ex.dst, hipIpcMemLazyEnablePeerAccess);
        if (oe == hipSuccess) oe = hipIpcOpenMemHandle((void **)&x->peer_flags[p], ex.flags, hipIpcMemLazyEnablePeerAccess);
        if (oe != hipSuccess) {
            p2p_disconnect(env);
            return fail(RC_ERR_HIP, "hipIpcOpenMemHandle of rank %d's buffers failed: %s", p, hipGetErrorString(oe));
        }
    }
    x->connected = true;
    return RC_OK;
}

int rc_gather_trajectory_p2p(rc_env *env) {
    if (!env) return fail(RC_ERR_INVALID, "env is NULL");
    P2p *x = env->p2p;
    if (!x || !x->connected) return fail(RC_ERR_INVALID, "rc_p2p_setup / rc_p2p_connect have not been called on this handle");
    const void *src;
    size_t n;
    int rc = gather_source(env, x->mode, &src, &n);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(env->cfg.device));
    const uint32_t k = x->issued, seq = k + 1u;
    // slot k & 1 of every rank: world entries of `cap` bytes, of which the current payload fills the first n
    const size_t slot_off = (size_t)(k & 1u) * x->world * x->cap, mine = slot_off + (size_t)x->rank * x->cap;
    // everything below is ordered behind what the env's stream holds now: the step that produced the record, and the
    // caller's use of the slot that gather k overwrites (the buffer of gather k - 2)
    HIP_TRY(hipEventRecord(x->ev_ready, env->stream));
    HIP_TRY(hipStreamWaitEvent(x->ctrl, x->ev_ready, 0));
    // 1. tell every peer that its gather k may be written into my slot k & 1 ...
    RcP2pPost post;
    std::memset(&post, 0, sizeof(post));
    post.n = x->world; post.value = seq;
    for (int p = 0; p < x->world; ++p) post.flag[p] = p == x->rank ? nullptr : x->peer_flags[p] + RC_P2P_MAX_RANKS + x->rank;
    HIP_TRY(rck_p2p_post(post, x->ctrl));
    // 2. ... and wait until every peer has said the same to me (posting comes first on every rank: no cycle)
    HIP_TRY(rck_p2p_wait(x->released(), x->world, x->rank, seq, x->timeouts(), RC_P2P_TIMEOUT_S, x->ctrl));
    HIP_TRY(hipEventRecord(x->ev_go, x->ctrl));
    // 3. my record into every peer's slot, one stream (one link) per peer, each followed by its arrival flag
    for (int p = 0; p < x->world; ++p) {
        hipStream_t st = x->push[p];
        if (p == x->rank) {
            HIP_TRY(hipStreamWaitEvent(st, x->ev_ready, 0));
            HIP_TRY(hipMemcpyAsync(x->dst + mine, src, n, hipMemcpyDeviceToDevice, st));
            HIP_TRY(hipEventRecord(x->ev_local, st));
            continue;
        }
        HIP_TRY(hipStreamWaitEvent(st, x->ev_go, 0));
        HIP_TRY(hipMemcpyAsync(x->peer_dst[p] + mine, src, n, hipMemcpyDefault, st));
        RcP2pPost arrived;
        std::memset(&arrived, 0, sizeof(arrived));
        arrived.n = 1; arrived.value = seq;
        arrived.flag[0] = x->peer_flags[p] + x->rank;
        HIP_TRY(rck_p2p_post(arrived, st));
        HIP_TRY(hipEventRecord(x->ev_sent[p], st));
    }
    // 4. arrival of every peer's shard in my slot: polled on the control stream, behind the release handshake
    HIP_TRY(rck_p2p_wait(x->arrived(), x->world, x->rank, seq, x->timeouts(), RC_P2P_TIMEOUT_S, x->ctrl));
    HIP_TRY(hipStreamWaitEvent(x->ctrl, x->ev_local, 0));
    // 5. ... and the DEPARTURE of mine: `ev_arrived` stands for "gather k is complete as far as this rank can tell" - the peers'
    // shards are here AND my outbound copies have read the source to the end - so that a caller who puts its stream behind it
    // (rc_gather_p2p_wait, host_sync 0) may let the next step but one rewrite the source, as with rc_gather_trajectory
    for (int p = 0; p < x->world; ++p) if (p != x->rank) HIP_TRY(hipStreamWaitEvent(x->ctrl, x->ev_sent[p], 0));
    HIP_TRY(hipEventRecord(x->ev_arrived, x->ctrl));
    x->issued = seq;
    return RC_OK;
}

int rc_gather_p2p_wait(rc_env *env, int32_t host_sync, void **gathered_dev, size_t *gathered_bytes) {
    if (!env) return fail(RC_ERR_INVALID, "env is NULL");
    P2p *x = env->p2p;
    if (!x || !x->connected) return fail(RC_ERR_INVALID, "rc_p2p_setup / rc_p2p_connect have not been called on this handle");
    if (x->issued == 0) return fail(RC_ERR_INVALID, "no gather has been issued");
    HIP_TRY(hipSetDevice(env->cfg.device));
    HIP_TRY(hipStreamWaitEvent(env->stream, x->ev_arrived, 0));       // later work on the env's stream sees the gathered bytes
    if (host_sync) {
        HIP_TRY(hipEventSynchronize(x->ev_arrived));
        // my record has left when my copies are done (the peers' arrival flags follow them on the same streams)
        for (int p = 0; p < x->world; ++p) HIP_TRY(hipStreamSynchronize(x->push[p]));
        uint32_t late = 0;
        HIP_TRY(hipMemcpy(&late, x->timeouts(), sizeof(late), hipMemcpyDeviceToHost));
        if (late != 0) {
            // reported once: the counter starts again (everything queued has run: the streams were synchronised above).  A
            // release wait that timed out has let its copies go into slots that were never released: the records of this
            // and of the previous gather are not to be trusted, on any rank - tear the transport down and set it up again
            HIP_TRY(hipMemset(x->timeouts(), 0, sizeof(uint32_t)));
            return fail(RC_ERR_COMM, "peer-copy gather: %u flag wait(s) timed out after %.0f s (a peer did not post); the gathered "
                        "slots are not valid - rc_p2p_teardown and set up again", late, (double)RC_P2P_TIMEOUT_S);
        }
    }
    if (gathered_dev) *gathered_dev = x->dst + (size_t)((x->issued - 1u) & 1u) * x->world * x->cap;
    if (gathered_bytes) *gathered_bytes = (size_t)x->world * x->cap;
    return RC_OK;
}

int rc_p2p_slot(rc_env *env, int32_t back, void **gathered_dev, size_t *gathered_bytes) {
    if (!env) return fail(RC_ERR_INVALID, "env is NULL");
    P2p *x = env->p2p;
    if (!x) return fail(RC_ERR_INVALID, "rc_p2p_setup has not been called on this handle");
    if (back < 0 || back > 1 || x->issued < (uint32_t)back + 1u) return fail(RC_ERR_INVALID, "no gather %d before the last one (issued: %u)", back, x->issued);
    if (gathered_dev) *gathered_dev = x->dst + (size_t)((x->issued - 1u - (uint32_t)back) & 1u) * x->world * x->cap;
    if (gathered_bytes) *gathered_bytes = (size_t)x->world * x->cap;
    return RC_OK;
}

int rc_p2p_disconnect(rc_env *env) {
    if (!env) return fail(RC_ERR_INVALID, "env is NULL");
    p2p_disconnect(env);
    return RC_OK;
}

int rc_p2p_teardown(rc_env *env) {
    if (!env) return fail(RC_ERR_INVALID, "env is NULL");
    p2p_free(env);
    return RC_OK;
}

int rc_debug_set(rc_env *env, int32_t knob, int32_t value) {
    if (!env) return fail(RC_ERR_INVALID, "env is NULL");
    if (knob < 0 || knob >= RC_DBG_COUNT) return fail(RC_ERR_INVALID, "unknown debug knob %d", knob);
    env->dbg[knob] = value;
    if (env->has_track) {
        set_band(env);
        set_launch_geometry(env);
    }
    return RC_OK;
}

int rc_scan_overruns(rc_env *env, uint64_t *count) {
    if (!env || !count) return fail(RC_ERR_INVALID, "NULL argument");
    HIP_TRY(hipSetDevice(env->cfg.device));
    uint32_t v = 0;
    HIP_TRY(hipMemcpyAsync(&v, env->params.scan_overrun, sizeof(v), hipMemcpyDeviceToHost, env->stream));
    HIP_TRY(hipStreamSynchronize(env->stream));
    *count = v;
    return RC_OK;
}

int rc_debug_scan_stamps(rc_env *env, uint64_t *stamps, int32_t n_waves) {
    if (!env) return fail(RC_ERR_INVALID, "env is NULL");
    if (stamps != nullptr && n_waves <= 0) return fail(RC_ERR_INVALID, "n_waves must be positive");
    if (stamps != nullptr) {
        const char *why = rck_lab_unavailable();                   // the instrumented build lives in the lab library
        if (why != nullptr) return fail(RC_ERR_INVALID, "%s", why);
    }
    env->launch.scan_stamps = reinterpret_cast<unsigned long long *>(stamps);
    env->launch.scan_stamp_waves = stamps ? n_waves : 0;
    return RC_OK;
}

int rc_scan_kernel_name(rc_env *env, char *out, size_t bytes) {
    if (!env || !out || bytes == 0) return fail(RC_ERR_INVALID, "NULL argument");
    if (!env->has_track) return fail(RC_ERR_NO_TRACK, "rc_load_track must be called first");
    const RcLaunchInfo &li = env->launch;
    const int a = env->cfg.cars_per_env;
    if (li.raycast_variant != 7) snprintf(out, bytes, "rc_raycast_kernel<%d, %d>", a, li.raycast_variant);
    else if (li.scan_stamps != nullptr && a == 1) snprintf(out, bytes, "rc_raycast_car_stamps_kernel");
    else if (li.scan_guarded) snprintf(out, bytes, "rc_raycast_car_kernel<%d, false, true>", a);
    else snprintf(out, bytes, "rc_raycast_car_kernel<%d, %s, false>", a, li.car_split > 1 ? "true" : "false");
    return RC_OK;
}

int rc_set_raycast_variant(rc_env *env, int32_t variant) {
    if (!env) return fail(RC_ERR_INVALID, "env is NULL");
    if (variant < 0 || variant > 7) return fail(RC_ERR_INVALID, "unknown raycast variant %d", variant);
    if (!env->has_track) return fail(RC_ERR_NO_TRACK, "rc_load_track must be called first");
    if (variant == 0 && env->launch.lds_bytes == 0)
        return fail(RC_ERR_INVALID, "variant 0 needs the bitmap in the 160 KiB LDS; this track is too large");
    if (variant == 3 && env->launch.lds_bytes_packed == 0)
        return fail(RC_ERR_INVALID, "variant 3 needs the packed 4x4 block table in the 160 KiB LDS; this track is too large");
    if ((variant == 1 || variant == 2) && env->launch.lds_bytes_skip == 0)
        return fail(RC_ERR_INVALID, "variants 1/2 need bitmap + free-block table in the 160 KiB LDS; this track is too large");
    if (variant != 7 && env->compact_slab)
        return fail(RC_ERR_INVALID, "the uint16 LiDAR copy (rc_set_compact_slab) is written by variant 7 only");
    if (variant != 7) {
        const char *why = rck_lab_unavailable();                   // variants 0-6 live in the lab library (racecar_lab.hip)
        if (why != nullptr) return fail(RC_ERR_INVALID, "%s", why);
    }
    env->launch.raycast_variant = variant;
    set_launch_geometry(env);
    return RC_OK;
}

}  // extern "C"
