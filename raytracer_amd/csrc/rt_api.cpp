// rt_api.cpp — host side of the C-ABI (include/raytracer_amd.h): validation, BVH build + upload, the per-frame
// launch schedule (passes -> levels -> fold -> resolve), statistics. No intersection or shading happens here.
//
// One rt_scene = one device. Its workspace (node pool, child-ray pool, counters) is owned by the scene and
// reused by every call, so calls on the SAME rt_scene must not overlap; different scenes are independent.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <string>
#include <vector>

#include "raytracer_amd.h"
#include "rt_device.h"
#include "rt_launch.h"
#include "bvh_device.h"

// ---------------------------------------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

void rt_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            rt_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            (void)hipGetLastError();      /* reported here: must not resurface in a later call's launch check */ \
            return RT_ERR_DEVICE;                                                                  \
        }                                                                                          \
    } while (0)

// ---------------------------------------------------------------------------------------------------------
// scene object
// ---------------------------------------------------------------------------------------------------------
struct PassCounters {                 // one per pass, zeroed by one memset per frame
    ShardCounters sc[RT_SHARDS];         // per shard: entries appended / queued per level
    uint32_t cancel[RT_MAX_LEVELS + 4];  // per level: children of parents that took trace()'s `return red` exit
    uint32_t overflow;
    uint32_t tail_levels[RT_MAX_LEVELS + 4];   // k_tail: entries per TRUE level (it counts them all in sc[].level[root + 1])
    uint32_t pad_[3];
    unsigned long long shadow[RT_STAT_SHARDS];
    unsigned long long work[RT_WORK_SHARDS][RT_N_KINDS][RT_WORK_WORDS];   // per shard: RT_K_* kernels
    uint32_t appended(int lv) const { uint32_t n = 0; for (uint32_t k = 0; k < RT_SHARDS; ++k) n += sc[k].level[lv]; return n; }
};

#define RT_MAX_PASSES 1024
#define RT_PASS_LOG2_MIN 12
#define RT_PASS_LOG2_MAX 26
// Level-0 samples per pass of a new scene (RT_PASS_LOG2 in the environment for experiments). A property of the scene
// (rt_scene::pass_samples, rt_scene_set_pass_samples), read when the scene is created, with its device current.
// Larger passes are faster - the bounce levels of a pass are a chain of dependent launches whose sizes grow with it:
// 4K 16 spp glass scene 38.2 ms per frame at 2^23, 36.6 at 2^24, 34.1 at 2^25, 32.0 at 2^26; 10 M-triangle field at 8K 64 spp
// 634 / 614 / 609 ms - and cost 672 B of workspace per sample (5.4 GB at 2^23, 45 GB at 2^26). So: the largest power of
// two up to 2^26 whose workspace stays below a quarter of what is FREE on the device when the scene is created (2^26 on
// an MI355X with 180 GB or more to spare; a process that keeps several scenes in flight on a device others use too gets
// smaller passes instead of taking memory it was never given), at least 2^23. A frame with fewer samples is one pass
// and allocates for what it has; the workspace itself is allocated by the first frame that needs it.
static uint32_t default_pass_samples(int env_log2) {
    if (env_log2 >= 0) return 1u << std::max(RT_PASS_LOG2_MIN, std::min(RT_PASS_LOG2_MAX, env_log2));
    size_t free_b = 0, total_b = 0;
    int k = 23;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess)
        while (k < RT_PASS_LOG2_MAX && (uint64_t(672) << (k + 1)) <= std::min(free_b, total_b) / 4u) ++k;
    else (void)hipGetLastError();
    return 1u << k;
}

#define RT_LEVEL_GRID 8192u           // workgroups of a level >= 1 launch (grid-stride over the device-side count)
// What a caller (a test, an experiment) may fix about the schedule instead of leaving it to the scene and the level
// sizes: rt_scene_set_option. The same names in upper case with an RT_ prefix in the environment are read ONCE, when a
// scene is created (tools/exp_*.sh); no call that renders reads the environment.
struct Knobs {
    int packet = -1;              // RT_PACKET: -1 by samples per pixel; 0 marks + quads at level 0; 1 packets at level 0; 2 at every large level
    long long large_thr = -1;     // RT_LARGE_THR: rays from which a bounce level runs as per-lane kernel (+ queue); -1: expect_large's table
    int large_maxlv = -1;         // RT_LARGE_MAXLV: deepest level that may be "large"; -1 default
    int serve = -1;               // RT_SERVE: bounce levels by quad service: 0 never, 1 always, -1 by pass size
    int shade_serve_lv = 0;       // RT_SHADE_SERVE_LV: with packets below level 0, the light loops of large levels >= this one by quad service (0: packets)
    // RT_SORT_SHADE: light loops that run as packets over levels >= 1 take their hits sorted by position (launch_sort_hits)
    // from this many rays on; 0: never. MEASURED AND NOT USED (tools/r3_run11.sh, r3_run12.sh; same bytes): launches one at
    // a time, the 8K field's level-1/2 light loops get a third faster (747 -> 679 ms of shadow packets over three frames,
    // 36 ms of key and sort launches), but the frame, whose launches overlap, gets slower: 403 -> 430 ms with 8 bits per
    // axis, 422 with 10, 440 with 6; 4K glass scene 27.2 -> 29.0. A wave's hits are then gathered from all over the pool
    // (three 16-B reads and a write per hit, each its own cache line) next to closest-hit launches that wait for memory.
    long long sort_shade = 0;
    int sort_bits = 8;            // RT_SORT_BITS: bits per axis of the sort key (1..10)
    int order0 = 1;               // RT_ORDER0: level-0 launches of a small one-pass frame hand out their groups in descending order of the
                                  // previous frame's cost (k_order_groups); 0: pool order
    int split_small = 1;          // RT_SPLIT_SMALL: runs of small levels shaded per lane + resuming quads (1) / one quad launch (0)
    int wide_waves_small = 4;     // RT_WIDE_WAVES_SMALL: waves per workgroup of the per-lane launches of a pass below the quad-service size (0: the default, one)
    int run_hits = 3 << 18;       // RT_RUN_HITS: a run of small levels that holds at least this many hits is shaded at once, not when three levels
                                  // have come together (0: never). A call of 20 1080p frames: level 1 alone is 7.4 M hits, and their light
                                  // loops, started behind level 1's closest hits instead of level 3's, end before level 0's do instead of
                                  // 0.17 ms after them: 0.241 -> 0.224 ms per frame (250 k .. 4 M: 0.222-0.234); 4K glass scene 26.3 -> 25.1 ms;
                                  // 800x600 in batches 0.0635 -> 0.0622; two calls of 32 frames in flight: the same; the 10 M-triangle field
                                  // at 1080p in batches 3.92 -> 3.98 (one frame at a time 4.58 -> 4.46). Where it starts to pay, a rank's share
                                  // of 10 tiled 1080p frames (tools/r3_run23.sh, threshold 250 k against never): a quarter of the tiles, level 1
                                  // 0.92 M hits, 0.0835 -> 0.077 ms per frame; an eighth, 0.46 M, 0.050 -> 0.053; one whole frame, 0.37 M,
                                  // 0.510 -> 0.527 - hence 3 x 2^18
    int fuse = 1;                 // RT_FUSE: a scene without meshes runs a level's closest hits and light loops in ONE launch (k_trace_wide)
    int tiny_run = 8192;          // RT_TINY_RUN: a run of at most this many hits is one quad launch (the run that ends a pass - the tail's
                                  // hits, at most tail_rays of them - is on the frame's critical path: one 1080p frame at a time
                                  // 0.513-0.519 ms at 1024, 0.504-0.507 at 6000-16384, 0.509 at 40000)
    long long tail = 8192;        // RT_TAIL: the levels that, with everything deeper, hold at most this many rays go to k_tail (0: never)
    int index64 = 0;              // RT_NO_FAST32: sample index arithmetic by 64-bit division even where 32 bits would do
    int chunk_log2 = RT_SHARD_CHUNK_LOG2;   // RT_CHUNK_LOG2
    int use_graph = 0;            // RT_USE_GRAPH: replay asynchronous frames as a hipGraph (measured slower)
    // RT_WIDE_GRID: most workgroups of a per-lane launch over levels >= 1 (its waves stride over the groups). 7168 = the wave
    // slots of the chip at the 7 waves per SIMD these kernels run at: one round, nothing waits for a slot. Measured (batched
    // bunny_1080p / 4K glass scene / field at 1080p, ms per frame): 2048: 0.254 / 31.0 / 4.32; 5120: 0.214 / 25.5 / 3.90;
    // 7168: 0.211 / 26.4 / 3.93; 8192: 0.214 / 26.6 / 3.93; 28672: 0.218 / 37.2 / 3.95; 262144 (a wave per group): 0.234 /
    // 56.0 / 4.14 - more workgroups than slots only adds the per-workgroup set-up (shard views, group scan) and takes the
    // rays out of pool order.
    int wide_grid = 7168;
    int grid0 = 0;                // RT_GRID0: cap of the level-0 per-lane grids (0: none)
    int pass_log2 = -1;           // RT_PASS_LOG2
    int shade_streams = 3;        // RT_SHADE_STREAMS
#ifdef RT_EXPERIMENTS
    uint32_t debug = 0;           // RT_DEBUG_SKIP: parts of the kernels switched off (timing experiments; WRONG frames)
    int wave_log_level = -1;      // RT_WAVE_LOG=<level>:<file>
    bool wave_log_shade = false;  // RT_WAVE_LOG_SHADE
    std::string wave_log_path;
#endif
};
static long long env_ll(const char* name, long long dflt) { const char* e = std::getenv(name); return e ? std::atoll(e) : dflt; }
static Knobs knobs_from_env() {
    Knobs k;
    k.packet = int(env_ll("RT_PACKET", k.packet));
    k.large_thr = env_ll("RT_LARGE_THR", k.large_thr);
    k.large_maxlv = int(env_ll("RT_LARGE_MAXLV", k.large_maxlv));
    k.serve = int(env_ll("RT_SERVE", k.serve));
    k.split_small = int(env_ll("RT_SPLIT_SMALL", k.split_small));
    k.order0 = int(env_ll("RT_ORDER0", k.order0)) ? 1 : 0;
    k.shade_serve_lv = int(env_ll("RT_SHADE_SERVE_LV", k.shade_serve_lv));
    k.sort_shade = std::max(0ll, env_ll("RT_SORT_SHADE", k.sort_shade));
    k.sort_bits = int(std::max(1ll, std::min(10ll, env_ll("RT_SORT_BITS", k.sort_bits))));
    k.tiny_run = int(std::max(0ll, env_ll("RT_TINY_RUN", k.tiny_run)));
    k.fuse = int(env_ll("RT_FUSE", k.fuse)) ? 1 : 0;
    k.run_hits = int(std::max(0ll, env_ll("RT_RUN_HITS", k.run_hits)));
    k.wide_waves_small = int(std::max(0ll, std::min(4ll, env_ll("RT_WIDE_WAVES_SMALL", k.wide_waves_small))));
    k.tail = std::max(0ll, env_ll("RT_TAIL", k.tail));
    k.index64 = std::getenv("RT_NO_FAST32") ? 1 : 0;
    k.chunk_log2 = int(std::min<long long>(RT_SHARD_CHUNK_LOG2_MAX, std::max(0ll, env_ll("RT_CHUNK_LOG2", k.chunk_log2))));
    k.use_graph = std::getenv("RT_USE_GRAPH") ? 1 : 0;
    k.grid0 = int(std::max(0ll, env_ll("RT_GRID0", 0)));
    k.wide_grid = int(std::max(64ll, std::min(1ll << 22, env_ll("RT_WIDE_GRID", k.wide_grid))));
    k.pass_log2 = int(env_ll("RT_PASS_LOG2", -1));
    k.shade_streams = int(std::max(1ll, std::min(3ll, env_ll("RT_SHADE_STREAMS", 3))));
#ifdef RT_EXPERIMENTS
    k.debug = uint32_t(env_ll("RT_DEBUG_SKIP", 0));
    if (const char* e = std::getenv("RT_WAVE_LOG")) {
        k.wave_log_level = std::atoi(e);
        if (const char* c = std::strchr(e, ':')) k.wave_log_path = c + 1; else k.wave_log_level = -1;
    }
    k.wave_log_shade = std::getenv("RT_WAVE_LOG_SHADE") != nullptr;
#endif
    return k;
}
#define RT_POOL_FACTOR 6u             // node pool = 6x the pass size when materials bounce (level 0 + 5x for deeper levels)
#define RT_MAX_TIMED 2048
#define RT_MAX_GRAPHS 8
enum { RT_K_EXTEND_WIDE = 0, RT_K_SHADE_WIDE = 1, RT_K_EXTEND_QUAD = 2, RT_K_SHADE_QUAD = 3, RT_K_SHADE_PARKED = 4 };

struct rt_scene {
    int device = 0;
    hipStream_t stream = nullptr;
    Knobs knobs;                         // rt_scene_set_option / the environment when the scene was created
    int n_shade_streams = 3;             // Knobs::shade_streams (the streams exist from the start): 1..3
    hipStream_t shade_stream[3] = {nullptr, nullptr, nullptr};   // the shade kernels run here, next to the extend kernels:
                                                                 // large levels on [level & 1], runs of small levels on [2]
    hipEvent_t ev_sync[2 * RT_MAX_LEVELS + 5] = {};      // per level: per-lane extend done, quad extend done; joins
    DScene d{};                        // device pointers
    std::vector<void*> owned;          // everything to hipFree
    uint32_t n_models = 0, n_faces = 0, n_nodes = 0, bvh_depth = 0;
    uint64_t scene_bytes = 0;
    bool has_bounce = false;           // some object's material is reflective or refractive
    // workspace
    uint32_t pool_cap = 0, sub_cap = 0, ws_n0 = 0;
    uint32_t n_leaves = 0, bvh_stack = 0;
    bool packet_auto = true;             // DScene::packet chosen per frame (render_frame)
    bool no_packets_flag = false;        // created with RT_SCENE_NO_PACKETS: option "packet" cannot bring them back
    bool built_on_device = false;
    double build_seconds = 0;
    std::vector<uint32_t> model_roots;
    float4* node_a = nullptr;
    uint4* node_b = nullptr;
    float4* rays = nullptr;
    float4* hit_n = nullptr;
    float4* hit_p = nullptr;
    uint2* queue_e = nullptr;
    uint2* queue_s = nullptr;
    FrameParams* d_fp = nullptr;         // camera + background of the frame being rendered (see FrameParams)
    std::vector<float> aa_cache;         // anti-aliasing pattern currently in d_aa
    // hipGraph of the launches of one asynchronous frame, replayed while nothing but camera / background changes
    struct GraphKey { LevelArgs base; uint8_t* d_out; uint64_t total_px; uint32_t pass_px; uint32_t large_mask; uint32_t count; uint32_t debug; };
    struct CachedGraph { GraphKey key; hipGraphExec_t exec; hipStream_t last_stream; };
    std::vector<CachedGraph> graphs;     // most recently used last; at most RT_MAX_GRAPHS
    bool graph_ok = true;
    ShardCounters* h_hint = nullptr;     // pinned: level sizes of the most recent pass (async copy, read without a sync:
                                         // only a scheduling hint, results never depend on it)
    int hint_depth = RT_MAX_LEVELS;      // recursion depth of the frame the hints were taken from
    uint32_t* h_tail_hint = nullptr;     // pinned, [pass][RT_MAX_LEVELS + 4]: k_tail's entries per TRUE level (it counts them all
                                         // in the level behind its roots; without these the next frame would take the deeper levels for empty)
    PassCounters* d_counters = nullptr;
    PassCounters* h_counters = nullptr;   // pinned
    float* d_aa = nullptr;
    uint8_t* d_frame = nullptr; size_t frame_bytes = 0;
    hipEvent_t ev_begin = nullptr, ev_end = nullptr;
    std::vector<hipEvent_t> ev_pool;
    std::vector<hipEvent_t> ev_pass;     // rt_render_progress: one per pass of the frame, recorded behind its k_resolve
    bool count_work = false;
    rt_work last_work[RT_N_KINDS]{};   // RT_K_* kernels
    int last_overflow = 0;
    uint32_t* h_overflow = nullptr;      // pinned, written by k_resolve: an asynchronous frame ran out of pool
    uint32_t shrink_log2 = 0;            // passes are cut 2^k times smaller than RT_PASS_SAMPLES: learned from overflows, sticky
    uint32_t noalloc_shrink_log2 = 0;    // the same for rt_render_device_noalloc's passes of the static pool (its overflows teach only it)
    bool last_noalloc = false;           // the most recent asynchronous frame used the static pool
    hipStream_t last_stream = nullptr;   // stream of the most recent asynchronous frame
    uint32_t last_spp = 1;
    uint32_t ws_min_n0 = 0;              // the workspace is sized for at least this many samples per pass (grown after overflows)
    // rt_render_batch_host_async: two device frame buffers in turn, copied out on a stream of their own, so that the copy of
    // one call runs under the kernels of the next call on this scene too
    hipStream_t copy_stream = nullptr;   // created by the first call that needs it
    uint8_t* d_host_frames[2] = {nullptr, nullptr};
    size_t host_frames_bytes = 0;
    hipEvent_t ev_rendered[2] = {nullptr, nullptr}, ev_copied[2] = {nullptr, nullptr};
    uint32_t host_turn = 0;
    void* noalloc_ws = nullptr;          // this device's copy of the file-scope `scratch` array (rt_render_device_noalloc)
    uint32_t pass_samples = 1u << 23;    // level-0 samples per pass (frames with more are cut into passes of whole pixels)
    // sorted light loops (launch_sort_hits): one set of buffers per shade stream, grown on demand; the box of the meshes
    struct SortBuf { uint32_t* keys[2] = {nullptr, nullptr}; uint32_t* vals[2] = {nullptr, nullptr}; void* temp = nullptr; size_t temp_bytes = 0; uint32_t cap = 0; };
    SortBuf sort_buf[3];
    // expensive groups first (k_order_groups): per-group cost of the level-0 closest-hit [0] and light-loop [1] launch of the
    // most recent small one-pass frame, and the group numbers in descending order of it for the next one
    // TWO sets, taken in turn: a frame logs its costs into its set, follows the order that set holds - made by the frame before
    // the previous one - and leaves its own order there. The ordering kernel runs behind the frame and is not part of what the
    // frame waits for; with one set the NEXT frame found it still running whenever a frame ends right behind level 0 (no
    // bounce levels) and went in pixel order, every other frame. With two, the kernel a frame asks about was launched two
    // frames ago.
    struct OrderSet {
        uint32_t* cost[2] = {nullptr, nullptr};
        uint32_t* order[2] = {nullptr, nullptr};
        uint32_t n = 0;                  // groups the set's order was made for (0: none yet)
        hipEvent_t ev = nullptr;         // recorded behind k_order_groups; a later frame asks whether it has happened
        bool pending = false;
    };
    OrderSet oset[2];
    uint32_t order_turn = 0;             // frames that took a set so far
    uint32_t order_cap = 0;              // groups the arrays hold
    SortBox sort_box{};
    bool sort_box_ok = false;
};
#define RT_PASS_SAMPLES (s->pass_samples)
#define RT_POOL_MAX_N0 std::max<uint32_t>(s->pass_samples, 1u << 23)   // a pool that overflowed on 64 pixels grows up to that of a default pass

static int free_scene(rt_scene* s) {
    if (!s) return RT_OK;
    hipSetDevice(s->device);
    for (void* p : s->owned) hipFree(p);
    hipFree(s->node_a); hipFree(s->node_b); hipFree(s->rays); hipFree(s->hit_n); hipFree(s->hit_p); hipFree(s->queue_e); hipFree(s->queue_s); hipFree(s->d_counters); hipFree(s->d_aa); hipFree(s->d_frame);
    hipFree(s->d_host_frames[0]); hipFree(s->d_host_frames[1]);
    for (auto& o : s->oset) {
        hipFree(o.cost[0]); hipFree(o.cost[1]); hipFree(o.order[0]); hipFree(o.order[1]);
        if (o.ev) hipEventDestroy(o.ev);
    }
    for (auto& b : s->sort_buf) { hipFree(b.keys[0]); hipFree(b.keys[1]); hipFree(b.vals[0]); hipFree(b.vals[1]); hipFree(b.temp); }
    for (int k = 0; k < 2; ++k) { if (s->ev_rendered[k]) hipEventDestroy(s->ev_rendered[k]); if (s->ev_copied[k]) hipEventDestroy(s->ev_copied[k]); }
    if (s->copy_stream) hipStreamDestroy(s->copy_stream);
    if (s->h_counters) hipHostFree(s->h_counters);
    if (s->h_hint) hipHostFree(s->h_hint);
    if (s->h_tail_hint) hipHostFree(s->h_tail_hint);
    if (s->h_overflow) hipHostFree(s->h_overflow);
    for (auto& g : s->graphs) hipGraphExecDestroy(g.exec);
    hipFree(s->d_fp);
    if (s->ev_begin) hipEventDestroy(s->ev_begin);
    if (s->ev_end) hipEventDestroy(s->ev_end);
    for (hipEvent_t e : s->ev_pool) hipEventDestroy(e);
    for (hipEvent_t e : s->ev_pass) hipEventDestroy(e);
    for (hipEvent_t e : s->ev_sync) if (e) hipEventDestroy(e);
    for (hipStream_t st : s->shade_stream) if (st) hipStreamDestroy(st);
    if (s->stream) hipStreamDestroy(s->stream);
    delete s;
    return RT_OK;
}

template <typename T>
static int upload(rt_scene* s, const T* host, size_t n, const T** out) {
    void* p = nullptr;
    const size_t bytes = std::max<size_t>(n, 1) * sizeof(T);
    HIP_TRY(hipMalloc(&p, bytes));
    s->owned.push_back(p);
    if (n) HIP_TRY(hipMemcpy(p, host, n * sizeof(T), hipMemcpyHostToDevice));
    s->scene_bytes += bytes;
    *out = static_cast<const T*>(p);
    return RT_OK;
}

static int validate(const rt_flat_scene* sc) {
    if (!sc) { rt_set_error("scene is NULL"); return RT_ERR_INVALID; }
    if (sc->n_objects && !sc->objects) { rt_set_error("objects is NULL"); return RT_ERR_INVALID; }
    if (sc->n_faces > RT_LEAF_START_MASK) { rt_set_error("n_faces %u above the maximum %u", sc->n_faces, RT_LEAF_START_MASK); return RT_ERR_LIMIT; }
    if (sc->n_faces && (!sc->v0 || !sc->v1 || !sc->v2 || !sc->n0 || !sc->n1 || !sc->n2)) { rt_set_error("triangle arrays missing"); return RT_ERR_INVALID; }
    for (uint32_t i = 0; i < sc->n_objects; ++i) {
        const rt_object& o = sc->objects[i];
        const uint32_t lim = o.type == RT_OBJ_SPHERE ? sc->n_spheres : o.type == RT_OBJ_PLANE ? sc->n_planes
                             : o.type == RT_OBJ_MODEL ? sc->n_models : 0;
        if (o.type < 0 || o.type > RT_OBJ_MODEL || o.index >= lim) { rt_set_error("object %u: bad type/index", i); return RT_ERR_INVALID; }
        if (o.material >= sc->n_materials) { rt_set_error("object %u: material %u out of range", i, o.material); return RT_ERR_INVALID; }
    }
    for (uint32_t m = 0; m < sc->n_models; ++m)
        if (uint64_t(sc->models[m].first_face) + sc->models[m].num_faces > sc->n_faces) { rt_set_error("model %u: face range out of bounds", m); return RT_ERR_INVALID; }
    return RT_OK;
}

extern "C" int rt_scene_create(const rt_flat_scene* sc, int device, rt_scene** out) { return rt_scene_create_ex(sc, device, 0u, out); }

extern "C" int rt_scene_create_ex(const rt_flat_scene* sc, int device, uint32_t flags, rt_scene** out) {
    if (!out) { rt_set_error("out is NULL"); return RT_ERR_INVALID; }
    *out = nullptr;
    int rc = validate(sc);
    if (rc != RT_OK) return rc;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { rt_set_error("no HIP device available"); return RT_ERR_NO_DEVICE; }
    if (device < 0 || device >= ndev) { rt_set_error("device %d out of range (%d devices)", device, ndev); return RT_ERR_INVALID; }
    HIP_TRY(hipSetDevice(device));

    BvhBuildResult bvh;
    std::memset(&bvh, 0, sizeof bvh);
    DeviceBvh dbvh;
    bool on_device = false;
    const auto t_build = std::chrono::steady_clock::now();
    if ((flags & RT_SCENE_BVH_DEVICE) && sc->n_faces) {
        const int brc = bvh_build_device(sc, &dbvh);
        if (brc < 0) return RT_ERR_DEVICE;
        on_device = brc == 0;
    }
    if (!on_device && bvh_build(sc, &bvh) != 0) { rt_set_error("BVH build failed"); return RT_ERR_INVALID; }
    const double build_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_build).count();

    rt_scene* s = new rt_scene();
    s->device = device;
    s->knobs = knobs_from_env();         // the only place the environment is read
    auto fail = [&](int code) { bvh_free(&bvh); hipFree(dbvh.nodes); hipFree(dbvh.leaves); free_scene(s); return code; };
    // (no stream priorities: low-priority shade streams gained ~1.5 % but hung hipDeviceSynchronize when two processes
    // shared one GPU)
    // (HIP deals streams to the process's hardware queues - 4 - in creation order, and the frame time depends on which
    // of these streams share a queue: with 1 / 2 / 3 other streams created between two scenes a batched 1080p frame took
    // 0.356 / 0.433 / 0.362 ms instead of 0.301. Create a process's scenes back to back, before other stream users: see
    // INTEGRATION.md)
    if (hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&s->shade_stream[0], hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&s->shade_stream[1], hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&s->shade_stream[2], hipStreamNonBlocking) != hipSuccess) { rt_set_error("hipStreamCreate failed"); return fail(RT_ERR_DEVICE); }
    s->n_shade_streams = s->knobs.shade_streams;
    for (hipEvent_t& e : s->ev_sync)
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { rt_set_error("hipEventCreate failed"); return fail(RT_ERR_DEVICE); }

    std::vector<uint4> objs(size_t(sc->n_objects) * 2 + 2, make_uint4(0, 0, 0, 0));      // + one entry of zeros: the object loops fetch entry k + 1 while they work on k
    for (uint32_t i = 0; i < sc->n_objects; ++i) {
        const rt_object& o = sc->objects[i];
        float prm[4] = {0, 0, 0, 0}, inv_r = 0;
        if (o.type == RT_OBJ_SPHERE) {
            const rt_sphere& sp = sc->spheres[o.index];
            prm[0] = sp.center[0]; prm[1] = sp.center[1]; prm[2] = sp.center[2]; prm[3] = sp.radius;
            inv_r = float(1.0 / double(sp.radius));      // reference src/container.h:72-75 (Sphere::getNormal's n /= radius)
        } else if (o.type == RT_OBJ_PLANE) {
            const rt_plane& pl = sc->planes[o.index];
            prm[0] = pl.a; prm[1] = pl.b; prm[2] = pl.c; prm[3] = pl.d;
        }
        uint32_t bits[5];
        std::memcpy(bits, prm, sizeof prm);
        std::memcpy(bits + 4, &inv_r, sizeof inv_r);
        objs[2 * i] = make_uint4(uint32_t(o.type), o.index, o.material, bits[4]);
        objs[2 * i + 1] = make_uint4(bits[0], bits[1], bits[2], bits[3]);
        const rt_material& m = sc->materials[o.material];
        if (m.reflective || m.refractive) s->has_bounce = true;
    }
    std::vector<float4> li(size_t(sc->n_lights) * 2);
    for (uint32_t i = 0; i < sc->n_lights; ++i) {
        li[2 * i] = make_float4(sc->lights[i].pos[0], sc->lights[i].pos[1], sc->lights[i].pos[2], 0);
        li[2 * i + 1] = make_float4(sc->lights[i].color[0], sc->lights[i].color[1], sc->lights[i].color[2], 0);
    }
    std::vector<DModel> models(sc->n_models);
    for (uint32_t m = 0; m < sc->n_models; ++m) {
        DModel& d = models[m];
        d.root = on_device ? dbvh.model_root[m] : bvh.model_root[m];
        d.first_face = sc->models[m].first_face;
        d.num_faces = sc->models[m].num_faces;
        std::memcpy(d.bb_center, sc->models[m].bb_center, sizeof d.bb_center);
        d.bb_radius = sc->models[m].bb_radius;
        d.pad = 0;
    }
    {   // the box the sorted light loops cut into cells: the models' bounding spheres, or their vertices where a model has none
        float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
        for (uint32_t m = 0; m < sc->n_models; ++m) {
            const rt_model& md = sc->models[m];
            if (md.bb_radius >= 0 && std::isfinite(md.bb_radius)) {
                for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], md.bb_center[k] - md.bb_radius); hi[k] = std::max(hi[k], md.bb_center[k] + md.bb_radius); }
            } else {
                for (uint64_t f = md.first_face; f < uint64_t(md.first_face) + md.num_faces; ++f)
                    for (const float* v : {sc->v0 + 3 * f, sc->v1 + 3 * f, sc->v2 + 3 * f})
                        for (int k = 0; k < 3; ++k) { if (v[k] < lo[k]) lo[k] = v[k]; if (v[k] > hi[k]) hi[k] = v[k]; }
            }
        }
        s->sort_box_ok = sc->n_models > 0;
        for (int k = 0; k < 3; ++k) {
            const float ext = hi[k] - lo[k];
            if (!(ext > 0) || !std::isfinite(ext) || !std::isfinite(lo[k])) { s->sort_box_ok = false; break; }
            s->sort_box.lo[k] = lo[k];
            s->sort_box.scale[k] = 256.0f / ext;
        }
    }
    std::vector<TriNormals> nrm(sc->n_faces);
    for (uint32_t f = 0; f < sc->n_faces; ++f) {
        TriNormals& t = nrm[f];
        for (int k = 0; k < 3; ++k) { t.n0[k] = sc->n0[3 * size_t(f) + k]; t.n1[k] = sc->n1[3 * size_t(f) + k]; t.n2[k] = sc->n2[3 * size_t(f) + k]; }
        t.pad0 = t.pad1 = t.pad2 = 0;
    }
    s->d.n_objects = sc->n_objects;
    s->d.generic_pow = 0;
    for (uint32_t m = 0; m < sc->n_materials; ++m) {
        const float e = sc->materials[m].specRefExp;
        if (!(e >= 0.0f && e <= float(RT_POW_MAX_INT) && float(int(e)) == e)) s->d.generic_pow = 1;
    }
    s->d.n_lights = sc->n_lights;
    // Level 0 as packets (packet_traverse). Measured faster even where triangles are much smaller than pixels (10 M
    // faces at 1080p 4 spp: 7.4 -> 6.4 ms/frame). Option "packet" (rt_scene_set_option) fixes the choice.
    s->packet_auto = true;
    s->d.packet = sc->n_models > 0 ? 1u : 0u;
    if (s->knobs.packet >= 0) { s->d.packet = sc->n_models > 0 ? uint32_t(s->knobs.packet) : 0u; s->packet_auto = false; }
    if (flags & RT_SCENE_NO_PACKETS) { s->d.packet = 0; s->packet_auto = false; s->no_packets_flag = true; }
    s->d.user_packet = 0;
    s->pass_samples = default_pass_samples(s->knobs.pass_log2);
    // set-up for rt_render_device_noalloc: the address of the static workspace ON THIS DEVICE (current since hipSetDevice
    // above), asked once per device and kept; the entry point itself then allocates and looks up nothing. A device on
    // which this fails (index beyond the table, no symbol) still gets its scene: only that entry point then reports it.
    if (hipError_t e = noalloc_scratch(device, &s->noalloc_ws); e != hipSuccess) { s->noalloc_ws = nullptr; (void)hipGetLastError(); }
    s->aa_cache.reserve(2 * RT_MAX_AA_POINTS);      // so that a new anti-aliasing pattern never allocates
    if ((rc = upload(s, objs.data(), objs.size(), &s->d.objects)) != RT_OK) return fail(rc);
    if ((rc = upload(s, models.data(), models.size(), &s->d.models)) != RT_OK) return fail(rc);
    if ((rc = upload(s, sc->materials, size_t(sc->n_materials), &s->d.materials)) != RT_OK) return fail(rc);
    if ((rc = upload(s, li.data(), li.size(), &s->d.lights)) != RT_OK) return fail(rc);
    if (on_device) {        // the builder's arrays are the scene's
        s->owned.push_back(dbvh.nodes); s->owned.push_back(dbvh.leaves);
        s->d.nodes = dbvh.nodes; s->d.leaves = dbvh.leaves;
        s->scene_bytes += sizeof(Bvh4Node) * size_t(dbvh.n_nodes) + sizeof(LeafBlock) * size_t(dbvh.n_leaves);
        s->n_nodes = dbvh.n_nodes; s->n_leaves = dbvh.n_leaves; s->bvh_depth = dbvh.max_depth; s->bvh_stack = dbvh.max_stack;
        dbvh.nodes = nullptr; dbvh.leaves = nullptr;
    } else {
        if ((rc = upload(s, bvh.nodes, size_t(bvh.n_nodes), &s->d.nodes)) != RT_OK) return fail(rc);
        if ((rc = upload(s, bvh.leaves, size_t(bvh.n_leaves), &s->d.leaves)) != RT_OK) return fail(rc);
        s->n_nodes = bvh.n_nodes; s->n_leaves = bvh.n_leaves; s->bvh_depth = bvh.max_depth; s->bvh_stack = bvh.max_stack;
    }
    if ((rc = upload(s, nrm.data(), nrm.size(), &s->d.normals)) != RT_OK) return fail(rc);
    s->n_models = sc->n_models;
    s->n_faces = sc->n_faces;
    s->built_on_device = on_device;
    s->build_seconds = build_seconds;
    s->model_roots.resize(sc->n_models);
    for (uint32_t m = 0; m < sc->n_models; ++m) s->model_roots[m] = models[m].root;
    bvh_free(&bvh);

    if (hipMalloc(reinterpret_cast<void**>(&s->d_counters), sizeof(PassCounters) * RT_MAX_PASSES) != hipSuccess ||
        hipHostMalloc(reinterpret_cast<void**>(&s->h_counters), sizeof(PassCounters) * RT_MAX_PASSES) != hipSuccess ||
        hipHostMalloc(reinterpret_cast<void**>(&s->h_hint), sizeof(ShardCounters) * RT_SHARDS * RT_MAX_PASSES) != hipSuccess ||
        hipHostMalloc(reinterpret_cast<void**>(&s->h_overflow), 64) != hipSuccess ||
        hipHostMalloc(reinterpret_cast<void**>(&s->h_tail_hint), sizeof(uint32_t) * (RT_MAX_LEVELS + 4) * RT_MAX_PASSES) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&s->d_fp), sizeof(FrameParams) * RT_MAX_BATCH) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&s->d_aa), sizeof(float) * 2 * (RT_MAX_AA_POINTS + 1)) != hipSuccess ||
        hipEventCreate(&s->ev_begin) != hipSuccess || hipEventCreate(&s->ev_end) != hipSuccess) {
        rt_set_error("workspace allocation failed: %s", hipGetErrorString(hipGetLastError()));
        free_scene(s);
        return RT_ERR_DEVICE;
    }
    std::memset(s->h_hint, 0, sizeof(ShardCounters) * RT_SHARDS * RT_MAX_PASSES);
    std::memset(s->h_tail_hint, 0, sizeof(uint32_t) * (RT_MAX_LEVELS + 4) * RT_MAX_PASSES);
    *s->h_overflow = 0;
    for (int p = 0; p < RT_MAX_PASSES; ++p)
        for (int i = 0; i < RT_MAX_LEVELS + 4; ++i) s->h_hint[size_t(p) * RT_SHARDS].level[i] = 0xFFFFFFFFu;   // unknown: assume large levels
    *out = s;
    return RT_OK;
}

extern "C" void rt_scene_destroy(rt_scene* s) { free_scene(s); }

static void learn_from_overflow(rt_scene* s, uint32_t full_px);

extern "C" int rt_scene_status(rt_scene* s) {
    if (!s) { rt_set_error("null scene"); return RT_ERR_INVALID; }
    HIP_TRY(hipSetDevice(s->device));
    if (s->last_stream) HIP_TRY(hipStreamSynchronize(s->last_stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    if (s->copy_stream) HIP_TRY(hipStreamSynchronize(s->copy_stream));
    const bool over = *s->h_overflow != 0 || s->last_overflow == 1;
    // a frame in the static pool that overflowed although its passes were already 64 pixels has nowhere to go
    const bool at_floor = over && s->last_noalloc &&
        (std::max<uint32_t>(64u, RT_NOALLOC_PASS_SAMPLES / std::max<uint32_t>(1u, s->last_spp)) >> s->noalloc_shrink_log2) <= 64u;
    if (*s->h_overflow && s->last_overflow != 1) learn_from_overflow(s, std::max<uint32_t>(64u, RT_PASS_SAMPLES / std::max<uint32_t>(1u, s->last_spp)));   // not yet accounted for
    *s->h_overflow = 0;
    s->last_overflow = 0;
    if (over) {
        if (at_floor) rt_set_error("an asynchronous frame ran out of ray pool in the static workspace even with 64-pixel passes: use rt_render_device, whose pool can grow");
        else rt_set_error("an asynchronous frame ran out of ray pool; the scene now uses smaller passes: render it again");
        return RT_ERR_OVERFLOW;
    }
    return RT_OK;
}

extern "C" void* rt_scene_stream(const rt_scene* s) { return s ? static_cast<void*>(s->stream) : nullptr; }

extern "C" int rt_scene_set_pass_samples(rt_scene* s, uint32_t samples) {
    if (!s) { rt_set_error("null scene"); return RT_ERR_INVALID; }
    if (samples == 0) { HIP_TRY(hipSetDevice(s->device)); s->pass_samples = default_pass_samples(s->knobs.pass_log2); s->shrink_log2 = 0; return RT_OK; }
    if (samples < (1u << RT_PASS_LOG2_MIN) || samples > (1u << RT_PASS_LOG2_MAX) || (samples & (samples - 1u))) {
        rt_set_error("pass size %u is not a power of two in [2^%d, 2^%d]", samples, RT_PASS_LOG2_MIN, RT_PASS_LOG2_MAX);
        return RT_ERR_LIMIT;
    }
    HIP_TRY(hipSetDevice(s->device));
    if (s->last_stream) HIP_TRY(hipStreamSynchronize(s->last_stream));      // frames in flight keep their passes
    HIP_TRY(hipStreamSynchronize(s->stream));
    s->pass_samples = samples;
    s->shrink_log2 = 0;
    return RT_OK;
}

extern "C" uint32_t rt_scene_pass_samples(const rt_scene* s) { return s ? s->pass_samples : 0u; }

extern "C" int rt_scene_set_option(rt_scene* s, const char* name, int64_t value) {
    if (!s || !name) { rt_set_error("null argument"); return RT_ERR_INVALID; }
    HIP_TRY(hipSetDevice(s->device));
    if (s->last_stream) HIP_TRY(hipStreamSynchronize(s->last_stream));      // frames in flight keep their schedule
    HIP_TRY(hipStreamSynchronize(s->stream));
    Knobs& k = s->knobs;
    const std::string n(name);
    if (n == "packet") {
        if (value < -1 || value > 2) { rt_set_error("option packet: -1 (automatic), 0, 1 or 2"); return RT_ERR_INVALID; }
        k.packet = int(value);
        if (s->no_packets_flag) return RT_OK;      // RT_SCENE_NO_PACKETS stands
        s->packet_auto = value < 0;
        s->d.packet = s->n_models ? uint32_t(value < 0 ? 1 : value) : 0u;
    }
    else if (n == "large_threshold") k.large_thr = value < 0 ? -1 : value;
    else if (n == "large_max_level") k.large_maxlv = value < 0 ? -1 : int(std::min<int64_t>(value, RT_MAX_LEVELS));
    else if (n == "serve") k.serve = value < 0 ? -1 : (value ? 1 : 0);
    else if (n == "split_small") k.split_small = value ? 1 : 0;
    else if (n == "order0") { k.order0 = value ? 1 : 0; s->oset[0].n = s->oset[1].n = 0; }
    else if (n == "shade_serve_level") k.shade_serve_lv = int(value);
    else if (n == "sort_shade") k.sort_shade = std::max<int64_t>(0, value);
    else if (n == "sort_bits") k.sort_bits = int(std::max<int64_t>(1, std::min<int64_t>(10, value)));
    else if (n == "fuse") k.fuse = value ? 1 : 0;
    else if (n == "run_hits") k.run_hits = int(std::max<int64_t>(0, std::min<int64_t>(value, 1 << 30)));
    else if (n == "tiny_run") k.tiny_run = int(std::max<int64_t>(0, std::min<int64_t>(value, 1 << 30)));
    else if (n == "tail_rays") k.tail = std::max<int64_t>(0, value);
    else if (n == "index64") k.index64 = value ? 1 : 0;
    else if (n == "chunk_log2") k.chunk_log2 = int(std::min<int64_t>(RT_SHARD_CHUNK_LOG2_MAX, std::max<int64_t>(0, value)));
    else if (n == "graph") k.use_graph = value ? 1 : 0;
    else { rt_set_error("unknown option '%s'", name); return RT_ERR_INVALID; }
    return RT_OK;
}

extern "C" int rt_scene_bvh_info(const rt_scene* s, rt_bvh_info* out) {
    if (!s || !out) { rt_set_error("null argument"); return RT_ERR_INVALID; }
    out->n_nodes = s->n_nodes; out->n_faces = s->n_faces; out->max_depth = s->bvh_depth;
    out->node_bytes = sizeof(Bvh4Node); out->tri_bytes = sizeof(LeafBlock) / 4; out->built_on_device = s->built_on_device ? 1u : 0u;
    out->device_bytes = s->scene_bytes;
    out->build_seconds = s->build_seconds;
    return RT_OK;
}

extern "C" int rt_scene_bvh_check(const rt_scene* s, const rt_flat_scene* flat, rt_bvh_check_result* out) {
    if (!s || !flat || !out) { rt_set_error("null argument"); return RT_ERR_INVALID; }
    if (flat->n_faces != s->n_faces || flat->n_models != s->n_models) { rt_set_error("not the scene this rt_scene was created from"); return RT_ERR_INVALID; }
    std::memset(out, 0, sizeof *out);
    HIP_TRY(hipSetDevice(s->device));
    std::vector<Bvh4Node> nodes(std::max<uint32_t>(1, s->n_nodes));
    std::vector<LeafBlock> leaves(std::max<uint32_t>(1, s->n_leaves));
    if (s->n_nodes) HIP_TRY(hipMemcpy(nodes.data(), s->d.nodes, sizeof(Bvh4Node) * size_t(s->n_nodes), hipMemcpyDeviceToHost));
    if (s->n_leaves) HIP_TRY(hipMemcpy(leaves.data(), s->d.leaves, sizeof(LeafBlock) * size_t(s->n_leaves), hipMemcpyDeviceToHost));
    std::vector<uint32_t> roots = s->model_roots;
    BvhBuildResult r;
    r.nodes = nodes.data(); r.n_nodes = s->n_nodes; r.leaves = leaves.data(); r.n_leaves = s->n_leaves;
    r.model_root = roots.data(); r.max_depth = s->bvh_depth; r.max_stack = s->bvh_stack;
    bvh_walk_check(flat, r, out);
    out->build_seconds = s->build_seconds;
    return RT_OK;
}

extern "C" int rt_set_count_work(rt_scene* s, int enable) {
    if (!s) { rt_set_error("null scene"); return RT_ERR_INVALID; }
    s->count_work = enable != 0;
    return RT_OK;
}
static void add_work(rt_work& w, const unsigned long long* c) {
    w.nodes_visited += c[0]; w.tris_tested += c[1]; w.spheres_tested += c[2]; w.planes_tested += c[3];
    w.wave_node_steps += c[4]; w.wave_tri_steps += c[5];
    w.trav_ticks_sum += c[6]; w.trav_ticks_max = std::max<uint64_t>(w.trav_ticks_max, c[7]);
    w.batches += c[8]; w.trav_waves += c[9];
    w.wave_ticks_sum += c[10]; w.wave_ticks_max = std::max<uint64_t>(w.wave_ticks_max, c[11]);
    w.wave_cycles_sum += c[12];
}
extern "C" int rt_get_work(const rt_scene* s, int kernel, rt_work* out) {
    if (!s || !out || kernel < -1 || kernel >= RT_N_KINDS) { rt_set_error("bad argument"); return RT_ERR_INVALID; }
    if (kernel >= 0) { *out = s->last_work[kernel]; return RT_OK; }
    *out = rt_work{};
    for (const rt_work& a : s->last_work) {
        out->nodes_visited += a.nodes_visited; out->tris_tested += a.tris_tested;
        out->spheres_tested += a.spheres_tested; out->planes_tested += a.planes_tested;
        out->wave_node_steps += a.wave_node_steps; out->wave_tri_steps += a.wave_tri_steps;
        out->trav_ticks_sum += a.trav_ticks_sum; out->trav_ticks_max = std::max(out->trav_ticks_max, a.trav_ticks_max);
        out->batches += a.batches; out->trav_waves += a.trav_waves;
        out->wave_ticks_sum += a.wave_ticks_sum; out->wave_ticks_max = std::max(out->wave_ticks_max, a.wave_ticks_max);
        out->wave_cycles_sum += a.wave_cycles_sum;
    }
    return RT_OK;
}

// ---------------------------------------------------------------------------------------------------------
// frame schedule
// ---------------------------------------------------------------------------------------------------------
static int ensure_workspace(rt_scene* s, uint32_t n0) {
    if (n0 <= s->ws_n0) return RT_OK;
    hipFree(s->node_a); hipFree(s->node_b); hipFree(s->rays); hipFree(s->hit_n); hipFree(s->hit_p); hipFree(s->queue_e); hipFree(s->queue_s);
    s->node_a = nullptr; s->node_b = nullptr; s->rays = nullptr; s->hit_n = nullptr; s->hit_p = nullptr;
    s->queue_e = nullptr; s->queue_s = nullptr; s->ws_n0 = 0; s->pool_cap = 0; s->sub_cap = 0;
    const uint64_t cap = uint64_t(n0) * (s->has_bounce ? RT_POOL_FACTOR : 1u);
    if (cap > 0xFFFFFFF0ull) { rt_set_error("pass too large"); return RT_ERR_LIMIT; }
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&s->node_a), cap * sizeof(float4)));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&s->node_b), cap * sizeof(uint4)));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&s->hit_n), cap * sizeof(float4)));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&s->hit_p), cap * sizeof(float4)));
    if (s->n_models) {
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&s->queue_e), cap * sizeof(uint2)));
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&s->queue_s), cap * sizeof(uint2)));
    }
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&s->rays), std::max<uint64_t>(cap - n0, 1) * 2 * sizeof(float4)));
    s->ws_n0 = n0;
    s->pool_cap = uint32_t(cap);
    s->sub_cap = uint32_t((cap - n0) / RT_SHARDS);
    return RT_OK;
}

static DCamera to_dcamera(const rt_camera* c) {
    DCamera d;
    std::memcpy(d.eye, c->eye, sizeof d.eye); std::memcpy(d.up, c->up, sizeof d.up);
    std::memcpy(d.right, c->right, sizeof d.right); std::memcpy(d.at, c->at, sizeof d.at);
    d.aspect = c->aspectRatio; d.imWidth = c->imWidth; d.imHeight = c->imHeight; d.pad = 0;
    // deg2rad (reference src/helper.cpp:13-15) returns Float; `* 0.5` and tan are double (src/camera.h:60)
    const float rad = float(double(c->fov) * M_PI / 180);
    d.tan_half = std::tan(double(rad) * 0.5);
    return d;
}

static int check_params(const rt_camera* cam, const rt_params* p) {
    if (!cam || !p) { rt_set_error("camera/params is NULL"); return RT_ERR_INVALID; }
    if (p->max_depth < 0 || p->max_depth >= RT_MAX_LEVELS) { rt_set_error("max_depth %d outside [0,%d]", p->max_depth, RT_MAX_LEVELS - 1); return RT_ERR_LIMIT; }
    if (p->n_aa > RT_MAX_AA_POINTS) { rt_set_error("n_aa %u above the maximum %d", p->n_aa, RT_MAX_AA_POINTS); return RT_ERR_LIMIT; }
    if (p->n_aa && !p->aa_xy) { rt_set_error("aa_xy is NULL"); return RT_ERR_INVALID; }
    if (cam->imWidth == 0 || cam->imHeight == 0) { rt_set_error("empty image"); return RT_ERR_INVALID; }
    return RT_OK;
}

struct FrameJob {
    rt_scene* s;
    hipStream_t st;
    LevelArgs base;          // everything that is constant over the frame
    uint8_t* d_out;
    bool timed;              // per-launch events, everything on one stream
    uint32_t n_timed;
    std::vector<uint8_t> timed_kind;   // RT_K_*
    float ms_by[RT_N_KINDS];           // drained event pairs, summed per kind
    uint32_t launches_by[RT_N_KINDS];
    uint32_t n_pass_used;
    bool noalloc;            // the pool is the static array
    uint32_t slots_cleared;  // the first so many PassCounters slots were cleared at the start of the frame
    uint32_t large_mask;     // bit k: level k runs as per-lane kernel + queue (expect_large at the start of the frame)
    rt_progress_fn progress; // synchronous frames: called as the passes complete (rt_render_progress)
    void* progress_user;
};

// Waits for the stream and adds the recorded event pairs to the job's per-kind sums; the events are then reused.
static int drain_timed(FrameJob& J, hipStream_t st) {
    if (!J.n_timed) return RT_OK;
    HIP_TRY(hipStreamSynchronize(st));
    for (uint32_t i = 0; i < J.n_timed; ++i) {
        float t = 0;
        HIP_TRY(hipEventElapsedTime(&t, J.s->ev_pool[2 * i], J.s->ev_pool[2 * i + 1]));
        J.ms_by[J.timed_kind[i]] += t;
        J.launches_by[J.timed_kind[i]] += 1;
    }
    J.n_timed = 0;
    J.timed_kind.clear();
    return RT_OK;
}

// One kernel launch, bracketed by events when the frame is timed.
template <typename F>
static int timed_launch(FrameJob& J, hipStream_t st, uint8_t kind, F&& launch) {
    rt_scene* s = J.s;
    hipEvent_t e1 = nullptr;
    if (J.timed && J.n_timed >= RT_MAX_TIMED) { const int rc = drain_timed(J, st); if (rc != RT_OK) return rc; }
    if (J.timed) {
        while (s->ev_pool.size() < size_t(2 * (J.n_timed + 1))) {
            hipEvent_t e;
            HIP_TRY(hipEventCreate(&e));
            s->ev_pool.push_back(e);
        }
        HIP_TRY(hipEventRecord(s->ev_pool[2 * J.n_timed], st));
        e1 = s->ev_pool[2 * J.n_timed + 1];
        J.timed_kind.push_back(kind);
        ++J.n_timed;
    }
    launch();
    if (e1) HIP_TRY(hipEventRecord(e1, st));
    return RT_OK;
}

// Is a level (or range) expected to hold enough rays to fill the chip one per lane? Sizes come from the pass with the
// same ordinal of the scene's previous frame (h_hint: the passes of a frame differ - the floor at the bottom of the
// picture bounces nothing, the glass in the middle a lot - but pass k of the next frame looks like pass k of this
// one); unknown = large.
static uint64_t level_hint(const rt_scene* s, uint32_t pass, int l) {
    if (l > s->hint_depth) return 0xFFFFFFFFull;      // the frame the hints come from did not go that deep: unknown, not empty
    if (const uint32_t t = s->h_tail_hint[size_t(pass) * (RT_MAX_LEVELS + 4) + l]) return t;      // the level was part of a tail
    uint64_t n = 0;
    const ShardCounters* h = s->h_hint + size_t(pass) * RT_SHARDS;
    for (uint32_t k = 0; k < RT_SHARDS; ++k) n += h[k].level[l];
    return n;
}
static bool expect_large(const rt_scene* s, uint32_t pass, int lo, int hi) {
    if (lo == 0) return true;
    // Measured (tools/exp_thr.sh): with level 0 traversed as packets, the bounce levels are fastest as ONE quad launch
    // per level whatever their size (bunny_1080p 0.389 -> 0.362 ms, 10 M-triangle field at 1080p 6.04 -> 5.48 ms); where
    // the deeper levels use packets too (16+ spp) a level pays for the per-lane kernels from about a million rays on
    // (field at 8K 64 spp: threshold 131 k 829 ms, 1 M 727 ms, 4 M 818 ms; 4K glass scene flat up to 1 M, 73 ms at 4 M).
    // Packets pay while the 64 samples of a block still travel together, and that wears off with every bounce: level 1
    // from a million rays on, level 2 from 8 M, deeper levels never (tools/exp_c5_pass.sh at 2^26-sample passes. The
    // field at 8K 64 spp, every level from 1 M on: 520 ms; level 1 only: 468; levels 1 and 2 from 1 M: 525, level 2 from
    // 8 M: 470; no packets below level 0: 603. The 4K glass scene, whose level 2 holds 9 M rays per pass: every level
    // 29.7 ms, level 1 only 29.7, levels 1 and 2 28.9.)
    const bool fixed = s->knobs.large_thr >= 0;       // option "large_threshold" (experiments and tests): one threshold for the levels up to "large_max_level"
    const int max_lv = s->knobs.large_maxlv >= 0 ? s->knobs.large_maxlv : (fixed || s->d.packet < 2) ? RT_MAX_LEVELS : 2;
    for (int l = lo; l <= hi && l <= max_lv; ++l) {
        const uint64_t thr = fixed ? uint64_t(s->knobs.large_thr) : s->d.packet >= 2 ? (l == 1 ? (1u << 20) : (8u << 20)) : s->d.packet == 1 ? ~0ull : 16u * RT_LEVEL_GRID;
        if (level_hint(s, pass, l) > thr) return true;
    }
    return false;
}

// Workgroups for a quad launch over the small levels lo..hi. The kernels spread a level's rays over at most 4096 waves
// (quads_per_wave: as few rays per wave as that allows, the launch is latency-bound) and deal the groups to workgroups
// in rounds of 2048 (group_of: runs of 8 per CU), so the launch needs the expected groups rounded up to 2048 and no
// more: a workgroup that finds nothing to do still costs ~280 instructions (shard view, group scan), and 9 x 8192 of
// them were 9 % of a 1080p frame's instructions. Expected = the count of the previous frame's pass plus a quarter; the
// kernels stride over the device-side count, so a short hint costs time, never rays.
static uint32_t small_grid(const rt_scene* s, uint32_t pass, int lo, int hi) {
    uint64_t n = 0;
    for (int l = lo; l <= hi; ++l) n += level_hint(s, pass, l);
    const uint64_t cap = RT_LEVEL_GRID;       // (16 k and 32 k measured the same, 64 k slower)
    const uint64_t want = n + n / 4u + 2u * RT_SHARDS;
    const uint64_t groups = std::max<uint64_t>(std::min<uint64_t>(4096u + 2u * RT_SHARDS, want), want / 16u);
    return uint32_t(std::min<uint64_t>(cap, (groups + 2047u) / 2048u * 2048u));
}

// Workgroups for the quad launch that resumes what k_shade_wide queued at level `lv` (queue_s): sized like small_grid
// from the queue length the same pass of the previous frame reached; unknown = the full grid.
static uint32_t resume_grid(const rt_scene* s, uint32_t pass, int lv) {
    const ShardCounters* h = s->h_hint + size_t(pass) * RT_SHARDS;
    if (h[0].level[lv] == 0xFFFFFFFFu) return 4u * RT_LEVEL_GRID;
    uint64_t n = 0;
    for (uint32_t k = 0; k < RT_SHARDS; ++k) n += h[k].qs[lv];
    const uint64_t want = n + n / 4u + 2u * RT_SHARDS;
    const uint64_t groups = std::max<uint64_t>(std::min<uint64_t>(4096u + 2u * RT_SHARDS, want), want / 16u);
    return uint32_t(std::min<uint64_t>(4u * RT_LEVEL_GRID, (groups + 2047u) / 2048u * 2048u));
}

// Workgroups of a per-lane launch over levels lo..hi (>= 1): one per 64 entries expected.
static uint32_t wide_grid_for(const rt_scene* s, uint32_t pass, int lo, int hi) {
    uint64_t n = 0;
    for (int l = lo; l <= hi; ++l) n += level_hint(s, pass, l);
    return uint32_t(std::min<uint64_t>(uint64_t(s->knobs.wide_grid), (n + n / 4u) / 64u + 2u * RT_SHARDS));
}

// Arrays of the expensive-groups-first order for at least n groups (grow-only). false: no memory - pool order then.
static bool ensure_order_buf(rt_scene* s, uint32_t n, hipStream_t st) {
    if (s->order_cap >= n && s->oset[0].ev && s->oset[1].ev) return true;
    if (s->order_cap) (void)hipDeviceSynchronize();      // (a frame in flight may still use the old ones)
    auto release = [&] {
        for (auto& o : s->oset)
            for (int k = 0; k < 2; ++k) { hipFree(o.cost[k]); hipFree(o.order[k]); o.cost[k] = o.order[k] = nullptr; }
    };
    release();
    s->order_cap = 0;
    const uint32_t cap = std::max<uint32_t>(n, 1u << 15);
    bool ok = true;
    for (auto& o : s->oset) {
        o.n = 0; o.pending = false;
        ok = ok && (o.ev || hipEventCreateWithFlags(&o.ev, hipEventDisableTiming) == hipSuccess);
        for (int k = 0; k < 2; ++k) {      // (costs: 4 words per group, cleared once here and then by k_order_groups)
            ok = ok && hipMalloc(reinterpret_cast<void**>(&o.cost[k]), size_t(cap) * 16u) == hipSuccess;
            // (on the frame's stream: a plain hipMemset runs on the null stream, which the scene's non-blocking streams do not wait for)
            ok = ok && hipMemsetAsync(o.cost[k], 0, size_t(cap) * 16u, st) == hipSuccess;
            ok = ok && hipMalloc(reinterpret_cast<void**>(&o.order[k]), size_t(RT_ORDER_SLOTS(cap)) * 4u + cap) == hipSuccess;      // + k_order_groups' keys
        }
    }
    if (!ok) {
        (void)hipGetLastError();
        release();
        return false;
    }
    s->order_cap = cap;
    return true;
}

// Buffers of the sorted light loops for shade stream `which`, at least n_slots slots (grow-only). false: no memory - the
// launch then walks the pool order.
static bool ensure_sort_buf(rt_scene* s, int which, uint32_t n_slots) {
    rt_scene::SortBuf& b = s->sort_buf[which];
    if (b.cap >= n_slots) return true;
    const uint32_t cap = std::max<uint32_t>(n_slots + n_slots / 4u, 1u << 16) & ~63u;
    const size_t tb = sort_hits_temp_bytes(cap);
    if (!tb) return false;
    if (b.cap) (void)hipDeviceSynchronize();      // (a frame in flight may still read the old ones)
    hipFree(b.keys[0]); hipFree(b.keys[1]); hipFree(b.vals[0]); hipFree(b.vals[1]); hipFree(b.temp);
    b = rt_scene::SortBuf{};
    bool ok = true;
    for (int k = 0; k < 2; ++k) {
        ok = ok && hipMalloc(reinterpret_cast<void**>(&b.keys[k]), size_t(cap) * 4u) == hipSuccess;
        ok = ok && hipMalloc(reinterpret_cast<void**>(&b.vals[k]), size_t(cap) * 4u) == hipSuccess;
    }
    ok = ok && hipMalloc(&b.temp, tb) == hipSuccess;
    if (!ok) {
        (void)hipGetLastError();
        hipFree(b.keys[0]); hipFree(b.keys[1]); hipFree(b.vals[0]); hipFree(b.vals[1]); hipFree(b.temp);
        b = rt_scene::SortBuf{};
        return false;
    }
    b.temp_bytes = tb; b.cap = cap;
    return true;
}

// Enqueue the passes covering pixels [px0, px0+npx) of the frame's pixel order, pass_px pixels at a time.
// Records which PassCounters slot each pass used in `slots` (px0, npx, slot).
//
// One recursion level = closest hits (extend) then, independently, the light loop of those hits (shade).
//   large level (or a scene without meshes):
//     Ew   per-lane extend: finishes the rays that never reach a mesh, parks the others in the level's queue
//     Eq   quad extend: resumes the parked rays                                      (scenes with meshes only)
//     Sw   per-lane shade of the hits Ew finished; parks the hits whose shadow rays reach a mesh
//     Sq   quad shade: resumes what Sw parked
//     Sp   quad shade of the hits Eq produced
//   small level: Ef = quad extend of all its rays; small levels are shaded together at the end by Sf (quad).
// Streams (an untimed frame): the extend kernels form the dependent chain and run on the caller's stream A; the
// shade kernels feed nothing but the fold, so level k's run on shade stream k%2 as soon as their input exists
// (Sw, Sq after Ew; Sp after Eq), next to Eq and the extend kernels of the deeper levels. A waits for both shade
// streams before folding. A timed frame (stats requested) issues the same launches on ONE stream, so that each
// kernel's duration is its own.
struct PassRec { uint64_t px0; uint32_t npx; uint32_t slot; };

static int enqueue_passes(FrameJob& J, uint64_t px0, uint64_t npx, uint32_t pass_px, std::vector<PassRec>* recs) {
    rt_scene* s = J.s;
    const uint32_t spp = J.base.spp;
    const bool meshes = s->n_models > 0;
    for (uint64_t done = 0; done < npx; done += pass_px) {
        // A synchronous frame reads every pass' counters back, so its passes need a slot each. An asynchronous frame reads
        // none (k_resolve raises the host flag): its passes take the slots in turn, and a slot that comes round again is
        // cleared on the frame's stream first - everything that used it is behind that point: a pass ends with its shade
        // streams joined into this stream, its folds, its resolve and the copy of its hints.
        if (J.n_pass_used >= RT_MAX_PASSES && recs) { rt_set_error("frame needs more than %d passes", RT_MAX_PASSES); return RT_ERR_LIMIT; }
        const uint32_t cnt = uint32_t(std::min<uint64_t>(pass_px, npx - done));
        const uint32_t slot = J.n_pass_used % RT_MAX_PASSES;
        PassCounters* pc = s->d_counters + slot;
        if (J.n_pass_used >= J.slots_cleared) HIP_TRY(hipMemsetAsync(pc, 0, sizeof(PassCounters), J.st));
        ++J.n_pass_used;
        LevelArgs A = J.base;
        A.n0 = cnt * spp;
        A.sample0 = (px0 + done) * spp;
        A.fast32 = (A.sample0 + A.n0 <= 0xFFFFFFFFull && !s->knobs.index64) ? 1u : 0u;      // (option "index64": tests of the 64-bit path)
        A.sc = pc->sc;
        A.cancelled = pc->cancel;
        A.host_overflow = s->h_overflow;
        // small passes: smaller level-0 chunks, so that all shards (sub-pools) get their share of the entries
        while (A.chunk_log2 > 0 && (A.n0 >> (6u + A.chunk_log2)) < RT_SHARDS) --A.chunk_log2;
        // The static pool cannot grow, so there a pass of fewer 64-sample blocks than shards, which uses only that many
        // sub-pools, shares the whole pool among them: cutting a pass further (after an overflow) keeps giving more room per
        // sample down to the 64-pixel pass. (The ordinary path grows its pool instead: render_frame's retry loop.)
        if (const uint32_t n_blocks = (A.n0 + 63u) >> 6; J.noalloc && A.chunk_log2 == 0 && n_blocks < RT_SHARDS)
            A.sub_cap = uint32_t(uint64_t(A.sub_cap) * RT_SHARDS / n_blocks);
        A.overflow = &pc->overflow;
        A.shadow_count = pc->shadow;
        A.work = &pc->work[0][0][0];
        const int levels = s->has_bounce ? A.max_depth : 0;
        const bool overlap = !J.timed;
        hipStream_t sa = J.st;
        uint32_t grid0 = (A.n0 + RT_BLOCK - 1) / RT_BLOCK;
        if (s->knobs.grid0) grid0 = std::min<uint32_t>(grid0, uint32_t(s->knobs.grid0));      // experiments
        const uint32_t grid0q = std::min<uint32_t>((A.n0 + 15) / 16, 4u * RT_LEVEL_GRID);
        // Expensive groups first (k_order_groups): a frame that is ONE pass of a few thousand to RT_ORDER_MAX_GROUPS 64-sample
        // groups - where a level-0 launch is mostly the drain of its longest waves - hands the groups out in descending order
        // of what they cost in the scene's previous frame of the same size. Not for the static pool (allocates) or a graph.
        const uint32_t n_grp0 = (A.n0 + 63u) / 64u;
        // A scene with meshes none of whose materials bounces: level 0 is the whole frame and, as packets, suspends nothing -
        // closest hits and light loops in ONE launch (k_trace_wide<0, false, PACKET>), in pixel order (measured, shadows-only bunny
        // at 1080p, one frame at a time: two launches in pixel order 0.312 ms, expensive groups first 0.275, one launch 0.266; in
        // batches 0.1313 -> 0.1295; the one launch with the expensive groups first - k_trace_wide<.., ORDERED>, a group's cost the whole
        // of its wave: see DESIGN.md)
        const bool fuse_l0 = meshes && !s->has_bounce && s->knobs.fuse && s->d.packet && !s->d.generic_pow && !s->count_work;
        bool order_on = s->knobs.order0 && px0 == 0 && done == 0 && npx <= pass_px && n_grp0 >= 2048u && n_grp0 <= RT_ORDER_MAX_GROUPS &&
                        s->d.packet && !s->d.generic_pow && !s->count_work &&      // (the kernels that log and follow an order: k_extend_wide / k_shade_wide <.., ORDERED>)
                        s->knobs.grid0 == 0 && !J.noalloc && !s->knobs.use_graph && ensure_order_buf(s, n_grp0, sa);
        const uint32_t* order_e = nullptr;
        const uint32_t* order_s = nullptr;
        rt_scene::OrderSet& O = s->oset[s->order_turn & 1u];      // this frame's set
        if (order_on && O.pending) {
            // the previous frame's k_order_groups is not part of what that frame waits for: when it has not finished yet (a frame
            // whose level-0 light loops are the last thing it does), this frame neither waits for it nor touches its arrays
            if (hipEventQuery(O.ev) == hipSuccess) O.pending = false;
            else { (void)hipGetLastError(); order_on = false; }
        }
        if (order_on && O.n == n_grp0) { order_e = O.order[0]; order_s = O.order[1]; }
        int rc;
        bool used_shade[3] = {false, false, false};
        // a shade stream waits for an event once per RECORDING of it: slots are re-recorded within a pass (ev_sync[2*lv]
        // before shade_small and again after that level's per-lane extend), so the elision compares (slot, generation)
        int last_wait[3] = {-1, -1, -1};
        int small_runs = 0;      // runs of small levels shaded so far in this pass
        int ev_gen[2 * RT_MAX_LEVELS + 5] = {};
        auto record = [&](int slot) -> hipError_t { ++ev_gen[slot]; return hipEventRecord(s->ev_sync[slot], sa); };
        // (a "lean" schedule for small passes - one shade stream, one shade launch for all small levels, deep levels folded
        // by one workgroup, 30 API calls instead of 50 - was measured SLOWER: 0.27 vs 0.24 ms for an eighth of the frame)
        uint4* wave_log = nullptr;
        int log_level = -1;
        bool log_shade = false;
#ifdef RT_EXPERIMENTS
        // diagnostics: RT_WAVE_LOG=<level>:<file> dumps per-workgroup timing of that level's quad extend kernel (COUNT builds)
        log_shade = s->knobs.wave_log_shade;      // the level's per-lane shade kernel instead of its extend kernel
        if (s->knobs.wave_log_level >= 0) {
            log_level = s->knobs.wave_log_level;
            hipMalloc(reinterpret_cast<void**>(&wave_log), sizeof(uint4) * 4u * RT_LEVEL_GRID);
            if (wave_log) hipMemsetAsync(wave_log, 0, sizeof(uint4) * 4u * RT_LEVEL_GRID, sa);
        }
#endif
        // shade launch `form` over levels lo..hi on the level's shade stream, after event `after` of stream A
        auto shade = [&](int form, int lo, int hi, int after, bool small_run = false) -> int {
            // (runs of small levels take the third stream and the second in turn: the run that ends a pass - the tail's hits -
            // would otherwise queue behind the resumed shadow rays of the run before it, 45 us of a one-frame call)
            // (Keeping the later runs off the stream of a run that was shaded early for its size - option run_hits: its light loops
            // hold that stream until the pass is nearly over, and the pass's last run waits 0.4 ms behind them in a call of 20
            // frames - measured slower: two calls of 32 frames in flight 0.210 -> 0.214 ms, 800x600 0.0622 -> 0.0641.)
            const int which = ((form == RT_FORM_QUAD_FRESH || small_run) ? 2 - (small_runs & 1) : (lo & 1)) % s->n_shade_streams;
            // (the run that ends a pass, when it is one small launch, on stream A behind the launch that made its hits - no
            // event to record and wait for on either side of it - measured no different: 0.503-0.510 against 0.504-0.507 ms)
            hipStream_t sb = overlap ? s->shade_stream[which] : sa;
            if (overlap) {
                const int stamp = after * 65536 + ev_gen[after];
                if (last_wait[which] != stamp) HIP_TRY(hipStreamWaitEvent(sb, s->ev_sync[after], 0));
                last_wait[which] = stamp;
                used_shade[which] = true;
            }
            // level 0 lies in sample order, the deeper levels in the sub-pools: a range that starts at level 0 is two launches
            for (int part = (lo == 0 && hi > 0) ? 0 : 1; part < 2; ++part) {
                LevelArgs B = A;
                B.level = part == 0 ? 0 : (lo == 0 && hi > 0 ? 1 : lo);
                B.level_hi = part == 0 ? 0 : hi;
                if (order_on && B.level_hi == 0 && form == RT_FORM_WIDE) { B.order0 = order_s; B.cost0 = O.cost[1]; }
                B.wave_log = (log_shade && B.level == log_level) ? wave_log : nullptr;
                const bool wide = form == RT_FORM_WIDE || form == RT_FORM_WIDE_SERVE;
                const uint32_t wide_grid = wide ? wide_grid_for(s, slot, B.level, B.level_hi) : RT_LEVEL_GRID;
                const uint32_t g = wide ? (B.level_hi == 0 ? (B.order0 ? RT_ORDER_SLOTS(grid0) : grid0) : wide_grid)
                                   : (B.level_hi == 0 ? grid0q : (form == RT_FORM_QUAD_FRESH ? small_grid(s, slot, B.level, B.level_hi)
                                                                  : small_run ? resume_grid(s, slot, B.level) : 4u * RT_LEVEL_GRID));
                // packets over a bounce level: the hits sorted by position first (k_sort_keys, radix sort), from knob sort_shade
                // rays on and where the previous frame's count of this pass says how many slots to sort; a level that grew past
                // the estimate is walked in pool order behind the sorted part. Never for the static pool (allocates).
                uint32_t sort_slots = 0;
                if (form == RT_FORM_WIDE && B.level >= 1 && s->d.packet >= 2 && s->knobs.sort_shade > 0 && s->sort_box_ok && !J.noalloc &&
                    !s->knobs.use_graph) {
                    uint64_t n = 0;
                    for (int l = B.level; l <= B.level_hi; ++l) n += level_hint(s, slot, l);
                    if (n >= uint64_t(s->knobs.sort_shade) && n < (1ull << 30)) {
                        sort_slots = uint32_t(((n + n / 8u) / 64u + 2u * RT_SHARDS) * 64u);
                        if (!ensure_sort_buf(s, which, sort_slots)) sort_slots = 0;
                    }
                }
                const int rc2 = timed_launch(J, sb, form == RT_FORM_WIDE ? RT_K_SHADE_WIDE : form == RT_FORM_QUAD_PARKED ? RT_K_SHADE_PARKED : RT_K_SHADE_QUAD,
                                             [&] {
                                                 if (sort_slots) {
                                                     rt_scene::SortBuf& sbuf = s->sort_buf[which];
                                                     B.perm = launch_sort_hits(s->d, B, s->sort_box, uint32_t(s->knobs.sort_bits), sbuf.keys, sbuf.vals, sbuf.temp, sbuf.temp_bytes, sort_slots, sb);
                                                     B.perm_n = B.perm ? sort_slots : 0u;
                                                 }
                                                 launch_shade(form, s->count_work, s->d, B, g, sb);
                                             });
                if (rc2 != RT_OK) return rc2;
            }
            return RT_OK;
        };
        // The hits of a run of small levels lo..hi (extended as whole rays by the quad kernel): the light loops' analytic
        // objects and the Phong terms one hit per LANE (a quad kernel computes them four times over), the shadow rays that
        // reach a mesh queued and resumed by quads: 39 M -> 23 M VALU instructions per 1080p frame.
        // (Where the pass-size rule switches, measured again at round 3's kernels, bunny_1080p, two calls in flight, quads / quad
        // service: 3 frames per call - 6.2 M samples - 0.287 / 0.307 ms per frame, 4 frames 0.265 / 0.268, 6 frames 0.253 / 0.245; a
        // rank's quarter of ten tiled frames, 5.2 M, 0.073 / 0.077; its half, 10.4 M, the same; 12 frames of 800x600, 5.8 M, 0.0762 /
        // 0.0776: from 10 M samples on - it was 4 M.)
        // Bounce levels of a pass that fills the chip by itself (10 M samples and more: a batch of frames, a 4K frame) are
        // bound by instruction issue, not by the length of their waves: there the per-lane kernels with quad service do
        // extend and shade (fewest instructions: bunny_1080p in batches 0.326 -> 0.302 ms, c1 0.089 -> 0.080, 10 M-triangle
        // field 5.40 -> 5.22); a smaller pass keeps one ray per quad, whose waves are a quarter as long (one 1080p frame at
        // a time: 0.73 ms against 0.91 with quad service). RT_SERVE=0/1 overrides.
        const int serve_env = s->knobs.serve;
            // (Scenes with packets at every level, 16+ spp: their small bounce levels only from 32 M-sample passes on - 4K glass
        // scene at 2^26 32.0 -> 31.0 ms, at 2^25 34.1 -> 33.1; 10 M-triangle field at 8K: 2^23 632 -> 673 ms, 2^26 the same.)
        const bool serve = s->d.packet >= 1 && (serve_env >= 0 ? serve_env != 0 : (A.n0 >= (s->d.packet == 1 ? (10u << 20) : (32u << 20))));
        // Waves per workgroup of the pass' per-lane launches: four where the frame comes alone and has bounce levels (a pass of at
        // most 4 M samples), one everywhere else. Measured (RT_WIDE_BLOCK 64 / 256 for every launch): bunny_1080p one frame at a
        // time 0.511 -> 0.484 ms, rt_render into host memory 0.656 -> 0.632, in batches 0.210 -> 0.224; 800x600 0.340 -> 0.336;
        // shadows-only bunny (no bounce levels) 0.206 -> 0.213.
        // (A single frame of the 10 M-triangle field at 1080p 4 spp, 8.3 M samples: 4.63 -> 5.03 ms with four - hence up to 4 M samples;
        // a rank's eighth of ten tiled frames, 2.6 M: 0.0503 -> 0.0484 ms per frame.)
        // (levels > 0: what four waves per workgroup buy is the bounce levels' share of the chip beside level 0's light loops - a
        // frame without bounce levels, or rendered at depth 0, is 2-4 % slower with them)
        A.wide_waves = (s->knobs.wide_waves_small > 0 && levels > 0 && meshes && !serve && A.n0 <= (4u << 20)) ? uint32_t(s->knobs.wide_waves_small) : 0u;
        const bool split_small = s->knobs.split_small != 0;
        auto shade_small_run = [&](int lo, int hi, int after) -> int {
            // (not where the per-lane kernels traverse meshes as packets at every level: the rays of a small bounce level
            // have nothing in common - 10 M-triangle field at 8K 64 spp 717 -> 970 ms)
            if (serve && level_hint(s, slot, lo) < 0xFFFFFFFFull) return shade(RT_FORM_WIDE_SERVE, lo, hi, after, true);
            if (!split_small || s->d.packet >= 2) return shade(RT_FORM_QUAD_FRESH, lo, hi, after);
            // a run of a few hundred hits: two launches would cost more than the instructions one quad launch repeats
            {
                uint64_t n = 0;
                for (int l = lo; l <= hi; ++l) n += level_hint(s, slot, l);
                if (n <= uint64_t(s->knobs.tiny_run)) return shade(RT_FORM_QUAD_FRESH, lo, hi, after);
            }
            const int rc2 = shade(RT_FORM_WIDE, lo, hi, after, true);
            if (rc2 != RT_OK) return rc2;
            return shade(RT_FORM_QUAD_RESUME, lo, hi, after, true);
        };
        auto shade_small = [&](int lo, int hi, int after) -> int { const int r = shade_small_run(lo, hi, after); ++small_runs; return r; };
        int small_lo = -1;      // first level of the run of small levels not shaded yet
        // The tail (option "tail_rays"): from the first level on that, together with everything deeper, is expected to hold
        // at most that many rays, ONE launch traces every such ray and its whole subtree depth first, one ray per quad
        // (k_tail), instead of one dependent closest-hit launch per level; the light loops of all those entries are part of
        // the last run of small levels, and k_fold_tree folds the subtrees. Same bytes and ray counts.
        int lv_tail = levels + 1;
        {
            const uint64_t thr = uint64_t(s->knobs.tail);      // rays; 0 = never
            uint64_t below = 0;
            if (meshes && thr && !wave_log)
                for (int l = levels; l >= 1; --l) {
                    below += level_hint(s, slot, l);
                    if (below > thr) break;
                    lv_tail = l;
                }
        }
        for (int lv = 0; lv <= levels; ++lv) {
            if (lv == lv_tail) {
                LevelArgs T = A;
                T.level = lv; T.level_hi = lv;
                if ((rc = timed_launch(J, sa, RT_K_EXTEND_QUAD, [&] { launch_tail(s->count_work, s->d, T, pc->tail_levels, small_grid(s, slot, lv, lv), sa); })) != RT_OK) return rc;
                if (small_lo < 0) small_lo = lv;      // its hits are shaded with the run of small levels that ends the pass
                break;
            }
            LevelArgs B = A;
            B.level = lv; B.level_hi = lv;
            if (order_on && lv == 0) { B.order0 = order_e; B.cost0 = O.cost[0]; }
            const int src = lv == 0 ? 0 : 1;
            const bool large = !meshes || lv == 0 || expect_large(s, slot, lv, lv);
            B.wave_log = (lv == log_level && !log_shade) ? wave_log : nullptr;
            if (large) {
                if (small_lo >= 0) {     // shade the small levels before this one
                    if (overlap) HIP_TRY(record(2 * lv));
                    if ((rc = shade_small(small_lo, lv - 1, 2 * lv)) != RT_OK) return rc;
                    small_lo = -1;
                }
                // a scene without meshes suspends nothing: the level's light loops in the same launch (k_trace_wide)
                // (... and so does level 0 as packets, where it is the whole frame: no material bounces)
                const bool fused = (!meshes || fuse_l0) && s->knobs.fuse && !s->count_work && !wave_log;
                if ((rc = timed_launch(J, sa, RT_K_EXTEND_WIDE, [&] { launch_extend(src, fused ? RT_FORM_WIDE_FUSED : RT_FORM_WIDE, s->count_work, s->d, B, lv == 0 ? (B.order0 ? RT_ORDER_SLOTS(grid0) : grid0) : wide_grid_for(s, slot, lv, lv), sa); })) != RT_OK) return rc;
                if (fused) continue;
                if (overlap) HIP_TRY(record(2 * lv));
                // level 0 of a packet scene: the per-lane kernels traverse the meshes themselves, nothing is suspended
                const bool quads = meshes && !((lv == 0 && s->d.packet) || s->d.packet >= 2);
                if (quads) {
                    if ((rc = timed_launch(J, sa, RT_K_EXTEND_QUAD, [&] { launch_extend(src, RT_FORM_QUAD_RESUME, s->count_work, s->d, B, lv == 0 ? grid0q : 4u * RT_LEVEL_GRID, sa); })) != RT_OK) return rc;
                    if (overlap) HIP_TRY(record(2 * lv + 1));
                }
                // (Level 0's light loops fill the chip and run next to the closest-hit launches of levels 1, 2, ...; issuing
                // them behind the extend launch of level 1 / 2 / 3 instead, so that the chain's first large levels have the
                // chip to themselves, was measured slower: one frame at a time 0.686 -> 0.706 / 0.72 / 0.745 ms.)
                const int ev_w = 2 * lv;
                const bool shade_by_service = !quads && lv >= 1 && s->knobs.shade_serve_lv > 0 && lv >= s->knobs.shade_serve_lv;
                if ((rc = shade(shade_by_service ? RT_FORM_WIDE_SERVE : RT_FORM_WIDE, lv, lv, ev_w)) != RT_OK) return rc;
                if (quads) {
                    if ((rc = shade(RT_FORM_QUAD_RESUME, lv, lv, ev_w)) != RT_OK) return rc;
                    if ((rc = shade(RT_FORM_QUAD_PARKED, lv, lv, 2 * lv + 1)) != RT_OK) return rc;
                }
            } else {
                // (the same split for the extend side - per-lane kernel, queue, resuming quads, launches sized by the hints -
                // was measured slower at every level size: bunny_1080p 0.325 -> 0.345 ms, 10 M-triangle field 5.40 -> 5.67..5.86:
                // most bounce rays reach the mesh anyway, and their state makes a round trip through memory)
                const uint64_t lh = level_hint(s, slot, lv);
                if (serve && lh < 0xFFFFFFFFull) {
                    if ((rc = timed_launch(J, sa, RT_K_EXTEND_QUAD, [&] { launch_extend(src, RT_FORM_WIDE_SERVE, s->count_work, s->d, B, wide_grid_for(s, slot, lv, lv), sa); })) != RT_OK) return rc;
                } else
                if ((rc = timed_launch(J, sa, RT_K_EXTEND_QUAD, [&] { launch_extend(src, RT_FORM_QUAD_FRESH, s->count_work, s->d, B, small_grid(s, slot, lv, lv), sa); })) != RT_OK) return rc;
                if (small_lo < 0) small_lo = lv;
                uint64_t run_hits = 0;
                for (int l = small_lo; l <= lv; ++l) run_hits += level_hint(s, slot, l);
                const bool big_run = s->knobs.run_hits > 0 && run_hits >= uint64_t(s->knobs.run_hits) && run_hits < 0xFFFFFFFFull;
                if ((lv - small_lo + 1 >= 3 || big_run) && lv < levels) {     // shade a run of three small levels (or one that holds a lot of hits) while the deeper ones are traced
                    if (overlap) HIP_TRY(record(2 * lv + 1));
                    if ((rc = shade_small(small_lo, lv, 2 * lv + 1)) != RT_OK) return rc;
                    small_lo = -1;
                }
            }
        }
#ifdef RT_EXPERIMENTS
        if (wave_log) {
            std::vector<uint4> h(4u * RT_LEVEL_GRID);
            hipStreamSynchronize(sa);
            hipMemcpy(h.data(), wave_log, sizeof(uint4) * h.size(), hipMemcpyDeviceToHost);
            if (FILE* f = std::fopen(s->knobs.wave_log_path.c_str(), "wb")) { std::fwrite(h.data(), sizeof(uint4), h.size(), f); std::fclose(f); }
            hipFree(wave_log);
        }
#endif
        if (small_lo >= 0) {
            const int ev = 2 * RT_MAX_LEVELS;
            if (overlap) HIP_TRY(record(ev));
            if ((rc = shade_small(small_lo, levels, ev)) != RT_OK) return rc;
        }
        for (int k = 0; k < 3; ++k)
            if (used_shade[k]) {
                HIP_TRY(hipEventRecord(s->ev_sync[2 * RT_MAX_LEVELS + 1 + k], s->shade_stream[k]));
                HIP_TRY(hipStreamWaitEvent(sa, s->ev_sync[2 * RT_MAX_LEVELS + 1 + k], 0));
            }
        if (order_on) {
            // behind level 0's light loops on their stream, NOT part of the join above: the frame does not wait for it. (At most
            // 65,536 groups: ~10 us, over long before the next frame's light loops, which queue behind it, can start. On a
            // stream of its own - a fifth per scene - one frame at a time got slower, 0.555 -> 0.567 ms: the streams share
            // four hardware queues.)
            hipStream_t so = overlap ? s->shade_stream[0] : sa;
            if (fuse_l0 && overlap) {      // (one launch did the level on stream A: the light-loop stream has nothing of this frame to follow)
                HIP_TRY(hipEventRecord(s->ev_sync[2 * RT_MAX_LEVELS + 4], sa));
                HIP_TRY(hipStreamWaitEvent(so, s->ev_sync[2 * RT_MAX_LEVELS + 4], 0));
            }
            launch_order_groups(O.cost[0], O.cost[1], O.order[0], O.order[1],
                                reinterpret_cast<uint8_t*>(O.order[0] + RT_ORDER_SLOTS(s->order_cap)),
                                reinterpret_cast<uint8_t*>(O.order[1] + RT_ORDER_SLOTS(s->order_cap)), n_grp0, so);
            HIP_TRY(hipEventRecord(O.ev, so));
            O.pending = true;
            O.n = n_grp0;
            ++s->order_turn;
        }
        // (folding the deep levels, which hold few nodes, by ONE workgroup in one launch was measured again with the sharded
        // pool: 85 us against 9 launches x 5 us, 0.706 vs 0.686 ms/frame)
        // (Everything below level 1 or 2 folded as subtrees by ONE k_fold_tree launch, one thread per entry of that level, instead
        // of a k_fold launch per level - the 4K glass scene's first pass ends in 1.9 ms of seven dependent fold launches, 9 M
        // entries each - measured slower: 25.0 -> 26.8 / 27.2 ms per frame; bunny_1080p the same.)
        const int fold_hi = std::min(levels, lv_tail) - 1;      // (the tail's levels are folded as subtrees)
        if (lv_tail <= levels) {
            A.level = lv_tail; A.level_hi = lv_tail;
            launch_fold_tree(s->d, A, uint32_t(std::min<uint64_t>(1024u, level_hint(s, slot, lv_tail) / 48u + 2u * RT_SHARDS)), sa);
        }
        // (the deep levels, a few hundred nodes each, folded by ONE 16-wave workgroup with a workgroup barrier per level:
        // measured no different, 0.685 against 0.676 ms one frame at a time - a level's fold is ~5 us of dependent loads
        // - counters, node, children - whether it is a launch of its own or not; all levels that way: 85 us against 9 x 5)
        for (int lv = fold_hi; lv >= 1; --lv) {
            A.level = lv; A.level_hi = lv;
            launch_fold(s->d, A, 1024, sa);
        }
        A.level = 0; A.level_hi = 0;
        launch_resolve(s->d, A, J.d_out, sa);
        HIP_TRY(hipGetLastError());
        // level sizes of this pass -> scheduling hint of the next one (no sync; see h_hint)
        HIP_TRY(hipMemcpyAsync(s->h_hint + size_t(slot) * RT_SHARDS, pc->sc, sizeof(ShardCounters) * RT_SHARDS, hipMemcpyDeviceToHost, sa));
        HIP_TRY(hipMemcpyAsync(s->h_tail_hint + size_t(slot) * (RT_MAX_LEVELS + 4), pc->tail_levels, sizeof(uint32_t) * (RT_MAX_LEVELS + 4), hipMemcpyDeviceToHost, sa));
        if (recs) recs->push_back({px0 + done, cnt, slot});
        if (J.progress && recs) {       // event number = index of the pass in recs
            while (s->ev_pass.size() < recs->size()) {
                hipEvent_t e;
                HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
                s->ev_pass.push_back(e);
            }
            HIP_TRY(hipEventRecord(s->ev_pass[recs->size() - 1], sa));
        }
    }
    return RT_OK;
}

// An asynchronous frame overflowed: the next ones get 4x the room per sample, by smaller passes while those can shrink,
// then by a larger pool.
static void learn_from_overflow(rt_scene* s, uint32_t full_px) {
    if (s->last_noalloc) {
        // the frame that overflowed used the static pool: only ITS passes shrink (of RT_NOALLOC_PASS_SAMPLES, down to 64
        // pixels); the scene's ordinary passes and its hipMalloc'ed pool have nothing to learn from it
        const uint32_t px = std::max<uint32_t>(64u, RT_NOALLOC_PASS_SAMPLES / std::max<uint32_t>(1u, s->last_spp));
        if ((px >> s->noalloc_shrink_log2) > 64u) s->noalloc_shrink_log2 = std::min(8u, s->noalloc_shrink_log2 + 2u);
        return;
    }
    if ((full_px >> s->shrink_log2) > 64u && s->shrink_log2 < 16) s->shrink_log2 = std::min(16u, s->shrink_log2 + 2u);
    else s->ws_min_n0 = uint32_t(std::min<uint64_t>(RT_POOL_MAX_N0, std::max<uint64_t>(uint64_t(s->ws_n0) * 8u, 65536u)));
}

// Renders n_frames frames that differ only in their cameras (cam[0..n_frames)) as ONE sample range: frame b's pixels
// follow frame b-1's, frame b's output starts b * out_stride bytes behind d_out.
// noalloc: the workspace is the static array of rt_kernels.hip (a pass of RT_NOALLOC_PASS_SAMPLES), nothing is
// allocated, grown or freed; asynchronous only.
static int render_frame(rt_scene* s, const rt_camera* cam, uint32_t n_frames, const rt_params* prm, const PixelMap& pm,
                        uint8_t* d_out, uint64_t out_stride, hipStream_t st, rt_stats* stats, bool sync, bool noalloc = false,
                        rt_progress_fn progress = nullptr, void* progress_user = nullptr) {
    HIP_TRY(hipSetDevice(s->device));
    const uint32_t spp = 1 + prm->n_aa;
    const uint64_t frame_px = uint64_t(pm.n_owned) * pm.tile_w * pm.tile_h;
    const uint64_t total_px = frame_px * n_frames;
    if (stats) std::memset(stats, 0, sizeof *stats);
    if (total_px == 0) return RT_OK;
    if (n_frames > 1 && frame_px * spp > 0xFFFFFFC0ull) { rt_set_error("frames of a batch are limited to 2^32 samples each"); return RT_ERR_LIMIT; }
    // The workspace is sized for the full pass; a scene whose ray trees outgrow the pool (learned from overflows, below
    // and in rt_scene_status) runs smaller passes in the same workspace: 2^k times the room per sample.
    uint32_t full_px = std::max<uint32_t>(64u, RT_PASS_SAMPLES / spp);
    if (*s->h_overflow && s->last_overflow == -1) { s->last_overflow = 1; learn_from_overflow(s, std::max<uint32_t>(64u, RT_PASS_SAMPLES / std::max<uint32_t>(1u, s->last_spp))); }
    uint32_t pass_px = uint32_t(std::min<uint64_t>(total_px, std::max<uint32_t>(64u, full_px >> s->shrink_log2)));
    pass_px = (pass_px + 63u) & ~63u;
    int rc = RT_OK;
    if (noalloc) {
        // whole pixels, a multiple of 64 of them, within the static pool (rounded DOWN; 64 pixels of <= 256 samples fit)
        const uint32_t fit_px = std::max<uint32_t>(64u, ((RT_NOALLOC_PASS_SAMPLES / spp) >> s->noalloc_shrink_log2) & ~63u);
        pass_px = uint32_t(std::min<uint64_t>((total_px + 63u) & ~63ull, fit_px));
        if (uint64_t(pass_px) * spp > RT_NOALLOC_PASS_SAMPLES) { rt_set_error("64 pixels of %u samples do not fit the static workspace", spp); return RT_ERR_LIMIT; }
    } else {
        rc = ensure_workspace(s, std::max(((uint32_t(std::min<uint64_t>(total_px, full_px)) + 63u) & ~63u) * spp, s->ws_min_n0));
        // no room for passes that large (other scenes, other users of the device): smaller passes, down to the 2^23 default
        while (rc == RT_ERR_DEVICE && s->pass_samples > (1u << 23) && uint64_t(pass_px) * spp > (1u << 23)) {
            s->pass_samples >>= 1;
            full_px = std::max<uint32_t>(64u, RT_PASS_SAMPLES / spp);
            pass_px = uint32_t(std::min<uint64_t>(total_px, std::max<uint32_t>(64u, full_px >> s->shrink_log2)));
            pass_px = (pass_px + 63u) & ~63u;
            rc = ensure_workspace(s, std::max(((uint32_t(std::min<uint64_t>(total_px, full_px)) + 63u) & ~63u) * spp, s->ws_min_n0));
        }
        if (rc != RT_OK) return rc;
    }

    // Packets below level 0 too (the large levels' per-lane kernels) when a 64-entry block holds the children of few
    // pixels, i.e. at 16+ samples per pixel: measured 1426 -> 844 ms (level 0) -> 791 ms (all levels) per frame on the 10 M-
    // triangle scene at 64 spp and 78 -> 69 -> 55..67 ms on the 4K 16 spp glass scene, but 6.3 -> 10.9 ms at 4 spp.
    if (s->packet_auto && s->n_models) s->d.packet = spp >= 16 ? 2u : 1u;
    // (the hints this frame schedules by were left by the previous frame, whose depth is s->hint_depth; the frames after
    // this one read this frame's: the new depth takes effect when this call returns)
    struct DepthGuard { rt_scene* s; int d; ~DepthGuard() { s->hint_depth = d; } } depth_guard{s, s->has_bounce ? prm->max_depth : 0};
    FrameJob J{};
    J.s = s; J.st = st; J.d_out = d_out; J.timed = stats != nullptr;
    J.progress = sync ? progress : nullptr; J.progress_user = progress_user;
    J.noalloc = noalloc;
    LevelArgs& B = J.base;
    std::memset(&B, 0, sizeof B);
    B.max_depth = prm->max_depth;
    B.spp = spp;
    B.sub_cap = s->sub_cap;
    B.rays = s->rays; B.node_a = s->node_a; B.node_b = s->node_b; B.hit_n = s->hit_n; B.hit_p = s->hit_p; B.queue_e = s->queue_e; B.queue_s = s->queue_s;
    if (noalloc) {      // the same arrays, carved out of this device's static workspace
        unsigned char* w = static_cast<unsigned char*>(s->noalloc_ws);
        const size_t cap = RT_NOALLOC_CAP;
        B.node_a = reinterpret_cast<float4*>(w); w += cap * sizeof(float4);
        B.node_b = reinterpret_cast<uint4*>(w);  w += cap * sizeof(uint4);
        B.hit_n = reinterpret_cast<float4*>(w);  w += cap * sizeof(float4);
        B.hit_p = reinterpret_cast<float4*>(w);  w += cap * sizeof(float4);
        B.queue_e = reinterpret_cast<uint2*>(w); w += cap * sizeof(uint2);
        B.queue_s = reinterpret_cast<uint2*>(w); w += cap * sizeof(uint2);
        B.rays = reinterpret_cast<float4*>(w);
        B.sub_cap = s->has_bounce ? (RT_NOALLOC_CAP - RT_NOALLOC_PASS_SAMPLES) / RT_SHARDS : 0u;
    }
    B.chunk_log2 = uint32_t(s->knobs.chunk_log2);
#ifdef RT_EXPERIMENTS
    B.debug = s->knobs.debug; s->d.debug = B.debug;      // timing experiments only (RT_DEBUG_SKIP)
#endif
    B.aa = s->d_aa;
    B.fp = s->d_fp;
    B.pm = pm;
    rt_fill_recips(B.pm);
    B.frame_samples = n_frames > 1 ? uint32_t(frame_px * spp) : 0u;
    B.frame_px = B.frame_samples / spp;
    B.z_spp = rt_recip32(spp); B.z_frame_samples = rt_recip32(B.frame_samples); B.z_frame_px = rt_recip32(B.frame_px);
    B.out_stride = out_stride;
    for (int lv = 0; lv <= prm->max_depth; ++lv)
        if (expect_large(s, 0, lv, lv)) J.large_mask |= 1u << lv;      // (key of the opt-in graph cache only)

    const uint64_t n_pass = (total_px + pass_px - 1) / pass_px;
    // (an asynchronous frame's passes take the counter slots in turn: enqueue_passes)
    if (n_pass > RT_MAX_PASSES && sync) { rt_set_error("frame needs %llu passes (max %d)", (unsigned long long)n_pass, RT_MAX_PASSES); return RT_ERR_LIMIT; }

    // per-frame inputs: the anti-aliasing pattern (only when it changed) and camera + background
    if (prm->n_aa && (s->aa_cache.size() != size_t(2 * prm->n_aa) ||
                      std::memcmp(s->aa_cache.data(), prm->aa_xy, sizeof(float) * 2 * prm->n_aa) != 0)) {
        s->aa_cache.assign(prm->aa_xy, prm->aa_xy + 2 * prm->n_aa);
        // (as kernel arguments, not a copy: a copy from the caller's pageable memory is staged by the runtime - it may
        // block and allocate - and a staging buffer of the scene's own could be overwritten while a frame in flight reads it)
        launch_set_aa(prm->aa_xy, prm->n_aa, s->d_aa, st);
    }
    FrameParamsBatch fp;
    std::memset(&fp, 0, sizeof fp);
    for (uint32_t b = 0; b < n_frames; ++b) {
        std::memcpy(fp.f[b].background, prm->background, sizeof fp.f[b].background);
        fp.f[b].cam = to_dcamera(cam + b);
    }
    const size_t counter_bytes = sizeof(PassCounters) * std::min<uint64_t>(RT_MAX_PASSES, n_pass * 2 + 16);
    J.slots_cleared = uint32_t(counter_bytes / sizeof(PassCounters));
    static_assert(sizeof(PassCounters) % 16 == 0, "counters are cleared 16 bytes at a time");
    launch_set_frame_params(fp, n_frames, s->d_fp, s->d_counters, counter_bytes, st);      // also clears the frame's pass counters
    std::vector<PassRec> recs;
    if (!sync) {
        // Asynchronous call: overflow (if any) is left in the counters. The launches of the frame depend on nothing
        // that changes with the camera, so they CAN be captured once into a hipGraph and replayed (~50 API calls
        // become two). Measured on ROCm 7.2 / MI355X the replay is slower than the direct launches (bunny_1080p,
        // two frames in flight: 1.25 vs 0.82 ms/frame; an eighth of the tiles: 0.46 vs 0.24 ms): the captured
        // multi-stream schedule loses its overlap. So the graph path is opt-in (RT_USE_GRAPH=1) and off by default.
        if (s->last_overflow != 1) s->last_overflow = -1;      // unknown until rt_scene_status (1 stays until it is read)
        s->last_stream = st;
        s->last_spp = spp;
        s->last_noalloc = noalloc;
        const bool no_graph = !s->knobs.use_graph;
        rt_scene::GraphKey key;
        std::memset(&key, 0, sizeof key);
        key.base = B; key.d_out = d_out; key.total_px = total_px; key.pass_px = pass_px; key.large_mask = J.large_mask;
        key.count = s->count_work; key.debug = s->d.debug;
        if (s->graph_ok && !no_graph && !s->count_work && !noalloc) {
            for (size_t g = 0; g < s->graphs.size(); ++g)
                if (std::memcmp(&key, &s->graphs[g].key, sizeof key) == 0) {
                    rt_scene::CachedGraph cg = s->graphs[g];
                    s->graphs.erase(s->graphs.begin() + long(g));
                    cg.last_stream = st;
                    s->graphs.push_back(cg);
                    HIP_TRY(hipGraphLaunch(cg.exec, st));
                    return RT_OK;
                }
            hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
            if (hipStreamIsCapturing(st, &cs) == hipSuccess && cs == hipStreamCaptureStatusNone &&
                hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) == hipSuccess) {
                int crc = RT_OK;
                if (hipMemsetAsync(s->d_counters, 0, counter_bytes, st) != hipSuccess) crc = RT_ERR_DEVICE;
                if (crc == RT_OK) crc = enqueue_passes(J, 0, total_px, pass_px, nullptr);
                hipGraph_t graph = nullptr;
                const hipError_t ee = hipStreamEndCapture(st, &graph);
                hipGraphExec_t exec = nullptr;
                if (crc == RT_OK && ee == hipSuccess && graph && hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) == hipSuccess) {
                    hipGraphDestroy(graph);
                    if (s->graphs.size() >= RT_MAX_GRAPHS) {      // evict the least recently used one, once it cannot be running
                        HIP_TRY(hipStreamSynchronize(s->graphs.front().last_stream));
                        hipGraphExecDestroy(s->graphs.front().exec);
                        s->graphs.erase(s->graphs.begin());
                    }
                    s->graphs.push_back({key, exec, st});
                    HIP_TRY(hipGraphLaunch(exec, st));
                    return RT_OK;
                }
                if (graph) hipGraphDestroy(graph);
                (void)hipGetLastError();
                s->graph_ok = false;            // this runtime cannot capture the schedule: launch directly from now on
                if (crc != RT_OK && crc != RT_ERR_DEVICE) return crc;
                J.n_pass_used = 0; J.n_timed = 0;
            }
        }
        return enqueue_passes(J, 0, total_px, pass_px, nullptr);
    }

    if (stats) HIP_TRY(hipEventRecord(s->ev_begin, st));
    rc = enqueue_passes(J, 0, total_px, pass_px, &recs);
    if (rc != RT_OK) return rc;

    // Synchronous: read the counters back, re-run overflowed passes at half the size until they fit.
    size_t checked = 0;
    uint32_t retries = 0;
    std::vector<PassRec> final_recs;
    const size_t first_passes = recs.size();      // the frame's own passes; what follows them re-runs overflowed ones
    uint64_t px_done = 0;
    while (checked < recs.size()) {
        const size_t upto = recs.size();
        if (J.progress)      // pixels of the frame's passes as they complete (a pass that is re-run later was counted once)
            for (size_t i = checked; i < upto; ++i) {
                HIP_TRY(hipEventSynchronize(s->ev_pass[i]));
                if (i < first_passes) {
                    px_done = std::min<uint64_t>(total_px, px_done + recs[i].npx);
                    if (px_done < total_px) J.progress(J.progress_user, px_done, total_px);      // (the last one: when the frame is good, below)
                }
            }
        HIP_TRY(hipMemcpyAsync(s->h_counters, s->d_counters, sizeof(PassCounters) * J.n_pass_used, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        for (size_t i = checked; i < upto; ++i) {
            const PassRec r = recs[i];
            if (!s->h_counters[r.slot].overflow) { final_recs.push_back(r); continue; }
            ++retries;
            // how much more room than it had does the fullest sub-pool need, as far as the pass got? (The counters hold
            // the ATTEMPTED appends; levels behind the overflow are missing, so this is a lower bound.)
            uint64_t worst = 0;
            for (uint32_t k = 0; k < RT_SHARDS; ++k) {
                uint64_t n = 0;
                for (int lv = 1; lv < RT_MAX_LEVELS; ++lv) n += s->h_counters[r.slot].sc[k].level[lv];
                worst = std::max(worst, n);
            }
            uint32_t cut = 2;
            while (uint64_t(cut) * s->sub_cap < worst * 2 && cut < 1024u) cut *= 2;
            uint32_t part = std::max<uint32_t>(64u, ((r.npx / cut) + 63u) & ~63u);
            if (part >= r.npx) {
                // a 64-pixel pass cannot be cut: the pool itself has to grow (as if for a larger frame), up to the size a
                // full pass gets. Everything enqueued so far must be done before the old arrays go.
                if (s->ws_n0 >= RT_POOL_MAX_N0) { rt_set_error("bounce queue overflow on a 64-pixel pass with the largest pool"); return RT_ERR_OVERFLOW; }
                HIP_TRY(hipStreamSynchronize(st));
                s->ws_min_n0 = uint32_t(std::min<uint64_t>(RT_POOL_MAX_N0, std::max<uint64_t>(uint64_t(s->ws_n0) * 8u, 65536u)));
                if ((rc = ensure_workspace(s, s->ws_min_n0)) != RT_OK) return rc;
                LevelArgs& Bn = J.base;
                Bn.sub_cap = s->sub_cap;
                Bn.rays = s->rays; Bn.node_a = s->node_a; Bn.node_b = s->node_b; Bn.hit_n = s->hit_n; Bn.hit_p = s->hit_p; Bn.queue_e = s->queue_e; Bn.queue_s = s->queue_s;
                part = r.npx;
            }
            const uint32_t n_new = (r.npx + part - 1) / part;
            if (J.n_pass_used + n_new > RT_MAX_PASSES) { rt_set_error("out of pass slots while retrying"); return RT_ERR_OVERFLOW; }
            HIP_TRY(hipMemsetAsync(s->d_counters + J.n_pass_used, 0, sizeof(PassCounters) * n_new, st));
            rc = enqueue_passes(J, r.px0, r.npx, part, &recs);
            if (rc != RT_OK) return rc;
            // remember it: the next frames of this scene start with passes that fit
            uint32_t lg = 0;
            while ((full_px >> lg) > part && (full_px >> lg) > 64u && lg < 16) ++lg;
            s->shrink_log2 = std::max(s->shrink_log2, lg);
        }
        checked = upto;
    }
    s->last_overflow = 0;
    *s->h_overflow = 0;
    if (stats) {
        HIP_TRY(hipEventRecord(s->ev_end, st));
        HIP_TRY(hipStreamSynchronize(st));
        rt_work w[RT_N_KINDS]{};
        for (const PassRec& r : final_recs) {
            const PassCounters& pc = s->h_counters[r.slot];
            // level 0: every valid sample is one trace() call; deeper levels: appended entries
            for (int k = 0; k < RT_STAT_SHARDS; ++k) stats->shadow_rays += pc.shadow[k];
            // (k_tail counts every entry of its subtrees in the level behind its roots and keeps the true levels apart)
            for (int lv = 1; lv < RT_MAX_LEVELS; ++lv) stats->level_rays[lv] += (pc.tail_levels[lv] ? pc.tail_levels[lv] : pc.appended(lv)) - pc.cancel[lv];
            for (int sh = 0; sh < RT_WORK_SHARDS; ++sh) for (int k = 0; k < RT_N_KINDS; ++k) add_work(w[k], pc.work[sh][k]);
        }
        for (int k = 0; k < RT_N_KINDS; ++k) s->last_work[k] = w[k];
        stats->passes = uint32_t(final_recs.size());
        stats->retries = retries;
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, s->ev_begin, s->ev_end));
        stats->kernel_ms = ms;
        if ((rc = drain_timed(J, st)) != RT_OK) return rc;
        for (int k = 0; k < RT_N_KINDS; ++k) {
            stats->kernel_ms_by[k] += J.ms_by[k];
            stats->kernel_launches_by[k] += J.launches_by[k];
        }
    }
    return RT_OK;
}

// Level-0 trace() calls = valid pixels x spp (known on the host).
static void finish_stats(rt_stats* stats, uint64_t valid_px, uint32_t spp) {
    if (!stats) return;
    stats->level_rays[0] = valid_px * spp;
    stats->trace_calls = 0;
    for (int lv = 0; lv < RT_MAX_LEVELS; ++lv) stats->trace_calls += stats->level_rays[lv];
}

static PixelMap rect_map(const rt_camera* cam, uint32_t x0, uint32_t y0, uint32_t w, uint32_t h) {
    PixelMap pm{};
    pm.x0 = x0; pm.y0 = y0; pm.w = w; pm.h = h;
    pm.tile_w = (w + 7u) & ~7u; pm.tile_h = (h + 7u) & ~7u;
    pm.tiles_x = 1; pm.tiles_y = 1; pm.part = 0; pm.n_parts = 1; pm.n_owned = (w && h) ? 1 : 0;
    pm.tiled_out = 0; pm.stride_px = cam->imWidth;
    return pm;
}

static int check_rect(const rt_camera* cam, uint32_t x0, uint32_t y0, uint32_t w, uint32_t h) {
    if (uint64_t(x0) + w > cam->imWidth || uint64_t(y0) + h > cam->imHeight) {
        rt_set_error("rectangle (%u,%u %ux%u) outside the %ux%u image", x0, y0, w, h, cam->imWidth, cam->imHeight);
        return RT_ERR_INVALID;
    }
    if (uint64_t(((w + 7u) & ~7u)) * ((h + 7u) & ~7u) > 0x7FFFFFFFull) { rt_set_error("rectangle too large"); return RT_ERR_LIMIT; }
    return RT_OK;
}

// every camera of a batch: valid, and the same image size as the first
static int check_batch(const rt_camera* cams, uint32_t n_frames, const rt_params* prm) {
    if (!cams || n_frames == 0 || n_frames > RT_MAX_BATCH) { rt_set_error("a batch holds 1..%d frames", RT_MAX_BATCH); return RT_ERR_INVALID; }
    for (uint32_t b = 0; b < n_frames; ++b) {
        const int rc = check_params(cams + b, prm);
        if (rc != RT_OK) return rc;
        if (cams[b].imWidth != cams[0].imWidth || cams[b].imHeight != cams[0].imHeight) { rt_set_error("the frames of a batch must have one image size"); return RT_ERR_INVALID; }
    }
    return RT_OK;
}

extern "C" int rt_render_batch_device(rt_scene* s, const rt_camera* cams, uint32_t n_frames, const rt_params* prm, uint32_t x0,
                                      uint32_t y0, uint32_t w, uint32_t h, uint8_t* d_bgr, uint64_t frame_stride_bytes,
                                      void* stream, rt_stats* stats) {
    if (!s || !d_bgr) { rt_set_error("null argument"); return RT_ERR_INVALID; }
    int rc = check_batch(cams, n_frames, prm);
    if (rc != RT_OK) return rc;
    if ((rc = check_rect(cams, x0, y0, w, h)) != RT_OK) return rc;
    if (n_frames > 1 && frame_stride_bytes < uint64_t(cams->imWidth) * cams->imHeight * 3u) { rt_set_error("frame stride smaller than a frame"); return RT_ERR_INVALID; }
    hipStream_t st = stream ? static_cast<hipStream_t>(stream) : s->stream;
    rc = render_frame(s, cams, n_frames, prm, rect_map(cams, x0, y0, w, h), d_bgr, frame_stride_bytes, st, stats, stats != nullptr);
    if (rc == RT_OK) finish_stats(stats, uint64_t(w) * h * n_frames, 1 + prm->n_aa);
    return rc;
}

extern "C" int rt_render_device(rt_scene* s, const rt_camera* cam, const rt_params* prm, uint32_t x0, uint32_t y0,
                                uint32_t w, uint32_t h, uint8_t* d_bgr, void* stream, rt_stats* stats) {
    return rt_render_batch_device(s, cam, 1, prm, x0, y0, w, h, d_bgr, 0, stream, stats);
}

// rt_render_device for callers that cannot allocate: same kernels, same schedule, the node pool in the static array.
// No hipMalloc / hipFree / stream or event creation here or below (render_frame with noalloc: no ensure_workspace, no
// graph cache, no pass records); everything else it touches was created by rt_scene_create.
extern "C" int rt_render_device_noalloc(rt_scene* s, const rt_camera* cam, const rt_params* prm, uint32_t x0, uint32_t y0,
                                        uint32_t w, uint32_t h, uint8_t* d_bgr, void* stream) {
    if (!s || !d_bgr) { rt_set_error("null argument"); return RT_ERR_INVALID; }
    int rc = check_params(cam, prm);
    if (rc != RT_OK) return rc;
    if ((rc = check_rect(cam, x0, y0, w, h)) != RT_OK) return rc;
    if (!s->noalloc_ws) { rt_set_error("the static workspace was not set up for device %d", s->device); return RT_ERR_DEVICE; }
    const uint64_t samples = uint64_t((w + 7u) & ~7u) * ((h + 7u) & ~7u) * (1u + prm->n_aa);
    if (samples > RT_NOALLOC_MAX_SAMPLES) {
        rt_set_error("rt_render_device_noalloc: %llu samples (rectangle rounded up to multiples of 8, times %u samples per pixel) "
                     "above the maximum %u; use rt_render_device", (unsigned long long)samples, 1u + prm->n_aa, RT_NOALLOC_MAX_SAMPLES);
        return RT_ERR_LIMIT;
    }
    hipStream_t st = stream ? static_cast<hipStream_t>(stream) : s->stream;
    return render_frame(s, cam, 1, prm, rect_map(cam, x0, y0, w, h), d_bgr, 0, st, nullptr, false, true);
}

extern "C" int rt_render(rt_scene* s, const rt_camera* cam, const rt_params* prm, uint32_t x0, uint32_t y0, uint32_t w,
                         uint32_t h, uint8_t* bgr, rt_stats* stats) {
    return rt_render_progress(s, cam, prm, x0, y0, w, h, bgr, stats, nullptr, nullptr);
}

// rt_render that reports how far it is: progress(user, pixels done, pixels of the rectangle) after every pass of the
// frame (a frame of up to rt_scene_pass_samples() samples - 2^26 on an MI355X - is one pass) and once more, with done == total, when the bytes are in bgr.
extern "C" int rt_render_progress(rt_scene* s, const rt_camera* cam, const rt_params* prm, uint32_t x0, uint32_t y0, uint32_t w,
                                  uint32_t h, uint8_t* bgr, rt_stats* stats, rt_progress_fn progress, void* user) {
    if (!s || !bgr) { rt_set_error("null argument"); return RT_ERR_INVALID; }
    int rc = check_params(cam, prm);
    if (rc != RT_OK) return rc;
    if ((rc = check_rect(cam, x0, y0, w, h)) != RT_OK) return rc;
    if (stats) std::memset(stats, 0, sizeof *stats);
    if (w == 0 || h == 0) return RT_OK;
    HIP_TRY(hipSetDevice(s->device));
    const size_t need = size_t(cam->imWidth) * cam->imHeight * 3;
    if (need > s->frame_bytes) {
        hipFree(s->d_frame); s->d_frame = nullptr; s->frame_bytes = 0;
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&s->d_frame), need));
        s->frame_bytes = need;
    }
    // (progress counts the pixels of the rectangle rounded up to whole 8x8 blocks, as the passes do; scaled back below)
    struct Scale { rt_progress_fn fn; void* user; uint64_t rect_px, padded_px; } sc{progress, user, uint64_t(w) * h,
                                                                                   uint64_t((w + 7u) & ~7u) * ((h + 7u) & ~7u)};
    const rt_progress_fn scaled = [](void* u, uint64_t done, uint64_t) {
        const Scale* c = static_cast<const Scale*>(u);
        c->fn(c->user, std::min<uint64_t>(c->rect_px, done * c->rect_px / std::max<uint64_t>(c->padded_px, 1)), c->rect_px);
    };
    rc = render_frame(s, cam, 1, prm, rect_map(cam, x0, y0, w, h), s->d_frame, 0, s->stream, stats, true, false,
                      progress ? scaled : nullptr, &sc);
    if (rc != RT_OK) return rc;
    finish_stats(stats, uint64_t(w) * h, 1 + prm->n_aa);
    const size_t pitch = size_t(cam->imWidth) * 3, off = (size_t(y0) * cam->imWidth + x0) * 3;
    if (w == cam->imWidth)     // whole rows: one contiguous copy (pitched copies into pageable memory are several times slower)
        HIP_TRY(hipMemcpyAsync(bgr + off, s->d_frame + off, pitch * h, hipMemcpyDeviceToHost, s->stream));
    else
        HIP_TRY(hipMemcpy2DAsync(bgr + off, pitch, s->d_frame + off, pitch, size_t(w) * 3, h, hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    if (progress) progress(user, uint64_t(w) * h, uint64_t(w) * h);
    return RT_OK;
}

// Frames into HOST memory without blocking: rendered into the scene's device frame buffer, copied out by a D2H copy
// enqueued behind them on the same stream. With page-locked host memory the copy runs at the PCIe rate next to the
// NEXT call's kernels (another scene, another stream): the pipeline a caller builds from two scenes delivers host
// frames at nearly the device-framebuffer rate.
extern "C" int rt_render_batch_host_async(rt_scene* s, const rt_camera* cams, uint32_t n_frames, const rt_params* prm, uint32_t x0,
                                          uint32_t y0, uint32_t w, uint32_t h, uint8_t* host_bgr, uint64_t frame_stride_bytes,
                                          void* stream) {
    if (!s || !host_bgr) { rt_set_error("null argument"); return RT_ERR_INVALID; }
    int rc = check_batch(cams, n_frames, prm);
    if (rc != RT_OK) return rc;
    if ((rc = check_rect(cams, x0, y0, w, h)) != RT_OK) return rc;
    const size_t frame = size_t(cams->imWidth) * cams->imHeight * 3;
    if (n_frames > 1 && frame_stride_bytes < frame) { rt_set_error("frame stride smaller than a frame"); return RT_ERR_INVALID; }
    if (w == 0 || h == 0) return RT_OK;
    HIP_TRY(hipSetDevice(s->device));
    hipStream_t st = stream ? static_cast<hipStream_t>(stream) : s->stream;
    const size_t need = frame * n_frames;
    if (!s->copy_stream) {
        HIP_TRY(hipStreamCreateWithFlags(&s->copy_stream, hipStreamNonBlocking));
        for (int k = 0; k < 2; ++k) {
            HIP_TRY(hipEventCreateWithFlags(&s->ev_rendered[k], hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&s->ev_copied[k], hipEventDisableTiming));
        }
    }
    if (need > s->host_frames_bytes) {
        HIP_TRY(hipStreamSynchronize(s->copy_stream));      // copies out of the old buffers
        for (int k = 0; k < 2; ++k) {
            hipFree(s->d_host_frames[k]); s->d_host_frames[k] = nullptr;
        }
        s->host_frames_bytes = 0;
        for (int k = 0; k < 2; ++k) HIP_TRY(hipMalloc(reinterpret_cast<void**>(&s->d_host_frames[k]), need));
        s->host_frames_bytes = need;
        s->host_turn = 0;
    }
    const uint32_t k = s->host_turn++ & 1u;
    uint8_t* d_frames = s->d_host_frames[k];
    HIP_TRY(hipStreamWaitEvent(st, s->ev_copied[k], 0));      // the copy that last read this buffer (none yet: no wait)
    rc = render_frame(s, cams, n_frames, prm, rect_map(cams, x0, y0, w, h), d_frames, frame, st, nullptr, false);
    if (rc != RT_OK) return rc;
    HIP_TRY(hipEventRecord(s->ev_rendered[k], st));
    HIP_TRY(hipStreamWaitEvent(s->copy_stream, s->ev_rendered[k], 0));
    const size_t pitch = size_t(cams->imWidth) * 3, off = (size_t(y0) * cams->imWidth + x0) * 3;
    if (w == cams->imWidth && h == cams->imHeight && (n_frames == 1 || frame_stride_bytes == frame)) {
        HIP_TRY(hipMemcpyAsync(host_bgr, d_frames, need, hipMemcpyDeviceToHost, s->copy_stream));      // whole frames back to back: one copy
    } else {
        for (uint32_t b = 0; b < n_frames; ++b) {
            uint8_t* dst = host_bgr + size_t(frame_stride_bytes) * b + off;
            const uint8_t* src = d_frames + frame * b + off;
            if (w == cams->imWidth) HIP_TRY(hipMemcpyAsync(dst, src, pitch * h, hipMemcpyDeviceToHost, s->copy_stream));
            else HIP_TRY(hipMemcpy2DAsync(dst, pitch, src, pitch, size_t(w) * 3, h, hipMemcpyDeviceToHost, s->copy_stream));
        }
    }
    HIP_TRY(hipEventRecord(s->ev_copied[k], s->copy_stream));
    return RT_OK;
}

extern "C" int rt_pinned_alloc(size_t bytes, void** out) {
    if (!out) { rt_set_error("out is NULL"); return RT_ERR_INVALID; }
    *out = nullptr;
    HIP_TRY(hipHostMalloc(out, std::max<size_t>(bytes, 16), hipHostMallocDefault));
    return RT_OK;
}
extern "C" void rt_pinned_free(void* p) { if (p) hipHostFree(p); }

extern "C" uint32_t rt_tiles_owned(uint32_t W, uint32_t H, uint32_t tw, uint32_t th, uint32_t part, uint32_t n_parts) {
    if (!tw || !th || !n_parts || part >= n_parts) return 0;
    const uint64_t n = uint64_t((W + tw - 1) / tw) * ((H + th - 1) / th);
    return n > part ? uint32_t((n - part + n_parts - 1) / n_parts) : 0;
}

extern "C" int rt_render_tiles_device(rt_scene* s, const rt_camera* cam, const rt_params* prm, uint32_t tw, uint32_t th,
                                      uint32_t part, uint32_t n_parts, uint8_t* d_tiles, void* stream, rt_stats* stats) {
    return rt_render_tiles_batch_device(s, cam, 1, prm, tw, th, part, n_parts, d_tiles, 0, stream, stats);
}

extern "C" int rt_render_tiles_batch_device(rt_scene* s, const rt_camera* cam, uint32_t n_frames, const rt_params* prm,
                                            uint32_t tw, uint32_t th, uint32_t part, uint32_t n_parts, uint8_t* d_tiles,
                                            uint64_t frame_stride_bytes, void* stream, rt_stats* stats) {
    if (!s || !d_tiles) { rt_set_error("null argument"); return RT_ERR_INVALID; }
    int rc = check_batch(cam, n_frames, prm);
    if (rc != RT_OK) return rc;
    if (!tw || !th || (tw & 7u) || (th & 7u)) { rt_set_error("tile size must be a positive multiple of 8"); return RT_ERR_INVALID; }
    if (!n_parts || part >= n_parts) { rt_set_error("part %u / n_parts %u invalid", part, n_parts); return RT_ERR_INVALID; }
    PixelMap pm{};
    pm.x0 = 0; pm.y0 = 0; pm.w = cam->imWidth; pm.h = cam->imHeight;
    pm.tile_w = tw; pm.tile_h = th;
    pm.tiles_x = (cam->imWidth + tw - 1) / tw; pm.tiles_y = (cam->imHeight + th - 1) / th;
    pm.part = part; pm.n_parts = n_parts;
    pm.n_owned = rt_tiles_owned(cam->imWidth, cam->imHeight, tw, th, part, n_parts);
    pm.tiled_out = 1; pm.stride_px = 0;
    hipStream_t st = stream ? static_cast<hipStream_t>(stream) : s->stream;
    if (n_frames > 1 && frame_stride_bytes < uint64_t(pm.n_owned) * tw * th * 3u) { rt_set_error("frame stride smaller than a part's tiles"); return RT_ERR_INVALID; }
    rc = render_frame(s, cam, n_frames, prm, pm, d_tiles, frame_stride_bytes, st, stats, stats != nullptr);
    if (rc == RT_OK && stats) {
        // valid pixels of the owned tiles
        uint64_t valid = 0;
        for (uint32_t j = 0; j < pm.n_owned; ++j) {
            const uint32_t T = part + j * n_parts, tx = T % pm.tiles_x, ty = T / pm.tiles_x;
            const uint32_t vw = std::min(tw, cam->imWidth - tx * tw), vh = std::min(th, cam->imHeight - ty * th);
            valid += uint64_t(vw) * vh;
        }
        finish_stats(stats, valid * n_frames, 1 + prm->n_aa);
    }
    return rc;
}

extern "C" int rt_untile_device(int device, uint32_t W, uint32_t H, uint32_t tw, uint32_t th, uint32_t n_parts,
                                uint32_t part_stride_tiles, const uint8_t* d_tiles, uint8_t* d_bgr, void* stream) {
    return rt_untile_batch_device(device, W, H, tw, th, n_parts, part_stride_tiles, 1, 0, 0, d_tiles, d_bgr, stream);
}

extern "C" int rt_untile_batch_device(int device, uint32_t W, uint32_t H, uint32_t tw, uint32_t th, uint32_t n_parts,
                                      uint32_t part_stride_tiles, uint32_t n_frames, uint64_t in_frame_stride_bytes,
                                      uint64_t out_frame_stride_bytes, const uint8_t* d_tiles, uint8_t* d_bgr, void* stream) {
    if (!d_tiles || !d_bgr || !tw || !th || !n_parts || !W || !H || !n_frames || n_frames > RT_MAX_BATCH) { rt_set_error("bad argument"); return RT_ERR_INVALID; }
    HIP_TRY(hipSetDevice(device));
    UntileArgs U{};
    U.W = W; U.H = H; U.tile_w = tw; U.tile_h = th;
    U.tiles_x = (W + tw - 1) / tw;
    U.n_tiles = U.tiles_x * ((H + th - 1) / th);
    U.n_parts = n_parts;
    U.part_stride = part_stride_tiles;
    U.n_frames = n_frames; U.in_stride = in_frame_stride_bytes; U.out_stride = out_frame_stride_bytes;
    launch_untile(U, d_tiles, d_bgr, static_cast<hipStream_t>(stream));
    HIP_TRY(hipGetLastError());
    return RT_OK;
}

// ---------------------------------------------------------------------------------------------------------
// unit entry points (parity tests)
// ---------------------------------------------------------------------------------------------------------
namespace {
struct DevBuf {
    void* p = nullptr;
    ~DevBuf() { if (p) hipFree(p); }
    int alloc(size_t bytes) { return hipMalloc(&p, std::max<size_t>(bytes, 16)) == hipSuccess ? 0 : -1; }
    template <typename T> T* as() { return static_cast<T*>(p); }
};
}  // namespace

extern "C" int rt_cast_rays(int device, const rt_camera* cam, const rt_params* prm, const uint32_t* xy, uint32_t n, float* out) {
    int rc = check_params(cam, prm);
    if (rc != RT_OK) return rc;
    if (!xy || !out) { rt_set_error("null argument"); return RT_ERR_INVALID; }
    if (n == 0) return RT_OK;
    HIP_TRY(hipSetDevice(device));
    const uint32_t spp = 1 + prm->n_aa;
    DevBuf daa, dxy, dout;
    if (daa.alloc(sizeof(float) * 2 * prm->n_aa) || dxy.alloc(sizeof(uint32_t) * 2 * n) || dout.alloc(sizeof(float) * 6 * size_t(n) * spp)) {
        rt_set_error("hipMalloc failed"); return RT_ERR_DEVICE;
    }
    if (prm->n_aa) HIP_TRY(hipMemcpy(daa.p, prm->aa_xy, sizeof(float) * 2 * prm->n_aa, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dxy.p, xy, sizeof(uint32_t) * 2 * n, hipMemcpyHostToDevice));
    launch_cast_rays(to_dcamera(cam), daa.as<float>(), spp, dxy.as<uint32_t>(), n, dout.as<float>(), nullptr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out, dout.p, sizeof(float) * 6 * size_t(n) * spp, hipMemcpyDeviceToHost));
    return RT_OK;
}

static int check_form(int form) {
    if (form < RT_TRAVERSE_QUAD || form > RT_TRAVERSE_SERVE) { rt_set_error("traversal form %d outside [0,2]", form); return RT_ERR_INVALID; }
    return RT_OK;
}

extern "C" int rt_intersect(rt_scene* s, int object, const float* rays, uint32_t n, int32_t* hit, float* dist, float* nrm) {
    return rt_intersect_form(s, RT_TRAVERSE_QUAD, object, rays, n, hit, dist, nrm);
}

extern "C" int rt_intersect_form(rt_scene* s, int form, int object, const float* rays, uint32_t n, int32_t* hit, float* dist, float* nrm) {
    if (!s || !rays || !hit || !dist || !nrm) { rt_set_error("null argument"); return RT_ERR_INVALID; }
    if (check_form(form) != RT_OK) return RT_ERR_INVALID;
    if (object >= int(s->d.n_objects)) { rt_set_error("object %d out of range", object); return RT_ERR_INVALID; }
    if (n == 0) return RT_OK;
    HIP_TRY(hipSetDevice(s->device));
    DevBuf dr, dh, dd, dn;
    if (dr.alloc(sizeof(float) * 6 * n) || dh.alloc(sizeof(int32_t) * n) || dd.alloc(sizeof(float) * n) || dn.alloc(sizeof(float) * 3 * n)) {
        rt_set_error("hipMalloc failed"); return RT_ERR_DEVICE;
    }
    HIP_TRY(hipMemcpy(dr.p, rays, sizeof(float) * 6 * n, hipMemcpyHostToDevice));
    launch_intersect(s->d, form, object, dr.as<float>(), n, dh.as<int32_t>(), dd.as<float>(), dn.as<float>(), nullptr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(hit, dh.p, sizeof(int32_t) * n, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(dist, dd.p, sizeof(float) * n, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(nrm, dn.p, sizeof(float) * 3 * n, hipMemcpyDeviceToHost));
    return RT_OK;
}

extern "C" int rt_occluded(rt_scene* s, const float* rays, const float* max_dist, uint32_t n, uint8_t* occ) {
    return rt_occluded_form(s, RT_TRAVERSE_QUAD, rays, max_dist, n, occ);
}

extern "C" int rt_occluded_form(rt_scene* s, int form, const float* rays, const float* max_dist, uint32_t n, uint8_t* occ) {
    if (!s || !rays || !max_dist || !occ) { rt_set_error("null argument"); return RT_ERR_INVALID; }
    if (check_form(form) != RT_OK) return RT_ERR_INVALID;
    if (n == 0) return RT_OK;
    HIP_TRY(hipSetDevice(s->device));
    DevBuf dr, dm, dout;
    if (dr.alloc(sizeof(float) * 6 * n) || dm.alloc(sizeof(float) * n) || dout.alloc(n)) { rt_set_error("hipMalloc failed"); return RT_ERR_DEVICE; }
    HIP_TRY(hipMemcpy(dr.p, rays, sizeof(float) * 6 * n, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dm.p, max_dist, sizeof(float) * n, hipMemcpyHostToDevice));
    launch_occluded(s->d, form, dr.as<float>(), dm.as<float>(), n, dout.as<uint8_t>(), nullptr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(occ, dout.p, n, hipMemcpyDeviceToHost));
    return RT_OK;
}

// One chunk of rt_trace: rays [first, first+n) at level 0. Sets *overflow instead of failing.
static int trace_chunk(rt_scene* s, int form, const rt_params* prm, const float* d_rays, uint32_t n, float* d_rgb, rt_stats* stats,
                       bool* overflow) {
    hipStream_t st = s->stream;
    // RT_TRAVERSE_PACKET / _SERVE: every level's rays and hits go through that traversal in the per-lane kernels, nothing
    // is suspended (no quad launches); consecutive caller rays form the packets
    DScene D = s->d;
    if (form == RT_TRAVERSE_PACKET) { D.packet = 2; D.user_packet = 1; }
    const int wide = form == RT_TRAVERSE_SERVE ? RT_FORM_WIDE_SERVE : RT_FORM_WIDE;
    LevelArgs A;
    std::memset(&A, 0, sizeof A);
    A.n0 = n; A.max_depth = prm->max_depth; A.spp = 1;
    A.sc = s->d_counters->sc; A.cancelled = s->d_counters->cancel; A.overflow = &s->d_counters->overflow;
    A.shadow_count = s->d_counters->shadow;
    A.work = &s->d_counters->work[0][0][0];
    A.node_a = s->node_a; A.node_b = s->node_b; A.hit_n = s->hit_n; A.hit_p = s->hit_p; A.user_rays = d_rays;
    A.queue_e = s->queue_e; A.queue_s = s->queue_s;
    // pool entry i >= n0 keeps its ray at rays[2*(i-n0)]: the sub-pools are the same size whatever n0 <= ws_n0 is
    A.rays = s->rays;
    A.sub_cap = s->sub_cap;
    A.chunk_log2 = RT_SHARD_CHUNK_LOG2;
    while (A.chunk_log2 > 0 && (A.n0 >> (6u + A.chunk_log2)) < RT_SHARDS) --A.chunk_log2;
    A.host_overflow = nullptr;
    FrameParamsBatch fp;
    std::memset(&fp, 0, sizeof fp);
    std::memcpy(fp.f[0].background, prm->background, sizeof fp.f[0].background);
    launch_set_frame_params(fp, 1, s->d_fp, s->d_counters, sizeof(PassCounters), st);
    A.fp = s->d_fp;
    const int levels = s->has_bounce ? A.max_depth : 0;
    const bool meshes = s->n_models > 0 && form == RT_TRAVERSE_QUAD;      // meshes that suspend rays
    const uint32_t gq = std::min<uint32_t>((n + 15) / 16, 4u * RT_LEVEL_GRID);
    // per-lane kernels + resumed queues at every level (level sizes are unknown here), one shade pass over all levels
    for (int lv = 0; lv <= levels; ++lv) {
        A.level = lv; A.level_hi = lv;
        launch_extend(lv == 0 ? 2 : 1, wide, s->count_work, D, A, lv == 0 ? (n + RT_BLOCK - 1) / RT_BLOCK : RT_LEVEL_GRID, st);
        if (meshes) launch_extend(lv == 0 ? 2 : 1, RT_FORM_QUAD_RESUME, s->count_work, D, A, lv == 0 ? gq : 4u * RT_LEVEL_GRID, st);
    }
    for (int part = 0; part < (levels ? 2 : 1); ++part) {      // level 0 (sample order), then levels 1.. (the sub-pools)
        A.level = part; A.level_hi = part ? levels : 0;
        launch_shade(wide, s->count_work, D, A, part ? RT_LEVEL_GRID : (n + RT_BLOCK - 1) / RT_BLOCK, st);
        if (meshes) launch_shade(RT_FORM_QUAD_RESUME, s->count_work, D, A, 4u * RT_LEVEL_GRID, st);
    }
    if (meshes) {
        for (int lv = 0; lv <= levels; ++lv) {      // the hits of the rays each level parked
            A.level = lv; A.level_hi = lv;
            launch_shade(RT_FORM_QUAD_PARKED, s->count_work, D, A, lv == 0 ? gq : 4u * RT_LEVEL_GRID, st);
        }
    }
    for (int lv = levels - 1; lv >= 1; --lv) { A.level = lv; A.level_hi = lv; launch_fold(s->d, A, 1024, st); }
    A.level = 0; A.level_hi = 0;
    launch_resolve_rgb(s->d, A, d_rgb, st);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(s->h_counters, s->d_counters, sizeof(PassCounters), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    const PassCounters& pc = s->h_counters[0];
    *overflow = pc.overflow != 0;
    if (!*overflow && stats) {
        stats->level_rays[0] += n;
        for (int lv = 1; lv < RT_MAX_LEVELS; ++lv) stats->level_rays[lv] += pc.appended(lv) - pc.cancel[lv];
        for (int k = 0; k < RT_STAT_SHARDS; ++k) stats->shadow_rays += pc.shadow[k];
        stats->passes += 1;
        for (int sh = 0; sh < RT_WORK_SHARDS; ++sh) for (int k = 0; k < RT_N_KINDS; ++k) add_work(s->last_work[k], pc.work[sh][k]);
    }
    return RT_OK;
}

extern "C" int rt_trace(rt_scene* s, const rt_params* prm, const float* rays, uint32_t n, float* rgb, rt_stats* stats) {
    return rt_trace_form(s, RT_TRAVERSE_QUAD, prm, rays, n, rgb, stats);
}

extern "C" int rt_trace_form(rt_scene* s, int form, const rt_params* prm, const float* rays, uint32_t n, float* rgb, rt_stats* stats) {
    if (!s || !prm || !rays || !rgb) { rt_set_error("null argument"); return RT_ERR_INVALID; }
    if (check_form(form) != RT_OK) return RT_ERR_INVALID;
    if (s->n_models == 0) form = RT_TRAVERSE_QUAD;      // nothing to traverse: one schedule
    if (prm->max_depth < 0 || prm->max_depth >= RT_MAX_LEVELS) { rt_set_error("max_depth %d outside [0,%d]", prm->max_depth, RT_MAX_LEVELS - 1); return RT_ERR_LIMIT; }
    if (stats) std::memset(stats, 0, sizeof *stats);
    if (n == 0) return RT_OK;
    HIP_TRY(hipSetDevice(s->device));
    uint32_t chunk = std::min<uint32_t>(n, RT_PASS_SAMPLES);
    int rc = ensure_workspace(s, (chunk + 63u) & ~63u);
    if (rc != RT_OK) return rc;
    DevBuf dr, dout;
    if (dr.alloc(sizeof(float) * 6 * size_t(n)) || dout.alloc(sizeof(float) * 3 * size_t(n))) { rt_set_error("hipMalloc failed"); return RT_ERR_DEVICE; }
    HIP_TRY(hipMemcpy(dr.p, rays, sizeof(float) * 6 * size_t(n), hipMemcpyHostToDevice));
    for (rt_work& w : s->last_work) w = rt_work{};
    for (uint32_t done = 0; done < n;) {
        const uint32_t cnt = std::min(chunk, n - done);
        bool overflow = false;
        rc = trace_chunk(s, form, prm, dr.as<float>() + 6 * size_t(done), cnt, dout.as<float>() + 3 * size_t(done), stats, &overflow);
        if (rc != RT_OK) return rc;
        if (overflow) {        // the ray trees of this chunk need more pool than it has: halve the chunk and redo it
            if (chunk <= 64) { rt_set_error("bounce queue overflow on a 64-ray chunk"); return RT_ERR_OVERFLOW; }
            chunk = (chunk / 2 + 63u) & ~63u;
            if (stats) stats->retries += 1;
            continue;
        }
        done += cnt;
    }
    HIP_TRY(hipMemcpy(rgb, dout.p, sizeof(float) * 3 * size_t(n), hipMemcpyDeviceToHost));
    if (stats) for (int lv = 0; lv < RT_MAX_LEVELS; ++lv) stats->trace_calls += stats->level_rays[lv];
    return RT_OK;
}

extern "C" const char* rt_last_error(void) { return g_err; }
extern "C" const char* rt_version(void) { return "raytracer_amd 0.1 (gfx950)"; }
