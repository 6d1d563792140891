// rt_launch.h — kernel argument blocks and launcher prototypes shared by rt_kernels.hip and rt_api.cpp.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rt_device.h"

#define RT_BLOCK 64          // one wave per workgroup: waves retire independently, no barrier is ever needed
#define RT_MAX_LEVELS 16     // recursion levels 0..max_depth, max_depth <= 15
#define RT_N_KINDS 5          // k_extend_wide, k_shade_wide, k_extend_quad, k_shade_quad<0|1>, k_shade_quad<2>
#define RT_WORK_SHARDS 64      // COUNT builds: [shard][kind][RT_WORK_WORDS] words
#define RT_WORK_WORDS 14
#define RT_STAT_SHARDS 256    // statistics atomics are spread over this many words: one word saturates at ~90 atomics/us

// Scene in HBM (SoA). All pointers are device pointers owned by rt_scene.
struct DScene {
    const uint4* objects;        // 2 per object, scene order: {type, index, material, sphere 1/r bits}, {4 float params}
    uint32_t n_objects;
    uint32_t n_lights;
    uint32_t debug;              // timing experiments only (RT_DEBUG_SKIP)
    uint32_t generic_pow;        // a material's specRefExp is not an integer in [0, RT_POW_MAX_INT]: shade kernels with pow()
    uint32_t packet;             // level 0 (primary rays and their shadow rays) traverses meshes as 64-ray packets in the per-lane kernels
    uint32_t user_packet;        // rt_trace_form only: caller-supplied rays (SRC 2) and their hits traverse meshes as packets too
    const DModel* models;
    const rt_material* materials;
    const float4* lights;        // 2 per light: pos, color
    const Bvh4Node* nodes;
    const LeafBlock* leaves;
    const TriNormals* normals;   // original face order
};

struct DCamera {
    float eye[3], up[3], right[3], at[3];
    float aspect;
    uint32_t imWidth, imHeight;
    uint32_t pad;
    double tan_half;             // tan(deg2rad(fov) * 0.5) evaluated in double on the host (camera.h:60-61)
};

// Which pixels a pass covers and where their bytes go. Pixel ordinal q -> owned tile j = q / (tile_w*tile_h),
// global tile T = part + j*n_parts (row-major over tiles_x x tiles_y), 8x8 bit-interleaved blocks inside a tile.
struct PixelMap {
    uint32_t x0, y0, w, h;       // rectangle the tiles are laid over
    uint32_t tile_w, tile_h;     // multiples of 8
    uint32_t tiles_x, tiles_y;
    uint32_t part, n_parts;
    uint32_t n_owned;            // tiles this part owns
    uint32_t tiled_out;          // 1: tile-major output buffer, 0: row-major full image
    uint32_t stride_px;          // row-major: pixels per image row
    uint32_t z_per_tile, z_sub_x, z_tiles_x;   // rt_recip32 of tile_w*tile_h, tile_w/8, tiles_x (rt_fill_recips)
};

// floor(2^32 / d) (d = 1: 2^32 - 1): with it n / d for any 32-bit n is one multiply-high and one correction step
// (rt_kernels.hip: udivmod32) instead of the ~20-instruction expansion of a 32-bit division or the ~100 (half of them
// scalar) of a 64-bit one. The sample index -> frame, pixel, tile arithmetic at the top of every level-0 wave was five
// such divisions, three of them 64-bit: a sixth of the level-0 closest-hit kernel's instructions.
static inline uint32_t rt_recip32(uint32_t d) { return d <= 1u ? 0xFFFFFFFFu : uint32_t((1ull << 32) / d); }
static inline void rt_fill_recips(PixelMap& pm) {
    pm.z_per_tile = rt_recip32(pm.tile_w * pm.tile_h); pm.z_sub_x = rt_recip32(pm.tile_w >> 3); pm.z_tiles_x = rt_recip32(pm.tiles_x);
}

// Pool entry i = one trace() invocation ("node"). Entries [0, n0) are level 0 (sample order). The rest of the pool is
// RT_SHARDS equal sub-pools; in each, level k >= 1 follows level k-1. An entry's shard is fixed for the whole tree:
// a level-0 entry's by its position (rt_shard_of_level0), a child's is its parent's. Every append counter exists once
// per shard, 256 B apart: a returning atomic on ONE address costs 11.4 ns on MI355X and they serialise
// (tools/atomic_rate.cpp: 65,536 one-per-wave appends = 0.75 ms on one counter, 0.02 ms on 64), which made the
// per-lane level-0 kernels atomic-bound.
//   k_extend writes  node_b (children, material, flags), hit_n, hit_p for a hit; node_a for a miss
//   k_shade  writes  node_a for a hit (light loop of trace()), may downgrade node_b.w (the `return red` exit)
//   k_fold / k_resolve read node_a / node_b
// What changes from frame to frame at a fixed resolution: it lives in device memory (written by k_set_frame_params at
// the start of every frame) so that the launches of a frame do not depend on it and can be replayed as a hipGraph.
struct FrameParams {
    float background[3];
    float pad;
    DCamera cam;
};
// A call may render a BATCH of frames that differ only in their cameras: the level-0 samples of frame b follow those
// of frame b-1, so every launch works on n_frames times as many rays (a small frame, or one rank's share of a frame,
// leaves most of the chip idle in its deep levels and is bound by the host's launch rate).
#define RT_MAX_BATCH 32
struct FrameParamsBatch { FrameParams f[RT_MAX_BATCH]; };

#define RT_POW_MAX_INT 64
#define RT_SHARDS 32u
#define RT_SHARD_CHUNK_LOG2 4u      // level 0: runs of 2^k consecutive 64-entry blocks share a shard (LevelArgs::chunk_log2)
#define RT_SHARD_CHUNK_LOG2_MAX 8u
struct ShardCounters {              // 256 B
    uint32_t level[RT_MAX_LEVELS + 4];  // entries appended per level (index 0 unused)
    uint32_t qe[RT_MAX_LEVELS + 4];     // per level: rays k_extend_wide queued for k_extend_quad
    uint32_t qs[RT_MAX_LEVELS + 4];     // per level: hits k_shade_wide queued for k_shade_quad
    uint32_t pad[64 - 3 * (RT_MAX_LEVELS + 4)];
};
static_assert(sizeof(ShardCounters) == 256, "one shard's counters fill 256 bytes");
static inline __host__ __device__ uint32_t rt_shard_of_level0(uint32_t i, uint32_t chunk_log2) { return (i >> (6u + chunk_log2)) & (RT_SHARDS - 1u); }

struct LevelArgs {
    uint32_t n0;                 // level-0 samples of this pass (multiple of spp)
    int32_t level;               // k_extend / k_fold: the level; k_shade: first level of the range
    int32_t level_hi;            // k_shade: last level of the range (inclusive)
    int32_t max_depth;
    uint32_t spp;
    uint32_t sub_cap;            // entries of one sub-pool; the pool holds n0 + RT_SHARDS * sub_cap
    uint64_t sample0;            // first sample of the pass in the frame's sample order
    ShardCounters* sc;           // [RT_SHARDS]
    uint32_t* cancelled;         // [RT_MAX_LEVELS+1] per level: speculative children whose parent took the red exit
    uint32_t* overflow;          // set to 1 if a child did not fit
    uint32_t* host_overflow;     // pinned host word: k_resolve raises it when the pass overflowed (asynchronous frames)
    unsigned long long* shadow_count;   // [RT_STAT_SHARDS], indexed by workgroup
    unsigned long long* work;    // [RT_WORK_SHARDS][RT_N_KINDS][RT_WORK_WORDS] traversal work counters (COUNT builds only)
    float4* rays;                // 2 per pool entry >= n0: origin, direction
    float4* node_a;              // per pool entry: r g b rCoef
    uint4* node_b;               // per pool entry: refl child, refr child, material, flags
    float4* hit_n;               // per pool entry with a hit: hitNormal xyz, tmpdist as the closest-hit loop left it
    float4* hit_p;               // per pool entry with a hit: hitPt xyz, rCoef
    uint2* queue_e;              // levels >= 1: rays k_extend_wide suspended {pool entry, object index}; the queue of
                                 // (level, shard) starts at the pool offset of that level in that sub-pool. (Level 0
                                 // keeps no queue: suspended entries are marked in node_b.w, see pick_marked.)
    uint2* queue_s;              // same for k_shade_wide {pool entry, light << 16 | object}
    const float* aa;             // anti-aliasing pattern (device)
    const float* user_rays;      // SRC 2
    const FrameParams* fp;       // [n_frames] camera + background of the frames (device); the background is the batch's
    uint32_t frame_samples;      // batch: level-0 samples per frame (a multiple of 64); 0 = one frame
    uint32_t fast32;             // this pass' sample indices fit 32 bits (sample0 + n0 <= 2^32 - 1): the divisions below by reciprocal
    uint32_t z_spp, z_frame_samples, frame_px, z_frame_px;   // rt_recip32 of spp, frame_samples, frame_samples / spp
    uint64_t out_stride;         // batch: bytes between the output buffers of consecutive frames
    uint32_t debug;              // timing experiments only (RT_DEBUG_SKIP)
    uint32_t chunk_log2;         // level-0 shard chunks (RT_SHARD_CHUNK_LOG2; RT_CHUNK_LOG2 in the environment for experiments)
    uint4* wave_log;             // diagnostics (COUNT builds, RT_WAVE_LOG): per workgroup {start, end, traversal ticks, batches}
    const uint32_t* order0;      // level-0 per-lane launches of a small one-pass frame: wave w takes group order0[w] - the groups in
                                 // descending order of what they cost in the previous frame (k_order_groups); null: group w
    uint32_t* cost0;             // ... and every wave leaves here how long its group took (10-ns ticks), for the next frame's order
    uint32_t wide_waves;         // waves per workgroup of the per-lane launches of this pass (0: the process default, one)
    const uint32_t* perm;        // k_shade_wide as packets over levels >= 1: the hits of the range sorted by position (launch_sort_hits),
    uint32_t perm_n;             // perm_n slots (a multiple of 64; 0xFFFFFFFF = no entry); null: pool order
    PixelMap pm;
};

// Hits of levels A.level..A.level_hi (>= 1) sorted by the cell of their hit point (sorted light loops, rt_kernels.hip).
// Slot d of the first n_slots slots of the range's dense enumeration (group d / 64 of the launch, lane d % 64) gets a
// key; the sorted entry numbers are the permutation k_shade_wide walks instead of the pool order.
struct SortBox { float lo[3]; float scale[3]; };      // cell (0..255 per axis) = (p - lo) * scale
size_t sort_hits_temp_bytes(uint32_t n_slots);
// keys[2], vals[2]: n_slots words each; returns the sorted entry numbers (one of vals[])
// axis_bits (1..10): bits per axis of the Morton key
const uint32_t* launch_sort_hits(const DScene& S, const LevelArgs& A, const SortBox& box, uint32_t axis_bits, uint32_t* const keys[2],
                                 uint32_t* const vals[2], void* temp, size_t temp_bytes, uint32_t n_slots, hipStream_t st);

struct UntileArgs {
    uint32_t W, H, tile_w, tile_h, tiles_x, n_tiles, n_parts;
    uint32_t part_stride;   // tile slots between parts; 0 = packed
    uint32_t n_frames;      // batch: frames per launch (blockIdx.y)
    uint64_t in_stride;     // batch: bytes between consecutive frames' tiles (within a part)
    uint64_t out_stride;    // batch: bytes between consecutive output frames
};

// ---- rt_render_device_noalloc: the frame workspace in a file-scope __device__ array (rt_kernels.hip: `scratch`) ----
// One pass of RT_NOALLOC_PASS_SAMPLES level-0 samples with TWICE the ordinary pool factor (a small pass can lie entirely on
// a glass sphere, where the ray trees are largest; the ordinary factor is an average over millions of samples): node_a,
// node_b, hit_n, hit_p (16 B per pool entry each), queue_e, queue_s (8 B each), rays (32 B per entry behind level 0)
// = 20.5 MiB.
#define RT_NOALLOC_PASS_SAMPLES (1u << 14)
#define RT_NOALLOC_POOL_FACTOR 12u
#define RT_NOALLOC_CAP (RT_NOALLOC_PASS_SAMPLES * RT_NOALLOC_POOL_FACTOR)
#define RT_NOALLOC_SCRATCH_BYTES (size_t(RT_NOALLOC_CAP) * (4u * 16u + 2u * 8u) + size_t(RT_NOALLOC_CAP - RT_NOALLOC_PASS_SAMPLES) * 32u)
// Device address of `scratch` on `device` (which must be the current device): hipGetSymbolAddress once per device,
// remembered. Returns hipSuccess or the error.
hipError_t noalloc_scratch(int device, void** ptr);

// Expensive groups first, the most expensive in parts (level-0 launches of a one-pass frame of at most RT_ORDER_MAX_GROUPS
// 64-sample groups; k_order_groups). An order entry = group | parts_log2 << 28 | part << 30: the wave takes lanes
// [part * 64 >> parts_log2, (part + 1) * 64 >> parts_log2) of the group (a whole group, a half or a quarter: 8x8, 8x4 or 4x4
// pixels); 0xFFFFFFFF: nothing. A launch with an order has RT_ORDER_SLOTS(n) waves. Costs: 4 words per group (one per
// quarter; a wave adds its life in 10-ns ticks to the word of its first quarter), cleared by k_order_groups.
#define RT_ORDER_MAX_GROUPS (1u << 16)
#ifndef RT_ORDER_SLOTS_SHIFT
#define RT_ORDER_SLOTS_SHIFT 4
#endif
#define RT_ORDER_SLOTS(n) ((n) + ((n) >> RT_ORDER_SLOTS_SHIFT))
#define RT_ORDER_NONE 0xFFFFFFFFu
// two workgroups, one per cost array (closest hits, light loops): cost_* 4 n words, order_* RT_ORDER_SLOTS(n) words, key_* n
// bytes of scratch
void launch_order_groups(uint32_t* cost_e, uint32_t* cost_s, uint32_t* order_e, uint32_t* order_s, uint8_t* key_e, uint8_t* key_s,
                         uint32_t n, hipStream_t st);

// Kernel forms: one ray per lane (suspends at meshes), one ray per quad from the start, one per quad resuming the queue.
// RT_FORM_QUAD_PARKED (shade only): one hit per quad from the start, for the rays of the level's extend queue.
// RT_FORM_WIDE_SERVE (levels >= 1): one ray / hit per lane like RT_FORM_WIDE, but nothing is suspended: the lanes whose ray
// reaches a mesh are served 16 at a time by the wave's own quads (serve_traverse).
// RT_FORM_WIDE_FUSED (extend only; scenes without meshes, not the counting build): RT_FORM_WIDE and the level's light loops in one
//   launch (k_trace_wide) - no shade launch follows.
enum { RT_FORM_WIDE = 0, RT_FORM_QUAD_FRESH = 1, RT_FORM_QUAD_RESUME = 2, RT_FORM_QUAD_PARKED = 3, RT_FORM_WIDE_SERVE = 4, RT_FORM_WIDE_FUSED = 5 };
void launch_extend(int src, int form, bool count, const DScene& S, const LevelArgs& A, uint32_t grid, hipStream_t st);
void launch_shade(int form, bool count, const DScene& S, const LevelArgs& A, uint32_t grid, hipStream_t st);
void launch_set_frame_params(const FrameParamsBatch& p, uint32_t n_frames, FrameParams* dst, void* clear, size_t clear_bytes, hipStream_t st);   // clear: 16-byte multiple
void launch_set_aa(const float* aa_xy, uint32_t n_points, float* dst, hipStream_t st);      // host pattern -> device, as kernel arguments
void launch_fold(const DScene& S, const LevelArgs& A, uint32_t grid, hipStream_t st);
// A.level = the tail's root level (>= 1): every ray of that level and its whole subtree in one launch, one ray per quad,
// depth first (k_tail); tail_levels: [RT_MAX_LEVELS + 4] per-level entry counts, zero at the start of the pass
void launch_tail(bool count, const DScene& S, const LevelArgs& A, uint32_t* tail_levels, uint32_t grid, hipStream_t st);
// folds the subtrees under the entries of level A.level (the tail's roots), one thread per root
void launch_fold_tree(const DScene& S, const LevelArgs& A, uint32_t grid, hipStream_t st);
void launch_resolve(const DScene& S, const LevelArgs& A, uint8_t* out, hipStream_t st);
void launch_resolve_rgb(const DScene& S, const LevelArgs& A, float* rgb, hipStream_t st);
void launch_untile(const UntileArgs& U, const uint8_t* tiles, uint8_t* out, hipStream_t st);
void launch_cast_rays(const DCamera& cam, const float* aa, uint32_t spp, const uint32_t* xy, uint32_t n, float* out,
                      hipStream_t st);
// form (RT_TRAVERSE_*): 0 one ray per quad, 1 one ray per lane as 64-ray packets, 2 one ray per lane with quad service
void launch_intersect(const DScene& S, int form, int object, const float* rays, uint32_t n, int32_t* hit, float* dist, float* nrm,
                      hipStream_t st);
void launch_occluded(const DScene& S, int form, const float* rays, const float* max_dist, uint32_t n, uint8_t* occ, hipStream_t st);
