/*
 * raytracer_amd.h — C-ABI of the MI355X render hot path.
 *
 * This is the drop-in boundary for the reference's `render()` free function
 * (reference src/raytracer.cpp:422-463) and everything below it: `trace()`
 * (src/raytracer.cpp:279-420), `Camera::castRays` (src/camera.h:57-79) and
 * `Primitive::intersect` for Sphere / Plane / Triangle / Model
 * (src/primitives.cpp:11-25, 42-126, 173-203, 364-384).
 *
 * The reference has no plugin/FFI layer; its seam is the C++ call
 *   render(img, x_start, y_start, cH, cW, camera, scene, background, finPix, thread_count)
 * The host surface (raytracer_amd/host/raytracer.hpp) keeps that signature and
 * flattens Scene + Camera into the plain structs below before calling in here.
 *
 * Plain pointers and sizes only; no C++ or torch types. All functions return
 * RT_OK (0) or a negative rt_status; rt_last_error() gives the message.
 */
#ifndef RAYTRACER_AMD_H
#define RAYTRACER_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum rt_status {
    RT_OK = 0,
    RT_ERR_INVALID = -1,    /* bad argument / malformed flat scene            */
    RT_ERR_DEVICE = -2,     /* HIP runtime error (message has hipGetErrorString) */
    RT_ERR_NO_DEVICE = -3,  /* no usable gfx950 device                         */
    RT_ERR_OVERFLOW = -4,   /* bounce queues could not be sized (internal)     */
    RT_ERR_LIMIT = -5       /* size above a stated maximum                     */
} rt_status;

/* Material: reference src/material.h:13-47 (same fields, bools as int32). */
typedef struct rt_material {
    float color[3];
    float ka, kd, ks;
    float kr, kt;
    float specRefExp;
    float refractiveIdx;
    int32_t reflective;
    int32_t refractive;
} rt_material;

/* Light: reference src/scene.h:15-27. color already multiplied by `factor` (src/scene.cpp:121). */
typedef struct rt_light {
    float pos[3];
    float color[3];
} rt_light;

/* Sphere: reference src/primitives.h:107-119. */
typedef struct rt_sphere {
    float center[3];
    float radius;
} rt_sphere;

/* Plane: reference src/primitives.h:31-62; (a,b,c) is the stored, already normalised normal. */
typedef struct rt_plane {
    float a, b, c, d;
} rt_plane;

/* Model: reference src/primitives.h:121-159. Faces [first_face, first_face+num_faces) of the
 * triangle arrays, in the Model's own face order; bb_* is Model::bbvol as the host computed it
 * (src/primitives.cpp:205-236) — the gate of Model::intersect (src/primitives.cpp:370) uses it as is. */
typedef struct rt_model {
    uint32_t first_face;
    uint32_t num_faces;
    float bb_center[3];
    float bb_radius;
} rt_model;

enum { RT_OBJ_SPHERE = 0, RT_OBJ_PLANE = 1, RT_OBJ_MODEL = 2 };

/* One entry of Scene::objects, in scene order (the tie-break of src/raytracer.cpp:292-300
 * depends on this order). `index` selects spheres[]/planes[]/models[] by `type`;
 * `material` indexes materials[] (a Model's hits carry the Model's material, src/raytracer.cpp:307). */
typedef struct rt_object {
    int32_t type;
    uint32_t index;
    uint32_t material;
    uint32_t reserved;
} rt_object;

/* Flattened scene, host memory, read only during rt_scene_create. Triangle data is SoA:
 * six arrays of n_faces*3 floats (Triangle::v0,v1,v2,n0,n1,n2, src/primitives.h:65). */
typedef struct rt_flat_scene {
    const rt_object* objects;     uint32_t n_objects;
    const rt_sphere* spheres;     uint32_t n_spheres;
    const rt_plane* planes;       uint32_t n_planes;
    const rt_model* models;       uint32_t n_models;
    const rt_material* materials; uint32_t n_materials;
    const rt_light* lights;       uint32_t n_lights;
    const float* v0; const float* v1; const float* v2;
    const float* n0; const float* n1; const float* n2;
    uint32_t n_faces;
} rt_flat_scene;

/* Camera state after construction/rotate/move: reference src/camera.h:14-37 (fields are private there;
 * the host surface exposes them). up/at are the normalised members, right = cross(at, up). */
typedef struct rt_camera {
    float eye[3];
    float up[3];
    float right[3];
    float at[3];
    float aspectRatio;
    float fov;            /* degrees */
    uint32_t imWidth;
    uint32_t imHeight;
} rt_camera;

#define RT_MAX_AA_POINTS 255

/* Everything the reference fixes at compile time in main()/file scope (src/raytracer.cpp:74,286,677;
 * src/camera.h:40-55; src/common.h:36). */
typedef struct rt_params {
    float background[3];      /* src/raytracer.cpp:677 */
    int32_t max_depth;        /* MAX_DEPTH, src/raytracer.cpp:74 (reference: 10) */
    uint32_t n_aa;            /* anti-aliasing pattern size; rays per pixel = 1 + n_aa (src/camera.h:72-79) */
    const float* aa_xy;       /* n_aa (x,y) pairs */
} rt_params;

/* Per-call counters. Rays are counted as SURVEY.md §8(d) defines them. */
typedef struct rt_stats {
    uint64_t trace_calls;     /* primary + reflection + refraction trace() invocations */
    uint64_t shadow_rays;     /* one per light per shaded hit                          */
    uint64_t level_rays[16];  /* trace() invocations per recursion level               */
    uint32_t passes;          /* image passes the frame was split into                 */
    uint32_t retries;         /* passes re-run with a smaller size after queue overflow */
    float    kernel_ms;       /* HIP-event time of all kernels of this call (timed calls run on one stream) */
    float    kernel_ms_by[5];        /* of which, summed over their launches: k_extend_wide, k_shade_wide,         */
    uint32_t kernel_launches_by[5];  /* k_extend_quad, k_shade_quad<0|1>, k_shade_quad<2> (closest-hit / light-loop */
    uint32_t reserved;               /* kernels, one ray per lane / per quad of lanes)                              */
} rt_stats;

typedef struct rt_scene rt_scene;   /* opaque; owns the HBM copy of the scene on one device */

/* Replaces: Scene ownership of objects/lights (reference src/scene.h:29-39) on the device side.
 * Builds the BVH on the host (one per Model) and uploads the SoA scene to `device`. */
int rt_scene_create(const rt_flat_scene* scene, int device, rt_scene** out);
/* The same with options. RT_SCENE_BVH_DEVICE: build the BVHs on the GPU (linear BVH from Morton order with its top
 * and its leaves rebuilt by surface-area heuristic: 0.2 s instead of 1.3 s for 10 M faces, frames 5-6 % slower than with the host's SAH tree; a model
 * whose tree would be too deep for the traversal stack silently gets the host builder). Results are the same with
 * either tree. No counterpart in the reference. */
#define RT_SCENE_BVH_DEVICE 1u
/* RT_SCENE_NO_PACKETS: every mesh ray is traversed by itself (one ray per quad of lanes), never as part of a 64-ray
 * packet. Same results, slower frames; with rt_set_count_work the work counters are then per RAY (the nodes and
 * triangles that ray's own traversal needs: SURVEY.md section 8d's N_node / N_tri), which is what bench.py's
 * algorithmic bytes are computed from. */
#define RT_SCENE_NO_PACKETS 2u
int rt_scene_create_ex(const rt_flat_scene* scene, int device, uint32_t flags, rt_scene** out);
void rt_scene_destroy(rt_scene* scene);

/* Replaces: render() (reference src/raytracer.cpp:422-463) for the rectangle [x0,x0+w) x [y0,y0+h).
 * bgr_out is a caller-owned HOST buffer laid out as the reference's img (BGR8, interleaved,
 * row-major over the FULL image: byte 3*(y*imWidth+x), src/raytracer.cpp:434,446-448);
 * only the rectangle's bytes are written. stats may be NULL. */
int rt_render(rt_scene* scene, const rt_camera* cam, const rt_params* params,
              uint32_t x0, uint32_t y0, uint32_t w, uint32_t h,
              uint8_t* bgr_out, rt_stats* stats);

/* rt_render with a progress report. Replaces: the per-pixel `++finPix` of render() (reference src/raytracer.cpp:450)
 * that the reference's main thread turns into its pix/s line (src/raytracer.cpp:217-225). `progress(user, done, total)`
 * is called on the calling thread after every pass of the frame (a frame of up to rt_scene_pass_samples() samples is one
 * pass: 2^26 on an MI355X, where an 8K frame at 64 samples per pixel is 32 passes) with the pixels of the rectangle finished so far, and once more with done == total
 * when the bytes are in bgr_out. progress == NULL: rt_render. */
typedef void (*rt_progress_fn)(void* user, uint64_t pixels_done, uint64_t pixels_total);
int rt_render_progress(rt_scene* scene, const rt_camera* cam, const rt_params* params,
                       uint32_t x0, uint32_t y0, uint32_t w, uint32_t h,
                       uint8_t* bgr_out, rt_stats* stats, rt_progress_fn progress, void* user);

/* Same as rt_render but the framebuffer is a DEVICE pointer (same full-image BGR8 layout) and the call
 * is asynchronous on `stream` (a hipStream_t, or NULL for the scene's own stream) unless stats != NULL.
 * Replaces: render() when the caller keeps frames on the GPU (multi-GPU gather, encoders). */
int rt_render_device(rt_scene* scene, const rt_camera* cam, const rt_params* params,
                     uint32_t x0, uint32_t y0, uint32_t w, uint32_t h,
                     uint8_t* d_bgr_out, void* stream, rt_stats* stats);

/* rt_render_device for callers that CANNOT ALLOCATE (tight serving loops, harnesses that forbid hipMalloc, code that
 * must not disturb an allocator's state). No counterpart in the reference. (Not meant to be CAPTURED into a hipGraph:
 * the call orders its launches over the scene's own streams and copies the level sizes back as a scheduling hint for
 * the next frame, which a replayed graph would freeze.) Same kernels and schedule, asynchronous on `stream` (NULL: the
 * scene's own), same bytes as rt_render_device; but the frame's node pool is not the scene's grow-only hipMalloc'ed
 * workspace, it is a fixed 20.5 MiB array that is part of the library's device image (one copy per GPU the library runs
 * on, held until the process ends), used in passes of 2^14 samples (12 pool entries per sample). The call performs no hipMalloc, hipFree or other
 * allocating call. Limits: the rectangle (rounded up to multiples of 8 in each direction) times the samples per pixel
 * must not exceed RT_NOALLOC_MAX_SAMPLES (2^21: a 1920x1080 frame at 1 sample per pixel fits), else RT_ERR_LIMIT.
 * Every call on one GPU - whatever the rt_scene - shares that GPU's array: calls that overlap on one GPU race; order
 * them on one stream or wait for the previous one. A frame whose ray trees outgrow the pool is reported by
 * rt_scene_status (RT_ERR_OVERFLOW) as for every asynchronous call: render it again, this entry point has switched to
 * passes a quarter the size for this scene (4x the room per sample, down to 64-pixel passes - the scene's ordinary
 * calls are not affected; a frame of many small passes reuses the scene's 1024 pass-counter slots in turn). A frame
 * that overflows even with 64-pixel passes is reported with a message that says so: it needs rt_render_device, whose
 * pool can grow. */
#define RT_NOALLOC_MAX_SAMPLES (1u << 21)
int rt_render_device_noalloc(rt_scene* scene, const rt_camera* cam, const rt_params* params,
                             uint32_t x0, uint32_t y0, uint32_t w, uint32_t h,
                             uint8_t* d_bgr_out, void* stream);

/* A BATCH of frames in one call: n_frames (1..32) frames of the same size, scene and params that differ only in
 * their cameras (an animation's consecutive frames, or the steps of a benchmark). Frame b is written
 * frame_stride_bytes * b behind d_bgr_out. Every launch then works on n_frames times as many rays: a small frame - or
 * one GPU's share of a frame - leaves most of the chip idle in its deep bounce levels and is bound by the host's
 * launch rate; a batch costs one frame's launches. Results are those of n_frames single calls, byte for byte.
 * Replaces: a loop of render() calls (reference src/raytracer.cpp:422-463). */
int rt_render_batch_device(rt_scene* scene, const rt_camera* cams, uint32_t n_frames, const rt_params* params,
                           uint32_t x0, uint32_t y0, uint32_t w, uint32_t h,
                           uint8_t* d_bgr_out, uint64_t frame_stride_bytes, void* stream, rt_stats* stats);
/* The tile form (see rt_render_tiles_device): frame b's tile-major buffer starts frame_stride_bytes * b behind
 * d_tiles_out. */
int rt_render_tiles_batch_device(rt_scene* scene, const rt_camera* cams, uint32_t n_frames, const rt_params* params,
                                 uint32_t tile_w, uint32_t tile_h, uint32_t part, uint32_t n_parts,
                                 uint8_t* d_tiles_out, uint64_t frame_stride_bytes, void* stream, rt_stats* stats);

/* Replaces: a loop of render() calls whose `img` is a HOST buffer (reference src/raytracer.cpp:434,446-448,666-667),
 * without blocking: the frames (n_frames cameras, as rt_render_batch_device) are rendered into a device buffer the scene
 * owns and copied into host_bgr_out (frame b at + b * frame_stride_bytes, the reference's full-image BGR8 layout, only
 * the rectangle written) by a copy on a stream of the scene's own, ordered behind the rendering on `stream` (NULL: the scene's own). The call
 * returns at once; the bytes are valid after rt_scene_status(scene) - NOT after a wait on `stream` alone. With PAGE-LOCKED host memory
 * (rt_pinned_alloc, hipHostMalloc, hipHostRegister) the copy overlaps the kernels of the next call on another scene:
 * two scenes alternating deliver host frames at nearly the device-framebuffer rate. Pageable memory works, slower. */
int rt_render_batch_host_async(rt_scene* scene, const rt_camera* cams, uint32_t n_frames, const rt_params* params,
                               uint32_t x0, uint32_t y0, uint32_t w, uint32_t h,
                               uint8_t* host_bgr_out, uint64_t frame_stride_bytes, void* stream);
/* Page-locked host memory for framebuffers (hipHostMalloc / hipHostFree). No counterpart in the reference (its img is
 * a plain array, src/raytracer.cpp:666). */
int rt_pinned_alloc(size_t bytes, void** out);
void rt_pinned_free(void* p);

/* Waits for the scene's outstanding asynchronous frames and says whether they are good: RT_OK, or RT_ERR_OVERFLOW
 * when one of them ran out of the pool that holds the child rays (a scene whose ray trees branch more than the
 * workspace allows for; the synchronous calls re-run such passes in smaller pieces by themselves, an asynchronous
 * call cannot). The pixels of that frame are then not valid; the scene has already switched to smaller passes, so
 * the caller renders the frame again. No counterpart in the reference (its recursion needs no pool). */
int rt_scene_status(rt_scene* scene);

/* The scene's own stream (a hipStream_t): the one asynchronous calls use when `stream` is NULL. Callers that have
 * to order other work behind a frame (a collective, a copy) enqueue it there or wait for an event recorded there.
 * No counterpart in the reference (its render() is synchronous). */
void* rt_scene_stream(const rt_scene* scene);

/* Multi-GPU tile ownership. Replaces: the tile decomposition of thread_render()
 * (reference src/raytracer.cpp:577-627: fixed tiles handed to workers). The full image is cut into
 * tile_w x tile_h tiles, numbered row-major; this call renders the tiles with
 * (tile_index % n_parts == part) into d_tiles_out, a DEVICE buffer in tile-major order
 * (k-th owned tile at byte k*tile_w*tile_h*3, pixels row-major inside the tile, BGR8; pixels of
 * edge tiles outside the image are written as 0). rt_tiles_owned() gives k's range. */
int rt_render_tiles_device(rt_scene* scene, const rt_camera* cam, const rt_params* params,
                           uint32_t tile_w, uint32_t tile_h, uint32_t part, uint32_t n_parts,
                           uint8_t* d_tiles_out, void* stream, rt_stats* stats);
uint32_t rt_tiles_owned(uint32_t imWidth, uint32_t imHeight, uint32_t tile_w, uint32_t tile_h,
                        uint32_t part, uint32_t n_parts);

/* Scatter gathered tile-major buffers back into the reference's row-major BGR8 frame. Part q's tiles start at
 * tile slot q*part_stride_tiles of d_tiles_all (part_stride_tiles >= every part's rt_tiles_owned(); pass 0 for parts
 * packed back to back). d_tiles_all and d_bgr_out are device pointers.
 * Replaces: the shared `img` all render() workers write into (reference src/raytracer.cpp:434). */
int rt_untile_device(int device, uint32_t imWidth, uint32_t imHeight, uint32_t tile_w, uint32_t tile_h,
                     uint32_t n_parts, uint32_t part_stride_tiles, const uint8_t* d_tiles_all, uint8_t* d_bgr_out,
                     void* stream);
/* The same for a batch (rt_render_tiles_batch_device): frame k's tiles of every part start in_frame_stride_bytes * k
 * behind those of frame 0, frame k is written out_frame_stride_bytes * k behind d_bgr_out; one launch. */
int rt_untile_batch_device(int device, uint32_t imWidth, uint32_t imHeight, uint32_t tile_w, uint32_t tile_h,
                           uint32_t n_parts, uint32_t part_stride_tiles, uint32_t n_frames,
                           uint64_t in_frame_stride_bytes, uint64_t out_frame_stride_bytes,
                           const uint8_t* d_tiles_all, uint8_t* d_bgr_out, void* stream);

/* One frame on several GPUs of ONE process. Replaces: the worker pool of thread_render() (reference
 * src/raytracer.cpp:577-627: fixed tiles handed to workers that write into one shared img) for a C/C++ caller that has
 * more than one GPU and no launcher. The scene is replicated (one rt_scene per entry of `devices`; a device may be
 * listed more than once), 32x32-pixel tiles are dealt round-robin (rt_render_tiles_device), the parts' tile buffers
 * are gathered on devices[0] by peer copies (hipMemcpyPeerAsync: direct xGMI transfers, each peer over its own link)
 * and scattered into the row-major frame there (rt_untile_device). The frame equals the one-GPU frame byte for byte.
 * `flags` as rt_scene_create_ex. (A one-process-per-GPU job uses the same entry points with a collective as the
 * exchange: bench.py, INTEGRATION.md.) */
#define RT_MULTI_MAX_DEVICES 64
typedef struct rt_multi rt_multi;
int rt_multi_create(const rt_flat_scene* scene, const int* devices, uint32_t n_devices, uint32_t flags, rt_multi** out);
void rt_multi_destroy(rt_multi* multi);
uint32_t rt_multi_parts(const rt_multi* multi);                    /* n_devices */
rt_scene* rt_multi_scene(const rt_multi* multi, uint32_t part);    /* part's replica (its stream, status, BVH info) */
/* render() of the WHOLE image (reference src/raytracer.cpp:422-463 with x_start = y_start = 0) into the caller's host
 * buffer, the reference's BGR8 layout. Synchronous. stats (may be NULL): the parts' counters summed; with stats the
 * parts run one after the other. A frame whose ray trees outgrow a replica's pool is rendered again with smaller passes. */
int rt_multi_render(rt_multi* multi, const rt_camera* cam, const rt_params* params, uint8_t* bgr_out, rt_stats* stats);
/* The same into a DEVICE buffer on devices[0], asynchronous (unless stats != NULL): the frame is complete when
 * rt_multi_status returns, or for work ordered behind the stream of rt_multi_scene(multi, 0). */
int rt_multi_render_device(rt_multi* multi, const rt_camera* cam, const rt_params* params, uint8_t* d_bgr_out, rt_stats* stats);
/* Waits for all parts; RT_OK or the first failing part's rt_scene_status. */
int rt_multi_status(rt_multi* multi);

/* Unit entry points used by the parity tests (each runs the same device functions the renderer uses). */

/* Replaces: Camera::castRays (reference src/camera.h:72-79) for n pixels; rays_out gets
 * n*(1+n_aa)*6 floats (origin xyz, direction xyz), host memory. */
int rt_cast_rays(int device, const rt_camera* cam, const rt_params* params,
                 const uint32_t* xy, uint32_t n_pixels, float* rays_out);

/* Replaces: Primitive::intersect over scene objects (reference src/primitives.h:23) for n rays given
 * as 6 floats each. object < 0: closest hit over the whole scene as trace() does
 * (src/raytracer.cpp:292-300), hit_obj_out gets the object index or -1. object >= 0: that object only.
 * dist_out/normal_out as intersect() leaves them on a hit (unspecified on a miss). */
int rt_intersect(rt_scene* scene, int object, const float* rays, uint32_t n_rays,
                 int32_t* hit_obj_out, float* dist_out, float* normal_out);

/* Replaces: the shadow test of trace() (reference src/raytracer.cpp:318-325): occluded_out[i] = 1 if any
 * object is hit with tmpdist < max_dist[i]. */
int rt_occluded(rt_scene* scene, const float* rays, const float* max_dist, uint32_t n_rays,
                uint8_t* occluded_out);

/* Replaces: trace() (reference src/raytracer.cpp:279-420) for n rays at depth 0; rgb_out n*3 floats. */
int rt_trace(rt_scene* scene, const rt_params* params, const float* rays, uint32_t n_rays,
             float* rgb_out, rt_stats* stats);

/* The same three with the mesh traversal chosen by the caller. The renderer takes a ray through a Model
 * (reference src/primitives.cpp:364-384: every face, lowest index wins ties) in one of three ways, picked per level
 * by ray coherence and level size; all three must give the reference's result bit for bit, so the parity tests run
 * the same rays through each:
 *   RT_TRAVERSE_QUAD    one ray per quad of lanes (what rt_intersect / rt_occluded / rt_trace use)
 *   RT_TRAVERSE_PACKET  one ray per lane, the wave's 64 rays walk the tree together; rays are taken 64 at a time in
 *                       the order given (primary rays and their shadow rays in rt_render)
 *   RT_TRAVERSE_SERVE   one ray per lane, the lanes that reach a mesh are served by the wave's own quads
 *                       (bounce levels of large passes in rt_render)
 * rt_trace_form applies the form to every recursion level and to the shadow rays. */
enum { RT_TRAVERSE_QUAD = 0, RT_TRAVERSE_PACKET = 1, RT_TRAVERSE_SERVE = 2 };
int rt_intersect_form(rt_scene* scene, int form, int object, const float* rays, uint32_t n_rays,
                      int32_t* hit_obj_out, float* dist_out, float* normal_out);
int rt_occluded_form(rt_scene* scene, int form, const float* rays, const float* max_dist, uint32_t n_rays,
                     uint8_t* occluded_out);
int rt_trace_form(rt_scene* scene, int form, const rt_params* params, const float* rays, uint32_t n_rays,
                  float* rgb_out, rt_stats* stats);

/* Level-0 samples per pass: a frame with more samples is rendered in passes of whole pixels that reuse one
 * workspace (`samples` a power of two in [2^12, 2^26], 0 = the default: the largest power of two up to 2^26 whose
 * workspace - 6 x 112 B per sample - stays below a fifth of the device's memory, at least 2^23; 2^26 on an MI355X. Larger
 * passes render faster; a scene that finds no room for its workspace falls back to smaller ones by itself). Waits for
 * the scene's frames in flight. Results never depend on it. No counterpart in the reference (its recursion needs no
 * workspace); for callers that bound the scene's HBM use and for the multi-pass parity tests. */
int rt_scene_set_pass_samples(rt_scene* scene, uint32_t samples);
/* Fixes one choice of the launch schedule that the library otherwise makes from the scene, the samples per pixel and
 * the level sizes of the previous frame. Results never depend on any of them (tests/test_gpu_parity.py renders one scene
 * through the combinations and compares the bytes); they exist for those tests and for timing experiments. Waits for the
 * scene's frames in flight. No counterpart in the reference. Names and values:
 *   "packet"           -1 automatic | 0 level 0 by marks + quads | 1 level 0 as 64-ray packets | 2 packets at every large level
 *   "large_threshold"  rays from which a bounce level runs as per-lane kernel + queue (-1: the built-in table)
 *   "large_max_level"  deepest level that may do so (-1: built-in)
 *   "serve"            bounce levels as per-lane kernels with quad service: 0 never | 1 always | -1 by pass size
 *   "split_small"      runs of small levels shaded per lane + resuming quads (1, default) | one quad launch (0)
 *   "fuse"             a scene without meshes runs each level as one launch, closest hits and light loops (1, default) | two (0)
 *   "tiny_run"         a run of at most this many hits is shaded by one quad launch (default 8192)
 *   "run_hits"         a run of small levels that holds at least this many hits is shaded at once instead of when three levels have
 *                      come together (default 786,432; 0 = never)
 *   "tail_rays"        the deepest levels that together hold at most this many rays are traced by ONE launch, depth first
 *                      (k_tail), instead of a launch per level (0 = never)
 *   "index64"          1: sample-index arithmetic by 64-bit division even where 32 bits would do
 *   "chunk_log2"       level-0 blocks per shard chunk
 *   "graph"            1: asynchronous frames are captured into a hipGraph and replayed (measured slower)
 *   "order0"           1 (default): the level-0 launches of a one-pass frame of 2,048 to 65,536 64-sample groups hand the groups
 *                      out in descending order of what they cost two frames earlier (two sets of arrays in turn), the dearest in parts
 *                      (k_order_groups); 0: pool order
 *   "shade_serve_level" with packets below level 0: the light loops of large levels >= this one by quad service instead of
 *                      packets (0, default: packets; measured slower)
 *   "sort_shade"       light loops that run as packets over a bounce level take that level's hits sorted by the cell of
 *                      their hit point (Morton key + radix sort) from this many rays on (0, default: never; measured slower)
 *   "sort_bits"        bits per axis of that key (1..10, default 8)
 * The environment variables of the same meaning (RT_PACKET, RT_LARGE_THR, ... tools/exp_*.sh) are read once, by
 * rt_scene_create; no rendering call reads the environment. Unknown name or value out of range: RT_ERR_INVALID. */
int rt_scene_set_option(rt_scene* scene, const char* name, int64_t value);
uint32_t rt_scene_pass_samples(const rt_scene* scene);      /* the current value */

/* BVH introspection for tests and the roofline's algorithmic bytes/flops. */
typedef struct rt_bvh_info {
    uint32_t n_nodes;
    uint32_t n_faces;
    uint32_t max_depth;
    uint32_t node_bytes;
    uint32_t tri_bytes;
    uint32_t built_on_device; /* 1: RT_SCENE_BVH_DEVICE took effect */
    uint64_t device_bytes;   /* total HBM held by the scene */
    double build_seconds;    /* BVH build (device builder: vertex upload included) */
} rt_bvh_info;
int rt_scene_bvh_info(const rt_scene* scene, rt_bvh_info* out);

/* Builds the scene's BVHs on the host only (no GPU is touched) and walks them: every face of a model in exactly one
 * leaf block, leaf data equal to the v0 / v1-v0 / v2-v0 floats Triangle::intersect forms (reference
 * src/primitives.cpp:52-53), every child box containing the vertices below it, children packed to the front, stack
 * bound within the traversal's. `errors` counts violations (0 = sound). No counterpart in the reference (it has no
 * acceleration structure); used by the CPU tests and to time the builder. */
typedef struct rt_bvh_check_result {
    uint32_t n_nodes, n_leaves, n_faces, max_depth, max_stack, reserved;
    uint64_t errors;
    uint64_t host_bytes;     /* nodes + leaf blocks */
    double build_seconds;
} rt_bvh_check_result;
int rt_bvh_check(const rt_flat_scene* scene, rt_bvh_check_result* out);
/* The same walk over the tree a scene actually holds in HBM (copied back), whichever builder made it; `flat` must be
 * the scene it was created from. build_seconds is the scene's. */
int rt_scene_bvh_check(const rt_scene* scene, const rt_flat_scene* flat, rt_bvh_check_result* out);

/* Traversal work of the last call that passed a non-NULL stats AND had counting enabled with
 * rt_set_count_work(1): node visits / triangle tests / sphere / plane tests summed over all rays, per kernel:
 * rt_get_work(kernel = 0: k_extend_wide, 1: k_shade_wide, 2: k_extend_quad, 3: k_shade_quad<0|1>, 4: k_shade_quad<2>
 * (the order of rt_stats.kernel_ms_by), -1: all). */
typedef struct rt_work {
    uint64_t nodes_visited;
    uint64_t tris_tested;
    uint64_t spheres_tested;
    uint64_t planes_tested;
    uint64_t wave_node_steps;   /* times a 64-lane wave executed the node step (any lane active) */
    uint64_t wave_tri_steps;    /* times a wave executed the triangle test                        */
    uint64_t trav_ticks_sum;    /* 10-ns ticks waves spent inside mesh traversal, summed over waves */
    uint64_t trav_ticks_max;    /* ... the largest single wave                                     */
    uint64_t batches;           /* 16-ray traversal batches                                        */
    uint64_t trav_waves;        /* waves that ran at least one batch                               */
    uint64_t wave_ticks_sum;    /* 10-ns ticks between a wave's start and its end, summed          */
    uint64_t wave_ticks_max;    /* ... the longest-lived wave                                      */
    uint64_t wave_cycles_sum;   /* shader cycles (s_memtime) between a wave's start and its end, summed: the clock the launch
                                   ran at = wave_cycles_sum / wave_ticks_sum x 100 MHz                */
} rt_work;
int rt_set_count_work(rt_scene* scene, int enable);
int rt_get_work(const rt_scene* scene, int kernel, rt_work* out);

const char* rt_last_error(void);   /* thread-local message of the last failing call */
const char* rt_version(void);

#ifdef __cplusplus
}
#endif
#endif /* RAYTRACER_AMD_H */
