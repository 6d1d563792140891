// rt_kernels.hip — hand-written HIP kernels (gfx950) of the render hot path.
//
// What runs here, and the reference code each part answers to:
//   cam_ray / cast_ray        Camera::getCamRay / castRays            reference src/camera.h:57-79
//   sphere_isect              Sphere::intersect + getNormal           reference src/primitives.cpp:173-203
//   plane_isect               Plane::intersect                        reference src/primitives.cpp:11-25
//   tri_isect                 Triangle::intersect (Moeller-Trumbore)  reference src/primitives.cpp:42-81,125
//   closest_loop / occluded_loop   the two object loops of trace()      reference src/raytracer.cpp:292-300, 319-325
//   quad_traverse             the face loop of Model::intersect        reference src/primitives.cpp:364-384
//   k_extend_wide / _quad     trace(): closest hit, Fresnel, child rays  reference src/raytracer.cpp:292-308, 347-396
//   k_shade_wide / _quad      trace(): light loop (shadow test, Phong)   reference src/raytracer.cpp:311-345
//   finish_color / k_fold     trace(): ambient + children + clamp     reference src/raytracer.cpp:399-406
//   k_resolve                 render(): sample average + BGR8 quantisation  reference src/raytracer.cpp:436-448
//
// Structure (not the reference's): the recursion becomes level-synchronous and every level is two independent stages,
// extend and shade (see the comment above k_extend_wide). Child rays are appended to level k+1 with one atomic per
// wave (ballot + prefix count). Because the reference clamps the colour at EVERY recursion level (raytracer.cpp:406)
// the tree is folded bottom-up afterwards, level by level, with exactly the reference's expression per node; a
// throughput-product accumulator would be wrong.
//
// Arithmetic: IEEE binary32 with the reference's operation order, compiled with -ffp-contract=off (no FMA
// contraction; the x86 reference build has none), correctly rounded f32 divide/sqrt (hipcc default), and double
// exactly where the reference's overload resolution computes in double (sphere roots, pow, camera tan product).
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <algorithm>
#include <float.h>
#include <stdint.h>

#include "rt_device.h"
#include "rt_launch.h"

#define RT_EPS 0.000001f   // reference src/common.h:36
// DScene::debug / LevelArgs::debug switch parts of the kernels off and the wave logs time single waves: timing
// experiments, built only with -DRT_EXPERIMENTS (make EXTRA=-DRT_EXPERIMENTS). The release build has neither the tests
// (scalar instructions in the loops whose scalar instructions were counted one by one) nor a way to reach them.
#ifdef RT_EXPERIMENTS
#define RT_DBG(x) (x)
#define RT_LOG_PTR(p) (p)
#else
#define RT_DBG(x) 0u
#define RT_LOG_PTR(p) ((uint4*)nullptr)
#endif
#define RT_WIDE_BLOCK_MAX 256
#ifndef RT_WIDE_BLOCK_DEFAULT
#define RT_WIDE_BLOCK_DEFAULT 64
#endif
#include <cstdlib>
#include <cstring>

struct V3 { float x, y, z; };
__device__ __forceinline__ V3 mk(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 operator*(V3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ V3 neg(V3 a) { return mk(-a.x, -a.y, -a.z); }
// reference src/container.h:134-138
__device__ __forceinline__ float dot(V3 a, V3 b) { return (a.x * b.x) + (a.y * b.y) + (a.z * b.z); }
// reference src/container.h:151-158
__device__ __forceinline__ V3 cross(V3 u, V3 v) { return mk(u.y * v.z - u.z * v.y, u.z * v.x - u.x * v.z, u.x * v.y - u.y * v.x); }
// reference src/container.h:140-142: double sqrt of the float dot, narrowed == correctly rounded sqrtf
__device__ __forceinline__ float length(V3 a) { return sqrtf(dot(a, a)); }
// reference src/container.h:144-146
__device__ __forceinline__ V3 normalize(V3 a) { return a * (1.0f / length(a)); }
__device__ __forceinline__ bool isnan3(V3 a) { return isnan(a.x) || isnan(a.y) || isnan(a.z); }
__device__ __forceinline__ float max_std(float a, float b) { return (a < b) ? b : a; }   // std::max
__device__ __forceinline__ float min_std(float a, float b) { return (b < a) ? b : a; }   // std::min
__device__ __forceinline__ V3 ld3(const float* p) { return mk(p[0], p[1], p[2]); }
__device__ __forceinline__ V3 f4xyz(float4 v) { return mk(v.x, v.y, v.z); }

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
struct Ray { V3 o, d; };

// The per-frame pool (child rays, node and hit records) is written once and read once or twice, gigabytes of it per
// frame at 4K / 8K; the BVH's nodes and leaf blocks are what every ray comes back to. Pool accesses therefore carry
// the non-temporal hint (`nt`: the line is a streaming one for the L2's replacement), so that they do not push the
// tree out of the 4-MB L2s.
#ifndef RT_STREAM_POOL
#define RT_STREAM_POOL 1      // (0: A/B builds)
#endif
typedef float f32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 pool_ld(const float4* p) {
    if (!RT_STREAM_POOL) return *p;
    const f32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ uint4 pool_ld(const uint4* p) {
    if (!RT_STREAM_POOL) return *p;
    const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void pool_st(float4* p, float4 v) {
    if (!RT_STREAM_POOL) { *p = v; return; }
    f32x4_t w; w.x = v.x; w.y = v.y; w.z = v.z; w.w = v.w;
    __builtin_nontemporal_store(w, reinterpret_cast<f32x4_t*>(p));
}
__device__ __forceinline__ void pool_st(uint4* p, uint4 v) {
    if (!RT_STREAM_POOL) { *p = v; return; }
    u32x4 w; w.x = v.x; w.y = v.y; w.z = v.z; w.w = v.w;
    __builtin_nontemporal_store(w, reinterpret_cast<u32x4*>(p));
}

// Wave-uniform, read-only scene tables (object table, models, lights) are read through the constant address space:
// the compiler may then use scalar loads (s_load_dwordx4/x8 into SGPRs) although the kernel also stores to global
// memory; through a plain pointer it falls back to per-lane vector loads whose latency sits in every object test.
#define RT_CONSTANT __attribute__((address_space(4)))
// First statement of a block guarded by a wave-uniform test: keeps that test a scalar branch of its own (s_cmp +
// s_cbranch). Without it the compiler folds it into the per-lane condition inside the block - compare, select, and
// with the lane mask, save exec, branch: five scalar instructions, paid by every object type for every object.
#define RT_UNIFORM_BRANCH asm volatile("")
__device__ __forceinline__ u32x4 ldc4(const void* base, uint32_t i16) {
    return *((const RT_CONSTANT u32x4*)(base) + i16);
}
__device__ __forceinline__ float4 as_f4(u32x4 v) {
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
struct ModelRef { uint32_t root; float4 bb; };   // what traversal needs of a DModel
__device__ __forceinline__ ModelRef load_model(const DModel* models, uint32_t m) {
    const u32x4 a = ldc4(models, 2 * m), b = ldc4(models, 2 * m + 1);   // {root, first, n, cx} {cy, cz, r, pad}
    ModelRef r;
    r.root = a.x;
    r.bb = make_float4(__uint_as_float(a.w), __uint_as_float(b.x), __uint_as_float(b.y), __uint_as_float(b.z));
    return r;
}
__device__ __forceinline__ V3 ray_point(const Ray& r, float t) { return r.o + r.d * t; }   // container.h:255

// ---------------------------------------------------------------------------------------------------------
// Camera (reference src/camera.h:57-79). tan_half = tan(deg2rad(fov)*0.5) in double, formed once on the host.
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ Ray cam_ray(const DCamera& c, float x, float y) {
    const float px_ndc = x / (float)c.imWidth;
    const float py_ndc = y / (float)c.imHeight;
    const float cam_x = (float)((double)((2 * px_ndc - 1) * c.aspect) * c.tan_half);
    const float cam_y = (float)((double)(1 - 2 * py_ndc) * c.tan_half);
    const V3 dir = (mk(c.right[0], c.right[1], c.right[2]) * cam_x + mk(c.up[0], c.up[1], c.up[2]) * cam_y) +
                   mk(c.at[0], c.at[1], c.at[2]);
    Ray r;
    r.o = mk(c.eye[0], c.eye[1], c.eye[2]);
    r.d = normalize(dir);
    return r;
}
__device__ __forceinline__ Ray cast_ray(const DCamera& c, const float* __restrict__ aa, uint32_t x, uint32_t y, uint32_t k) {
    if (k == 0) return cam_ray(c, (float)x, (float)y);
    return cam_ray(c, (float)x + aa[2 * (k - 1)], (float)y + aa[2 * (k - 1) + 1]);
}

// ---------------------------------------------------------------------------------------------------------
// Analytic primitives
// ---------------------------------------------------------------------------------------------------------
// reference src/primitives.cpp:173-203. `dist` is in/out: left untouched when neither branch applies.
// inv_r = (float)(1.0 / (double)radius), the reciprocal Vec3::operator/= forms (container.h:72-75), taken once on the host.
#ifndef RT_SPHERE_EARLY
#define RT_SPHERE_EARLY 1      // (0: A/B builds, tools/exp_ab_builds.sh)
#endif
__device__ __forceinline__ bool sphere_isect(float4 s, float inv_r, const Ray& ray, float& dist, V3& normal, bool coherent = false) {
    const V3 c = f4xyz(s);
    const V3 temp = ray.o - c;
    const float val1 = dot(temp, ray.d);
    // Most rays miss most spheres, and the correctly rounded sqrtf of length() is 16 of the 36 instructions up to the
    // `val3 < 0` exit: when every active lane of the wave misses for certain, leave before it. With D = dot(temp, temp),
    // A = val1^2, R = r^2 (all as the reference rounds them): val2^2 = D (1 + e), |e| < 1.8e-7, and each of the two
    // additions is off by at most 2^-24 of its result, so |val3 - ((A - D) + R)| < 4.2e-7 (A + D + R): a lane with
    // (A - D) + R < -1e-6 (A + D + R) has val3 < 0. (-1e-36: denormal operands; inf or NaN anywhere compares false = no
    // early exit.) The exit leaves `dist` untouched, as `val3 < 0` does.
    // `coherent` (wave-uniform): only where the wave's rays are neighbours - camera rays and the shadow rays of their hits.
    // Among 64 unrelated bounce rays one nearly always comes close, and the test only costs (measured: 10 M-triangle
    // field at 1080p 4.53 -> 4.63 ms with it at every level; analytic 1080p 4 spp 0.420 -> 0.409 at level 0).
    const float d2 = dot(temp, temp), a2 = val1 * val1, r2 = s.w * s.w;
    if (RT_SPHERE_EARLY && coherent) {
        const bool maybe = !((a2 - d2) + r2 < __builtin_fmaf((a2 + d2) + r2, -1e-6f, -1e-36f));
        if (__builtin_amdgcn_ballot_w64(maybe) == 0) return false;                  // wave-uniform
    }
    const float val2 = sqrtf(d2);                                                    // = length(temp)
    const float val3 = a2 - val2 * val2 + r2;
    if (val3 < 0) return false;
    const double sq = sqrt((double)val3);           // the reference's sqrt(val3) is the double function
    const float dist1 = (float)((double)(-val1) + sq);
    const float dist2 = (float)((double)(-val1) - sq);
    if (dist1 < 0 || dist2 < 0) dist = max_std(dist1, dist2);
    else if (dist1 > 0 && dist2 > 0) dist = min_std(dist1, dist2);
    const V3 n = ray_point(ray, dist) - c;
    normal = n * inv_r;
    return dist > RT_EPS;
}
// The gate of Model::intersect only needs the boolean.
__device__ __forceinline__ bool sphere_gate(float4 s, const Ray& ray, bool coherent = false) {
    float d = 0;   // uninitialised in the reference (primitives.cpp:366); defined as 0
    V3 n;
    return sphere_isect(s, 0.0f, ray, d, n, coherent);
}
// reference src/primitives.cpp:11-25
__device__ __forceinline__ bool plane_isect(float4 p, const Ray& ray, float& dist, V3& normal) {
    normal = f4xyz(p);
    const float vd = dot(normal, ray.d);
    const float vo = dot(normal, ray.o) + p.w;
    dist = -vo / vd;
    return dist > RT_EPS;
}

// reference src/primitives.cpp:52-78,125 with e1 = v1-v0, e2 = v2-v0 formed on the host.
__device__ __forceinline__ bool tri_isect(V3 v0, V3 e1, V3 e2, const Ray& ray, float& t, float& u, float& v) {
    const V3 pVec = cross(ray.d, e2);
    const float det = dot(pVec, e1);
    if (fabsf(det) < RT_EPS) return false;
    const float invDet = 1 / det;
    const V3 tVec = ray.o - v0;
    u = invDet * dot(pVec, tVec);
    if (u < 0 || u > 1) return false;
    const V3 qVec = cross(tVec, e1);
    v = invDet * dot(qVec, ray.d);
    if (v < 0 || u + v > 1) return false;
    t = invDet * dot(qVec, e2);
    return t > RT_EPS;
}

// The same test for a PACKET (one ray per lane, the triangle wave-uniform): the same operations in the same order, with
// the reference's two rejections (primitives.cpp:63-64,68-69 and :73-74) as WAVE-UNIFORM exits. Written as tri_isect the
// compiler flattens the whole test into one block of 63 vector instructions that every visited triangle pays in full;
// but a packet reaches a leaf when ANY of its rays hits the leaf's box, and most triangles of such a leaf are missed by
// every ray of the wave: when no lane is left after the `u` test the wave leaves after 36 instructions, after the `v`
// test after 53. `m` (in: the lanes that take part; out: the lanes whose ray hits, t / u / v valid there) is a lane mask
// held as a scalar: the ballot of each comparison ANDed on the scalar unit (the ballot of a compound per-lane condition
// goes through a VGPR and back). Same answer per lane as tri_isect: a lane that failed a rejection keeps computing
// values that its cleared bit in `m` discards.
// EXITS: bit 0 = leave after the `u` test, bit 1 = after the `v` test, when no lane is left.
#ifndef RT_TRI_EXITS_ANY
#define RT_TRI_EXITS_ANY 3         // shadow packets (A/B builds: 0..3)
#endif
#ifndef RT_TRI_MASKS_ANY
#define RT_TRI_MASKS_ANY 1         // shadow packets: the leaf test on scalar lane masks with wave-uniform exits (0: the flat per-lane form)
#endif
#ifndef RT_TRI_MASKS_CLOSEST
#define RT_TRI_MASKS_CLOSEST 0
#endif
#ifndef RT_TRI_PIN_LOADS
#define RT_TRI_PIN_LOADS 1
#endif
#ifndef RT_TRI_EXITS_CLOSEST
#define RT_TRI_EXITS_CLOSEST 0     // closest-hit packets: the rays of an 8x8-pixel block mostly DO hit the triangles of the leaves they reach
#endif
template <int EXITS>
__device__ __forceinline__ void tri_isect_packet(V3 v0, V3 e1, V3 e2, const Ray& ray, unsigned long long& m, float& t, float& u, float& v) {
    const V3 pVec = cross(ray.d, e2);
    const float det = dot(pVec, e1);
    const float invDet = 1 / det;
    const V3 tVec = ray.o - v0;
    u = invDet * dot(pVec, tVec);
    m &= __builtin_amdgcn_ballot_w64(!(fabsf(det) < RT_EPS)) & __builtin_amdgcn_ballot_w64(!(u < 0)) & __builtin_amdgcn_ballot_w64(!(u > 1));
    if ((EXITS & 1) && m == 0) return;      // wave-uniform
    const V3 qVec = cross(tVec, e1);
    v = invDet * dot(qVec, ray.d);
    m &= __builtin_amdgcn_ballot_w64(!(v < 0)) & __builtin_amdgcn_ballot_w64(!(u + v > 1));
    if ((EXITS & 2) && m == 0) return;      // wave-uniform
    t = invDet * dot(qVec, e2);
    m &= __builtin_amdgcn_ballot_w64(t > RT_EPS);
}

// ---------------------------------------------------------------------------------------------------------
// BVH traversal: one ray per QUAD of lanes (16 rays per wave).
//
// This workload is bound by the length of dependent chains, not by arithmetic or bandwidth: a frame holds only a few
// machine-fulls of rays, and a wave that walks 64 divergent rays through a binary tree executes the UNION of their
// paths, one dependent fetch after the other, while most of the chip idles. So a mesh ray gets four lanes: lane j of
// the quad fetches and tests child j of a 4-wide node (or triangle j of a leaf), the quad orders the hit children
// with DPP moves and keeps ONE stack (LDS, [entry][quad]: 16 quads -> 16 banks, the four lanes of a quad broadcast).
// A wave's traversal then lasts as long as its LONGEST ray, not the union of 64, with ~4x fewer instructions per step
// on the critical path and half the tree depth.
//
// The 64 lanes of the wave are "clients" (one ray each, as everywhere else in the kernels). A mesh object is reached
// with wave-uniform control flow; the clients that pass the bounding-sphere gate are served 16 at a time: client of
// rank r goes to quad r%16 in batch r/16, its ray is pulled into the quad's lanes with ds_bpermute and the result is
// pulled back the same way. No LDS exchange buffers, no barriers.
// ---------------------------------------------------------------------------------------------------------
struct Work {
    uint32_t nodes, tris, spheres, planes, wave_nodes, wave_tris;
    uint32_t batches;                 // quad-traversal batches this wave ran (COUNT builds)
    unsigned long long trav_ticks;    // wall-clock ticks (10 ns) this wave spent inside them
    unsigned long long t_start;       // wave start
    unsigned long long c_start;       // wave start, core clock
};

__device__ __forceinline__ uint32_t lane_id();
// 1 on exactly one active lane of the wave: counts how often the WAVE executes a loop body (COUNT builds)
__device__ __forceinline__ uint32_t wave_once() { return lane_id() == (uint32_t)(__ffsll((long long)__ballot(1)) - 1) ? 1u : 0u; }

// Ray/box slab test for culling only (never decides a hit): t = lo*inv - o*inv as one FMA per plane. Its error
// (<= 2^-24 |o| in space) is far inside the 2e-5 relative padding of the boxes. fminf/fmaxf return the non-NaN
// operand, so an axis whose products are inf-inf / 0*inf (zero direction component) simply does not cull.
__device__ __forceinline__ float slab(float lox, float loy, float loz, float hix, float hiy, float hiz, V3 inv, V3 ood,
                                      float tmax, bool& hit) {
    const float tx0 = __builtin_fmaf(lox, inv.x, -ood.x), tx1 = __builtin_fmaf(hix, inv.x, -ood.x);
    const float ty0 = __builtin_fmaf(loy, inv.y, -ood.y), ty1 = __builtin_fmaf(hiy, inv.y, -ood.y);
    const float tz0 = __builtin_fmaf(loz, inv.z, -ood.z), tz1 = __builtin_fmaf(hiz, inv.z, -ood.z);
    const float tn = fmaxf(fmaxf(fmaxf(fminf(tx0, tx1), fminf(ty0, ty1)), fminf(tz0, tz1)), 0.0f);
    float tf = fminf(fminf(fminf(fmaxf(tx0, tx1), fmaxf(ty0, ty1)), fmaxf(tz0, tz1)), tmax);
    tf *= 1.000001f;    // conservative far bound (robust slab test): 8 ulp, see safe_rcp
    hit = tn <= tf;
    return tn;
}

// (v_rcp_f32, 1 ulp, instead of the 11-instruction IEEE division: the reciprocal only scales an axis' plane distances
// by (1 + e), |e| <= 2^-23, the same for both planes of the axis; slab()'s far bound is widened by 8 ulp, which covers
// one ulp here and the half ulp of each FMA on either side of the comparison several times over. |a| >= 1e-30: no denormal in or out.)
__device__ __forceinline__ float safe_rcp(float d) {
    const float a = fabsf(d) < 1e-30f ? copysignf(1e-30f, d) : d;   // NaN stays NaN: its axis then never culls
    return __builtin_amdgcn_rcpf(a);
}

// shadow rays (ANY) visit the children of a node in node order: 0 = nearest first like the closest-hit rays (A/B builds)
#ifndef RT_CLOSEST_ORDER
#define RT_CLOSEST_ORDER 0      // packets, closest hit: 0 children sorted, 1 nearest first and the rest as they come, 2 node order (A/B builds;
                                // measured, level-0 extend launch: bunny_1080p 0.125 / 0.131 / 0.158 ms, 10 M-triangle field 0.95 / 1.02 / 1.32)
#endif
#ifndef RT_ANY_UNORDERED
#define RT_ANY_UNORDERED 1
#endif
#define RT_DONE 0xFFFFFFFFu   // stack sentinel; has the leaf bit set, so the descent loop stops on it too

// quad-local cross-lane moves (DPP quad_perm: no LDS, one VALU instruction each)
template <int CTRL> __device__ __forceinline__ uint32_t dpp_u(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
}
template <int CTRL> __device__ __forceinline__ float dpp_f(float v) { return __uint_as_float(dpp_u<CTRL>(__float_as_uint(v))); }
#define QP_BCAST(k) ((k) * 0x55)   // every lane of the quad reads lane k
#define QP_XOR1 0xB1               // [1,0,3,2]
#define QP_XOR2 0x4E               // [2,3,0,1]
#define QP_ROT1 0x39               // [1,2,3,0]: lane j reads lane j+1
#define QP_ROT2 0x4E               // [2,3,0,1]
#define QP_ROT3 0x93               // [3,0,1,2]
__device__ __forceinline__ uint32_t quad_max_u(uint32_t v) {
    v = max(v, dpp_u<QP_XOR1>(v));
    return max(v, dpp_u<QP_XOR2>(v));
}
__device__ __forceinline__ float quad_min_f(float v) {
    v = fminf(v, dpp_f<QP_XOR1>(v));
    return fminf(v, dpp_f<QP_XOR2>(v));
}
// the 4 predicate bits of this lane's quad
__device__ __forceinline__ uint32_t quad_bits(bool p, uint32_t lane) {
    return (uint32_t)(__ballot(p) >> (lane & 60u)) & 15u;
}

struct QuadHit { float t, u, v; uint32_t face; };   // face == 0xFFFFFFFF: nothing

// Traverses the tree under `root` for the quad's ray (all four lanes hold the same ray; `have` false: idle quad).
// ANY = false: closest hit, ties broken by the lower ORIGINAL face index, which is what the reference's ascending
//              strict-< scan keeps (primitives.cpp:376-382). Result (identical in the quad's four lanes) in `out`.
// ANY = true : is there a face with EPS < t < tmax (the shadow test `hit && tmpdist < ldist`, raytracer.cpp:320-321);
//              out.face != 0xFFFFFFFF then.
// `qs` points at this quad's column of the LDS stack (stride 16 words). Called with all 64 lanes active.
// The traversal is resumable: its state is a QTrav (registers) plus the quad's LDS stack column, and quad_run can
// return while some quads are still on their way (serve_traverse hands idle quads new rays then). cur == RT_DONE: done.
struct QTrav {
    uint32_t cur;
    int sp;
    float best, bu, bv;      // this LANE's best candidate
    uint32_t face;
    float tcull;             // the quad's culling distance
    V3 inv, ood;
};
template <bool ANY>
__device__ __forceinline__ void quad_begin(QTrav& t, uint32_t root, bool have, const Ray& ray, float tmax, uint32_t* qs) {
    // culling direction: a zero (or denormal) component is replaced by +-1e-30 so that lo*inv - o*inv keeps the sign of
    // lo - o instead of becoming inf - inf; only the box tests use it, the triangle test gets the ray as it is
    t.inv = mk(safe_rcp(ray.d.x), safe_rcp(ray.d.y), safe_rcp(ray.d.z));
    t.ood = mk(ray.o.x * t.inv.x, ray.o.y * t.inv.y, ray.o.z * t.inv.z);
    t.best = ANY ? tmax : FLT_MAX; t.bu = 0; t.bv = 0;
    t.face = 0xFFFFFFFFu;
    t.tcull = t.best;
    if ((lane_id() & 3u) == 0) qs[0] = RT_DONE;
    t.sp = 1;
    t.cur = have ? root : RT_DONE;      // RT_NO_CHILD == RT_DONE for an empty model
}
// Steps every quad that is not done. keep_lanes == 0: until all are done. Otherwise also returns, between two leaves,
// as soon as fewer than keep_lanes LANES (4 per quad) are still traversing.
template <bool ANY, bool COUNT>
__device__ __forceinline__ void quad_run(const DScene& S, QTrav& t, const Ray& ray, float tmax, uint32_t* qs, uint32_t keep_lanes, Work& wk) {
    const uint32_t lane = lane_id(), j = lane & 3u;
    uint32_t cur = t.cur;
    int sp = t.sp;
    float best = t.best, bu = t.bu, bv = t.bv, tcull = t.tcull;
    uint32_t face = t.face;
    const V3 inv = t.inv, ood = t.ood;
    while (cur != RT_DONE) {
        while (!(cur & RT_LEAF_BIT)) {
            const float4* np = reinterpret_cast<const float4*>(S.nodes + cur) + j;
            const float4 a = np[0], b = np[4];
            if (COUNT) { wk.nodes += (j == 0); wk.wave_nodes += wave_once(); }
            const uint32_t ref = __float_as_uint(a.w);
            bool h;
            float tn = slab(a.x, a.y, a.z, b.x, b.y, b.z, inv, ood, tcull, h);
            // (the ballots of the two comparisons ANDed as scalars: the ballot of `h && ref != ...` goes through a VGPR)
            const uint32_t bits = (uint32_t)((__builtin_amdgcn_ballot_w64(h) & __builtin_amdgcn_ballot_w64(ref != RT_NO_CHILD)) >> (lane & 60u)) & 15u;
            h = h && ref != RT_NO_CHILD;
            const uint32_t n_hit = (uint32_t)__popc(bits);
            // rank of this lane's child among the hit ones, nearest first. tn >= 0, so its bit pattern orders like its
            // value; with the slot in the two lowest bits the four keys are distinct (the visiting order only affects
            // how much is culled, never the result: see tcull)
            // (Shadow rays too: in node order - rank = popc(bits below j), 9 instructions less per step - a single ray finds
            // its occluder later; measured bunny_1080p 0.2438 -> 0.2457 ms, the other workloads unchanged. The PACKETS'
            // shadow rays do go in node order: their "nearest first" is the order of one ray of 64.)
            tn = h ? tn : FLT_MAX;
            const uint32_t key = (__float_as_uint(tn) & ~3u) | j;
            // (the borrow of `neighbour's key - key` IS the comparison: a subtract with the DPP move folded in and an
            // add-with-carry per neighbour, six instructions where move + compare + select + add are nine; s_nop: a DPP
            // operand written by the instruction before needs two wait states, and the compiler does not see in here)
            uint32_t rank = 0, scratch_;
            asm("s_nop 1\n\t"
                "v_sub_co_u32_dpp %1, vcc, %2, %2 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf\n\t"
                "v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n\t"
                "v_sub_co_u32_dpp %1, vcc, %2, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                "v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n\t"
                "v_sub_co_u32_dpp %1, vcc, %2, %2 quad_perm:[3,0,1,2] row_mask:0xf bank_mask:0xf\n\t"
                "v_addc_co_u32_e32 %0, vcc, 0, %0, vcc"
                : "+v"(rank), "=&v"(scratch_) : "v"(key) : "vcc");
            // the nearest is visited next, the others are pushed so that the second nearest is on top
            if (h && rank > 0) qs[(sp + (int)n_hit - 1 - (int)rank) * 16] = ref;
            cur = quad_max_u((h && rank == 0) ? ref : 0u);     // a child reference is never 0 (node 0 is a root)
            if (n_hit == 0) { --sp; cur = qs[sp * 16]; }
            else sp += (int)n_hit - 1;
        }
        if (cur != RT_DONE) {
            const uint32_t start = cur & RT_LEAF_START_MASK;
            const uint32_t cnt = ((cur >> 27) & 15u) + 1u;
            --sp;
            cur = qs[sp * 16];
            if (COUNT) { wk.tris += (j < cnt); wk.wave_tris += wave_once(); }
            bool got = false;
            if (j < cnt) {
                const float4* tp = reinterpret_cast<const float4*>(S.leaves + start) + j;
                const float4 q0 = tp[0], q1 = tp[4], q2 = tp[8];
                float tt, u, v;
                if (tri_isect(f4xyz(q0), f4xyz(q1), f4xyz(q2), ray, tt, u, v)) {
                    const uint32_t f = __float_as_uint(q0.w);
                    if (ANY) {
                        got = tt < tmax;
                    } else if (tt < best || (tt == best && f < face)) {
                        best = tt; bu = u; bv = v; face = f;
                    }
                }
            }
            if (ANY) {
                if (quad_bits(got, lane)) { face = 0; cur = RT_DONE; }
            } else {
                tcull = quad_min_f(best) * 1.000001f;
            }
        }
        // (the lanes that are still in this loop are the ones that count)
        if (keep_lanes && (uint32_t)__popcll(__ballot(cur != RT_DONE)) < keep_lanes) break;
    }
    t.cur = cur; t.sp = sp; t.best = best; t.bu = bu; t.bv = bv; t.tcull = tcull; t.face = face;
}
// The quad's result once its traversal is done (identical in its four lanes).
template <bool ANY>
__device__ __forceinline__ void quad_result(const QTrav& t, QuadHit& out) {
    float best = t.best, bu = t.bu, bv = t.bv;
    uint32_t face = t.face;
    if (!ANY) {
        // the quad's winner: lexicographic minimum of (t, face) over its four lanes
        {
            const float ot = dpp_f<QP_XOR1>(best), ou = dpp_f<QP_XOR1>(bu), ov = dpp_f<QP_XOR1>(bv);
            const uint32_t of = dpp_u<QP_XOR1>(face);
            if (ot < best || (ot == best && of < face)) { best = ot; bu = ou; bv = ov; face = of; }
        }
        {
            const float ot = dpp_f<QP_XOR2>(best), ou = dpp_f<QP_XOR2>(bu), ov = dpp_f<QP_XOR2>(bv);
            const uint32_t of = dpp_u<QP_XOR2>(face);
            if (ot < best || (ot == best && of < face)) { best = ot; bu = ou; bv = ov; face = of; }
        }
    }
    out.t = best; out.u = bu; out.v = bv; out.face = face;
}
template <bool ANY, bool COUNT>
__device__ __forceinline__ void quad_traverse(const DScene& S, uint32_t root, bool have, const Ray& ray, float tmax,
                                              QuadHit& out, uint32_t* qs, Work& wk) {
    QTrav t;
    quad_begin<ANY>(t, root, have, ray, tmax, qs);
    quad_run<ANY, COUNT>(S, t, ray, tmax, qs, 0u, wk);
    quad_result<ANY>(t, out);
}

// ---- packet traversal: the wave's (up to) 64 rays, one per lane, walk the tree TOGETHER --------------------------
// For rays that are coherent - the primary rays of an 8x8-pixel block and the shadow rays of their hits - most of the
// tree a ray visits is the tree its neighbours visit. The packet visits a node when ANY of its rays reaches it: the
// node (128 B) and the leaf blocks come in by SCALAR loads (one address per wave, no vector-memory traffic at all), the
// four children are slab-tested by every lane against its own ray and its own culling distance, the children that
// some ray hit are visited nearest first (by the distance of the first ray that hit each), and the traversal stack is
// one per wave, kept across the lanes of one VGPR (entry k in lane k: a select on the lane number / v_readlane).
// Per 64 rays that is a third of the instructions and dependent loads of four 16-ray quad batches. Same results: a
// lane only ever sees the triangles of leaves whose box its own ray hit or, superfluously, did not - the accepted hit
// is still the lexicographic minimum of (t, original face index) over the triangles its ray hits (closest), or
// "some triangle with EPS < t < tmax" (ANY).
// v_writelane_b32: lane `lane` (wave-uniform, < 64) of v := value (wave-uniform); one VALU instruction
__device__ __forceinline__ int write_lane(int v, uint32_t value, uint32_t lane) {
    const int sv = __builtin_amdgcn_readfirstlane((int)value), sl = __builtin_amdgcn_readfirstlane((int)lane);      // (no-ops on scalars)
    asm("v_writelane_b32 %0, %1, m0" : "+v"(v) : "s"(sv), "{m0}"(sl));      // (one SGPR per VALU instruction: the lane goes through M0)
    return v;
}
// 1 if key != 0xFFFFFFFF else 0, on the scalar unit. (Opaque to the compiler on purpose: written as `key != ~0u ? 1 : 0`
// the count goes through SCC, the stack pointer's whole chain is moved to the vector unit - v_cndmask, v_addc,
// v_readfirstlane back for every push - and the node step grows by 13 vector instructions. The stack pointer itself
// must not be an asm operand either: s_cmp + s_addc on it put it into a VGPR in the closest-hit kernels.)
__device__ __forceinline__ uint32_t s_is_valid(uint32_t key) {
    uint32_t r;
    asm("s_not_b32 %0, %1\n\ts_min_u32 %0, %0, 1" : "=s"(r) : "s"(key) : "scc");
    return r;
}
// Leaves worked off ACROSS the lanes (shadow packets). A packet reaches a leaf when any of its rays hits the leaf's box,
// but on average only a quarter of its rays do, and in the flat form every lane tests all (up to) four triangles:
// 3.6 x ~50 vector instructions per leaf step, three quarters of the lanes for nothing. The node step knows which
// lanes hit each child (the ballots it branches on), so those masks travel with the stack entries (two more lane
// stacks). A leaf reached by <= 16 live rays is then tested in ONE go: the reaching lanes put their rays into LDS by
// rank, quad q reads ray q, lane j of the quad fetches triangle j of the block itself (three 16-B vector loads, as the
// quad kernels do) and tests it, and a reaching lane whose quad reports a hit leaves the packet. A leaf that no live
// ray reaches any more (they were occluded on the way) is skipped. Same answer: every (reaching ray, triangle) pair is
// tested with the same arithmetic; a ray that did not hit the box cannot hit a triangle inside it.
// MEASURED AND NOT USED (default 0; A/B builds). The test itself gets cheaper - level-0 shade launch of bunny_1080p 60.5 M
// -> 56.4 M vector instructions, 0.152 -> 0.140 ms; 10 M-triangle field at 1080p 1.54 -> 1.20 ms - but the path needs ~25
// more VGPRs (the quad's ray and triangle next to the lane's own state): at the kernel's 7 waves per SIMD they spill
// (8K field 402 -> 416 ms, 4K glass scene 27.4 -> 29.1, c3 0.133 -> 0.149), and at 6 waves the batched frames, which live
// on occupancy, lose more than the leaves win (bunny_1080p 0.2105 -> 0.219 ms per frame; the same kernel WITHOUT
// compaction at 6 waves: 0.219). Only the field at 1080p 4 spp gained end to end (3.93 -> 3.76 ms).
#ifndef RT_LEAF_COMPACT
#define RT_LEAF_COMPACT 0
#endif
#ifndef RT_LEAF_COMPACT_TRI
#define RT_LEAF_COMPACT_TRI 0     // the compacted leaf's triangles: 0 three vector loads per lane, 1 scalar loads + per-lane selects
#endif
#ifndef RT_LEAF_COMPACT_MAX
#define RT_LEAF_COMPACT_MAX 16u   // (A/B builds: a leaf reached by more live rays is tested in the flat form)
#endif
#define RT_PK_LDS_WORDS 128      // per wave: 16 rays x (origin, tmax, direction, pad)
template <bool ANY, bool COUNT>
__device__ __forceinline__ void packet_traverse(const DScene& S, uint32_t root, bool have, const Ray& ray, float tmax,
                                                QuadHit& out, Work& wk, uint32_t* pk_lds = nullptr) {
    const bool compact = ANY && RT_LEAF_COMPACT && RT_TRI_MASKS_ANY && RT_ANY_UNORDERED && pk_lds != nullptr;      // (compile time but for the pointer)
    int mlo = 0, mhi = 0;                   // lane k = the lanes that hit the box of stack entry k (compact)
    unsigned long long cur_m = ~0ull;       // ... of the node or leaf being visited
    const V3 inv = mk(safe_rcp(ray.d.x), safe_rcp(ray.d.y), safe_rcp(ray.d.z));
    const V3 ood = mk(ray.o.x * inv.x, ray.o.y * inv.y, ray.o.z * inv.z);
    float best = ANY ? tmax : FLT_MAX, bu = 0, bv = 0;
    uint32_t face = 0xFFFFFFFFu;
    int stack = 0;                          // lane k = stack entry k
    uint32_t sp = 0;
    bool live = have;                       // (the flat form of the leaf test: a lane's bit of live_m)
    const unsigned long long have_m = __builtin_amdgcn_ballot_w64(have);
    unsigned long long live_m = have_m;     // the lanes whose ray is still on its way, a scalar (ANY: a lane leaves once its ray is occluded)
    uint32_t cur = (uint32_t)__builtin_amdgcn_readfirstlane((int)(live_m ? root : RT_DONE));   // RT_NO_CHILD == RT_DONE
    while (cur != RT_DONE) {
        while (!(cur & RT_LEAF_BIT)) {      // (RT_DONE has the leaf bit)
            const RT_CONSTANT u32x4* np = (const RT_CONSTANT u32x4*)(const void*)(S.nodes) + 8u * (size_t)cur;
            if (COUNT) { wk.nodes += __builtin_amdgcn_inverse_ballot_w64(live_m); wk.wave_nodes += wave_once(); }
            const float cull = ANY ? best : best * 1.000001f;
            uint32_t key[4], ref[4];
            unsigned long long cm[4] = {0, 0, 0, 0};      // (compact) the live lanes that hit each child
            // the whole node is requested at once (scalar loads return in any order, so the only wait there is is for
            // all of them: four requests one after the other would be four latencies), and nothing below branches
            u32x4 nd[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) nd[c] = np[c];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float4 a = as_f4(nd[c]), b = as_f4(nd[4 + c]);
                ref[c] = __float_as_uint(a.w);
                bool h;
                const float tn = slab(a.x, a.y, a.z, b.x, b.y, b.z, inv, ood, cull, h);
                // (the ballot of the comparison itself, masked by scalars: a ballot of `h && live && ...` goes through a VGPR)
                // The scalar side of a node step weighs as much as its vector side (the ANY form below, which drops the sort,
                // made the level-0 shade launch 7 % faster), so the key of a child is four instructions written out: s_and_b64
                // leaves "some ray hit" in SCC, which the compiler's own sequence (compare with zero, selects for the lane
                // number and for the key: 8 scalar instructions) does not use. v_readlane takes the lane from bits 5:0 of
                // its operand (s_ff1 of an empty mask is -1: lane 63, and the key becomes -1 anyway). No wait states needed:
                // the lane select is written by the scalar unit, the select after it reads an SGPR like any other.
                // (m = the rays that hit the child, none for an empty slot)
                const unsigned long long hm = __builtin_amdgcn_ballot_w64(h);
                unsigned long long m;
                if ((ANY && RT_ANY_UNORDERED) || RT_CLOSEST_ORDER == 2) {       // 1 = hit by some ray
                    asm("s_cmp_lg_u32 %4, -1\n\ts_cselect_b64 %0, %3, 0\n\ts_and_b64 %0, %2, %0\n\ts_cselect_b32 %1, 1, 0"
                        : "=&s"(m), "=s"(key[c]) : "s"(hm), "s"(live_m), "s"(ref[c]) : "scc");
                    cm[c] = m;
                    continue;
                }
                // distance of the first ray that hits the child (tn >= 0: its bits order like its value); none: last
                uint32_t first_lane;
                asm("s_cmp_lg_u32 %6, -1\n\ts_cselect_b64 %0, %4, 0\n\ts_and_b64 %0, %3, %0\n\ts_ff1_i32_b64 %1, %0\n\t"
                    "v_readlane_b32 %2, %5, %1\n\ts_cselect_b32 %2, %2, -1"
                    : "=&s"(m), "=&s"(first_lane), "=&s"(key[c]) : "s"(hm), "s"(live_m), "v"(tn), "s"(ref[c]) : "scc");
            }
            if ((ANY && RT_ANY_UNORDERED) || RT_CLOSEST_ORDER == 2) {
                // shadow rays: any occluder will do, so the children are visited in the order they lie in the node - no
                // v_readlane of the distances, no sort (same answer: occluded or not does not depend on the order)
                if (compact) {
#pragma unroll
                    for (int c = 3; c >= 1; --c) {
                        stack = write_lane(stack, ref[c], sp);
                        mlo = write_lane(mlo, (uint32_t)cm[c], sp); mhi = write_lane(mhi, (uint32_t)(cm[c] >> 32), sp);
                        sp += key[c];
                    }
                    if (key[0]) { cur = ref[0]; cur_m = cm[0]; }
                    else if (sp) {
                        --sp; cur = (uint32_t)__builtin_amdgcn_readlane(stack, (int)sp);
                        cur_m = (uint32_t)__builtin_amdgcn_readlane(mlo, (int)sp) | ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane(mhi, (int)sp) << 32);
                    }
                    else cur = RT_DONE;
                    continue;
                }
                stack = write_lane(stack, ref[3], sp); sp += key[3];
                stack = write_lane(stack, ref[2], sp); sp += key[2];
                stack = write_lane(stack, ref[1], sp); sp += key[1];
                if (key[0]) cur = ref[0];
                else if (sp) { --sp; cur = (uint32_t)__builtin_amdgcn_readlane(stack, (int)sp); }
                else cur = RT_DONE;
                continue;
            }
            // (Measured and not kept: the packet walked once per direction OCTANT of its rays, so that the entering / leaving
            // plane of each axis is known and slab()'s six min/max per box fall away - with the planes picked by scalar selects
            // 0.301 -> 0.310 ms per frame, with one instantiation of the box tests per octant 0.327: an SALU instruction costs
            // a SIMD about as much issue time as a VALU min/max - one scalar unit serves the CU's four SIMDs,
            // tools/valu_rate.hip - and both forms pushed more scalars into spill lanes.)
            // (Also measured and not kept: leaves worked off ACROSS the lanes. A leaf is reached by few of the packet's rays,
            // yet every lane tests its four triangles; with the mask of the lanes that hit each box kept beside the stack
            // entries, a leaf reached by <= 16 rays was tested in one go - quad q pulls the q-th ray, lane j tests triangle
            // j, the owner pulls the result back. Same bytes; the masks' bookkeeping (two more lane stacks, the sort's
            // payload) cost 8-14 % of the level-0 kernels and the cheaper leaves won back only part of it: bunny_1080p
            // 0.257 -> 0.299 ms per frame, 10 M-triangle field at 1080p 4.57 -> 4.83; tools/exp_ab_builds.sh.)
            // sort the four (key, ref) pairs, all wave-uniform: 5 compare-exchanges
            // (s_min_u32 leaves "the first operand is the minimum" in SCC: four instructions per compare-exchange, not five)
#define RT_CSWAP(i, j) { uint32_t k0_, k1_, r0_, r1_; \
                         asm("s_min_u32 %0, %4, %5\n\ts_cselect_b32 %2, %6, %7\n\ts_cselect_b32 %3, %7, %6\n\ts_max_u32 %1, %4, %5" \
                             : "=&s"(k0_), "=&s"(k1_), "=&s"(r0_), "=&s"(r1_) : "s"(key[i]), "s"(key[j]), "s"(ref[i]), "s"(ref[j]) : "scc"); \
                         key[i] = k0_; key[j] = k1_; ref[i] = r0_; ref[j] = r1_; }
            RT_CSWAP(0, 1) RT_CSWAP(2, 3) RT_CSWAP(0, 2)
            if (RT_CLOSEST_ORDER == 0) { RT_CSWAP(1, 3) RT_CSWAP(1, 2) }
#undef RT_CSWAP
            // the farthest goes in first, the nearest is visited next
            // (one v_writelane per entry, written whether the child was hit or not: an entry that was not hit lands above the
            // top of the stack - sp <= 63, the builders bound the traversal stack by RT_QSTACK - 1 pending entries - and is
            // overwritten by the next push)
            stack = write_lane(stack, ref[3], sp); sp += s_is_valid(key[3]);
            stack = write_lane(stack, ref[2], sp); sp += s_is_valid(key[2]);
            stack = write_lane(stack, ref[1], sp); sp += s_is_valid(key[1]);
            if (key[0] != 0xFFFFFFFFu) cur = ref[0];
            else if (sp) { --sp; cur = (uint32_t)__builtin_amdgcn_readlane(stack, (int)sp); }
            else cur = RT_DONE;
        }
        if (cur == RT_DONE) break;
        {
            const uint32_t cnt = ((cur >> 27) & 15u) + 1u;
            const RT_CONSTANT u32x4* lp = (const RT_CONSTANT u32x4*)(const void*)(S.leaves) + 12u * (size_t)(cur & RT_LEAF_START_MASK);
            if (COUNT) { wk.tris += __builtin_amdgcn_inverse_ballot_w64(live_m) ? cnt : 0u; wk.wave_tris += wave_once(); }
            const unsigned long long reach = cur_m & live_m;
            const uint32_t n_reach = compact ? (uint32_t)__popcll(reach) : 64u;
            if (compact && n_reach <= RT_LEAF_COMPACT_MAX) {
                if (n_reach) {
                    const uint32_t lane = lane_id(), q = lane >> 2, j = lane & 3u;
#if RT_LEAF_COMPACT_TRI == 0
                    const float4* tp = reinterpret_cast<const float4*>(S.leaves + (cur & RT_LEAF_START_MASK)) + j;
                    const float4 q0 = tp[0], q1 = tp[4], q2 = tp[8];            // triangle j of the block (requested first: the longest wait)
#else
                    // the block by scalar loads (as the flat form), triangle j picked per lane: 27 selects, no vector memory
                    float4 q0, q1, q2;
                    {
                        u32x4 lf[12];
#pragma unroll
                        for (int c = 0; c < 12; ++c) lf[c] = lp[c];
                        const bool j1 = (j & 1u) != 0, j2 = (j & 2u) != 0;
#define RT_PICK(c, k) __uint_as_float(j2 ? (j1 ? lf[c + 3].k : lf[c + 2].k) : (j1 ? lf[c + 1].k : lf[c].k))
                        q0 = make_float4(RT_PICK(0, x), RT_PICK(0, y), RT_PICK(0, z), 0);
                        q1 = make_float4(RT_PICK(4, x), RT_PICK(4, y), RT_PICK(4, z), 0);
                        q2 = make_float4(RT_PICK(8, x), RT_PICK(8, y), RT_PICK(8, z), 0);
#undef RT_PICK
                    }
#endif
                    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(reach >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)reach, 0u));
                    const bool mine = __builtin_amdgcn_inverse_ballot_w64(reach);
                    float4* slot = reinterpret_cast<float4*>(pk_lds);
                    if (mine) {
                        slot[2u * rank] = make_float4(ray.o.x, ray.o.y, ray.o.z, tmax);
                        slot[2u * rank + 1u] = make_float4(ray.d.x, ray.d.y, ray.d.z, 0.0f);
                    }
                    const float4 ro = slot[2u * q], rd = slot[2u * q + 1u];      // (LDS operations of one wave complete in order)
                    Ray qr;
                    qr.o = f4xyz(ro); qr.d = f4xyz(rd);
                    unsigned long long m = __builtin_amdgcn_ballot_w64(q < n_reach) & __builtin_amdgcn_ballot_w64(j < cnt);
                    float tt = 0, u = 0, v = 0;
                    tri_isect_packet<0>(f4xyz(q0), f4xyz(q1), f4xyz(q2), qr, m, tt, u, v);
                    m &= __builtin_amdgcn_ballot_w64(tt < ro.w);
                    if (m) {                                                                             // wave-uniform
                        const bool occ = mine && ((uint32_t)(m >> (4u * rank)) & 15u) != 0u;      // my quad reports a hit
                        live_m &= ~__builtin_amdgcn_ballot_w64(occ);
                    }
                }
            } else {
            // two triangles per request (6 x 16 B, one wait): four requests one after the other are four latencies, the whole
            // block at once (48 SGPRs) makes the kernel spill scalars around the loop - and every spilled SGPR is a VALU
            // instruction (v_writelane / v_readlane), which is what this path is short of
#pragma unroll
            for (uint32_t t0 = 0; t0 < RT_MAX_LEAF_TRIS; t0 += 2) {
                if (t0 < cnt) {                                                                  // wave-uniform
                    u32x4 lf[6];
#pragma unroll
                    for (int c = 0; c < 3; ++c) { lf[2 * c] = lp[4 * c + t0]; lf[2 * c + 1] = lp[4 * c + t0 + 1]; }
                    // (both triangles' words are waited for HERE: with exits in the first triangle's test the compiler sinks the
                    // second triangle's loads below it, and a leaf is four fetch latencies one after the other instead of two)
                    if (RT_TRI_PIN_LOADS && (ANY ? RT_TRI_MASKS_ANY : RT_TRI_MASKS_CLOSEST)) asm volatile("" :: "s"(lf[1]), "s"(lf[3]), "s"(lf[5]));
#pragma unroll
                    for (uint32_t dt = 0; dt < 2; ++dt) {
                        if (t0 + dt < cnt) {                                                     // wave-uniform
                            RT_UNIFORM_BRANCH;
                            const float4 q0 = as_f4(lf[dt]), q1 = as_f4(lf[2 + dt]), q2 = as_f4(lf[4 + dt]);
                            float tt = 0, u = 0, v = 0;
                            if (!(ANY ? RT_TRI_MASKS_ANY : RT_TRI_MASKS_CLOSEST)) {      // the flat form: one block per triangle, per-lane conditions
                                if (live && tri_isect(f4xyz(q0), f4xyz(q1), f4xyz(q2), ray, tt, u, v)) {
                                    const uint32_t f = __float_as_uint(q0.w);
                                    if (ANY) {
                                        if (tt < tmax) live = false;
                                    } else if (tt < best || (tt == best && f < face)) {
                                        best = tt; bu = u; bv = v; face = f;
                                    }
                                }
                                continue;
                            }
                            unsigned long long m = live_m;      // the lanes whose ray is tested; then: whose ray hits
                            tri_isect_packet<ANY ? RT_TRI_EXITS_ANY : RT_TRI_EXITS_CLOSEST>(f4xyz(q0), f4xyz(q1), f4xyz(q2), ray, m, tt, u, v);
                            if (ANY) {      // occluded lanes leave the packet (a scalar mask; `face` is set at the end)
                                if (m) live_m &= ~(m & __builtin_amdgcn_ballot_w64(tt < tmax));         // wave-uniform
                            } else {        // (no short-circuit: the three comparisons and four selects, no branches)
                                const uint32_t f = __float_as_uint(q0.w);
                                const unsigned long long better = m & (__builtin_amdgcn_ballot_w64(tt < best) |
                                    (__builtin_amdgcn_ballot_w64(tt == best) & __builtin_amdgcn_ballot_w64(f < face)));
                                if (__builtin_amdgcn_inverse_ballot_w64(better)) { best = tt; bu = u; bv = v; face = f; }
                            }
                        }
                    }
                }
            }
            }
            if (ANY && !RT_TRI_MASKS_ANY) live_m = __builtin_amdgcn_ballot_w64(live);
            if (ANY && live_m == 0) cur = RT_DONE;
            else if (sp) {
                --sp; cur = (uint32_t)__builtin_amdgcn_readlane(stack, (int)sp);
                if (compact) cur_m = (uint32_t)__builtin_amdgcn_readlane(mlo, (int)sp) | ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane(mhi, (int)sp) << 32);
            }
            else cur = RT_DONE;
        }
    }
    if (ANY) face = __builtin_amdgcn_inverse_ballot_w64(have_m & ~live_m) ? 0u : 0xFFFFFFFFu;      // occluded: some face
    out.t = best; out.u = bu; out.v = bv; out.face = face;
}

// ---- quad service: a per-lane kernel whose rays reach a mesh traverses it with its own quads ---------------------
// The lanes that need the traversal (`need`) are served by the wave's 16 quads: a quad takes over the ray of the next
// needing lane (7 cross-lane moves in), traverses it (quad_run) and leaves the result in that lane's LDS mailbox.
// Quads do not wait for each other in batches: rays differ a lot in length (bounce rays of the 10 M-triangle field:
// 18 node steps on average, 47 for the longest of 16 - lockstep batches kept the quads 38 % busy), so as soon as a
// quarter of the quads have finished, the traversal pauses, the idle quads take the next rays, and it goes on. Everything else of the
// ray - analytic objects, Fresnel, Phong - stays one ray per lane, where a quad kernel computes it four times over.
// For the bounce levels, whose rays have nothing in common (no packets) and mostly do reach the mesh (so queueing
// them, which costs a round trip of their state through memory, does not pay).
// `lds` = RT_SERVE_LDS words: the wave's 16-column stack (RT_QSTACK x 16) and the mailboxes (4 words per lane). Called
// with all 64 lanes active; same results as quad_traverse on each ray.
#define RT_SERVE_LDS (RT_QSTACK * 16 + 256)
#ifndef RT_SERVE_KEEP
#define RT_SERVE_KEEP 12u      // quads: the traversal pauses for a refill when fewer are still busy (and rays are left)
#endif
__device__ __forceinline__ uint32_t select_bit(unsigned long long mask, uint32_t q);      // index of the q-th set bit (below)
template <bool ANY, bool COUNT>
__device__ __forceinline__ void serve_traverse(const DScene& S, uint32_t root, bool need, const Ray& ray, float tmax,
                                               QuadHit& out, uint32_t* lds, Work& wk) {
    const uint32_t lane = lane_id(), quad = lane >> 2, j = lane & 3u;
    uint32_t* qs = lds + quad;
    uint32_t* mail = lds + RT_QSTACK * 16;
    out.face = 0xFFFFFFFFu; out.t = 0; out.u = 0; out.v = 0;
    const unsigned long long m = __builtin_amdgcn_ballot_w64(need);
    const uint32_t n = (uint32_t)__popcll(m);                                        // wave-uniform
    if (n == 0) return;
    uint32_t next = 0;                   // needing lanes handed out so far (wave-uniform)
    bool busy = false;                   // this quad is on a ray
    uint32_t owner = lane;               // ... of this lane
    Ray qr;
    qr.o = mk(0, 0, 0); qr.d = mk(0, 0, 0);
    float qt = FLT_MAX;
    QTrav t;
    t.cur = RT_DONE; t.sp = 1; t.best = 0; t.bu = 0; t.bv = 0; t.face = 0xFFFFFFFFu; t.tcull = 0; t.inv = mk(0, 0, 0); t.ood = mk(0, 0, 0);
    for (;;) {
        // the idle quads, lowest first, take the next needing lanes
        const unsigned long long idle_q = __builtin_amdgcn_ballot_w64(!busy) & 0x1111111111111111ull;      // one bit per quad
        const uint32_t n_idle = (uint32_t)__popcll(idle_q);
        const uint32_t want = next + (uint32_t)__popcll(idle_q & ((1ull << (lane & 60u)) - 1ull));
        const bool take = !busy && want < n;
        const int src = (int)(take ? select_bit(m, want) : lane);
        const float ox = __shfl(ray.o.x, src), oy = __shfl(ray.o.y, src), oz = __shfl(ray.o.z, src);
        const float dx = __shfl(ray.d.x, src), dy = __shfl(ray.d.y, src), dz = __shfl(ray.d.z, src);
        const float st = ANY ? __shfl(tmax, src) : FLT_MAX;
        if (take) {
            qr.o = mk(ox, oy, oz); qr.d = mk(dx, dy, dz); qt = st;
            owner = (uint32_t)src;
            busy = true;
            quad_begin<ANY>(t, root, true, qr, qt, qs);
        }
        next = next + n_idle < n ? next + n_idle : n;
        if (__builtin_amdgcn_ballot_w64(busy) == 0) break;                             // every ray served
        const uint32_t keep = (RT_DBG(S.debug) >> 16) ? ((RT_DBG(S.debug) >> 16) & 31u) : RT_SERVE_KEEP;      // (experiments: RT_DEBUG_SKIP = keep << 16)
        quad_run<ANY, COUNT>(S, t, qr, qt, qs, next < n ? keep * 4u : 0u, wk);
        if (busy && t.cur == RT_DONE) {
            QuadHit q;
            quad_result<ANY>(t, q);
            if (j == 0) {
                uint32_t* box = mail + owner * 4u;
                box[0] = q.face; box[1] = __float_as_uint(q.t); box[2] = __float_as_uint(q.u); box[3] = __float_as_uint(q.v);
            }
            busy = false;
        }
    }
    if (need) {      // (LDS operations of one wave complete in order: the mailbox holds the quad's words)
        const uint32_t* box = mail + lane * 4u;
        out.face = box[0];
        if (!ANY) { out.t = __uint_as_float(box[1]); out.u = __uint_as_float(box[2]); out.v = __uint_as_float(box[3]); }
    }
}

// ---- lane traversal: one ray per LANE, every lane walks the tree by itself --------------------------------------
// Quad service keeps 16 rays of a wave on their way at a time, and a ray's traversal is a chain of dependent fetches:
// where the tree does not fit the L2 (the 10 M-triangle field: 700 MB) or shares it with the pool streaming through
// (the 4K glass scene), the bounce-level kernels wait for memory two thirds of their time at a fraction of the HBM
// bandwidth - they are bound by how many fetch chains the chip holds, not by instructions. Here each lane fetches and
// slab-tests the four children of ITS node (eight 16-B loads), descends into the nearest hit child and pushes the others
// onto its own stack (LDS, [entry][lane]: conflict-free), then tests the up to four triangles of its leaf one after the
// other: 64 chains per wave instead of 16, for about the same instructions per ray while the lanes keep busy (a wave's
// node loop runs until every lane has reached a leaf, its leaf step until every lane is through).
// Same results as quad_traverse on each ray: every triangle a ray hits is found (boxes only cull, with the same
// conservative slab test), the closest hit is the lexicographic minimum of (t, original face index), ANY stops at the
// first triangle with EPS < t < tmax.
// The stack holds RT_LANE_STACK entries per lane (the wave's quad-service block of LDS, used as [20][64]); a node step
// writes its four children above the top and keeps those that were hit, so a lane that arrives at a node with more
// than RT_LANE_STACK_LIMIT entries gives up (returns false) and its ray is traversed again by quad service, whose
// stack the builders bound: the result never depends on the short stack.
// MEASURED AND NOT USED (default 0; A/B builds, tools/r3_run8.sh: same bytes in every parity test, also with a stack
// limit of 1, where every ray takes the fallback). Slower wherever quad service runs today: bunny_1080p in batches 0.212
// -> 0.233 ms per frame (k_extend_serve 0.062 -> 0.084 ms, k_shade_serve 0.045 -> 0.059), 10 M-triangle field at 1080p 3.93
// -> 4.34 ms, 4K glass scene 27.4 -> 34.2 ms. A lane's eight 16-B loads lie in ONE cache line and the lines of the 64 lanes
// are all different: a node step is 8 vector-memory instructions of 64 line look-ups each where a quad step is 2
// instructions of 16 (each quad's four lanes read 64 contiguous bytes) - four times the work for the L1's address and tag
// pipeline per ray, the unit these latency-bound kernels queue at - and the leaf step tests a lane's four triangles one
// after the other at the pace of the slowest lane. More fetch chains per wave do not pay for that.
#ifndef RT_LANE_TRAVERSE
#define RT_LANE_TRAVERSE 0      // (1: A/B builds)
#endif
#define RT_LANE_STACK (RT_SERVE_LDS / 64)
#ifndef RT_LANE_STACK_LIMIT
#define RT_LANE_STACK_LIMIT (RT_LANE_STACK - 4)      // (test builds: 1, so that the fallback runs)
#endif
template <bool ANY, bool COUNT>
__device__ __forceinline__ bool lane_traverse(const DScene& S, uint32_t root, bool need, const Ray& ray, float tmax,
                                              QuadHit& out, uint32_t* lds, Work& wk) {
    uint32_t* st = lds + lane_id();
    const V3 inv = mk(safe_rcp(ray.d.x), safe_rcp(ray.d.y), safe_rcp(ray.d.z));
    const V3 ood = mk(ray.o.x * inv.x, ray.o.y * inv.y, ray.o.z * inv.z);
    float best = ANY ? tmax : FLT_MAX, bu = 0, bv = 0;
    uint32_t face = 0xFFFFFFFFu;
    uint32_t cur = need ? root : RT_DONE;      // RT_NO_CHILD == RT_DONE for an empty model
    uint32_t sp = 0;
    bool ok = true;
    while (__builtin_amdgcn_ballot_w64(cur != RT_DONE)) {
        while (!(cur & RT_LEAF_BIT)) {      // (per lane; RT_DONE has the leaf bit)
            if (sp > RT_LANE_STACK_LIMIT) { ok = false; cur = RT_DONE; break; }
            const float4* np = reinterpret_cast<const float4*>(S.nodes + cur);
            const float4 a0 = np[0], a1 = np[1], a2 = np[2], a3 = np[3], b0 = np[4], b1 = np[5], b2 = np[6], b3 = np[7];
            if (COUNT) wk.nodes += 1;
            const float cull = ANY ? best : best * 1.000001f;
            const uint32_t r0 = __float_as_uint(a0.w), r1 = __float_as_uint(a1.w), r2 = __float_as_uint(a2.w), r3 = __float_as_uint(a3.w);
            bool h0, h1, h2, h3;
            const float t0 = slab(a0.x, a0.y, a0.z, b0.x, b0.y, b0.z, inv, ood, cull, h0);
            const float t1 = slab(a1.x, a1.y, a1.z, b1.x, b1.y, b1.z, inv, ood, cull, h1);
            const float t2 = slab(a2.x, a2.y, a2.z, b2.x, b2.y, b2.z, inv, ood, cull, h2);
            const float t3 = slab(a3.x, a3.y, a3.z, b3.x, b3.y, b3.z, inv, ood, cull, h3);
            h0 = h0 && r0 != RT_NO_CHILD; h1 = h1 && r1 != RT_NO_CHILD; h2 = h2 && r2 != RT_NO_CHILD; h3 = h3 && r3 != RT_NO_CHILD;
            // entry distance >= 0: its bits order like its value; the slot in the two lowest bits makes the keys distinct
            const uint32_t k0 = h0 ? ((__float_as_uint(t0) & ~3u) | 0u) : 0xFFFFFFFFu;
            const uint32_t k1 = h1 ? ((__float_as_uint(t1) & ~3u) | 1u) : 0xFFFFFFFFu;
            const uint32_t k2 = h2 ? ((__float_as_uint(t2) & ~3u) | 2u) : 0xFFFFFFFFu;
            const uint32_t k3 = h3 ? ((__float_as_uint(t3) & ~3u) | 3u) : 0xFFFFFFFFu;
            const uint32_t kmin = min(min(k0, k1), min(k2, k3));
            // the nearest is visited next; the other hit children are pushed as they lie in the node (every child is written
            // above the top, the top moves past those that stay)
            st[sp * 64u] = r0; sp += (h0 && k0 != kmin) ? 1u : 0u;
            st[sp * 64u] = r1; sp += (h1 && k1 != kmin) ? 1u : 0u;
            st[sp * 64u] = r2; sp += (h2 && k2 != kmin) ? 1u : 0u;
            st[sp * 64u] = r3; sp += (h3 && k3 != kmin) ? 1u : 0u;
            if (kmin == 0xFFFFFFFFu) {
                if (sp) { --sp; cur = st[sp * 64u]; } else cur = RT_DONE;
            } else {
                cur = k0 == kmin ? r0 : k1 == kmin ? r1 : k2 == kmin ? r2 : r3;
            }
        }
        if (cur != RT_DONE) {
            const uint32_t cnt = ((cur >> 27) & 15u) + 1u;
            const float4* tp = reinterpret_cast<const float4*>(S.leaves + (cur & RT_LEAF_START_MASK));
            if (COUNT) wk.tris += cnt;
            bool got = false;
#pragma unroll
            for (uint32_t t = 0; t < RT_MAX_LEAF_TRIS; ++t) {
                if (t < cnt) {
                    const float4 q0 = tp[t], q1 = tp[4 + t], q2 = tp[8 + t];
                    float tt, u, v;
                    if (tri_isect(f4xyz(q0), f4xyz(q1), f4xyz(q2), ray, tt, u, v)) {
                        const uint32_t f = __float_as_uint(q0.w);
                        if (ANY) {
                            got = got || tt < tmax;
                        } else if (tt < best || (tt == best && f < face)) {
                            best = tt; bu = u; bv = v; face = f;
                        }
                    }
                }
            }
            if (ANY && got) { face = 0; cur = RT_DONE; }
            else if (sp) { --sp; cur = st[sp * 64u]; }
            else cur = RT_DONE;
        }
    }
    out.t = best; out.u = bu; out.v = bv; out.face = face;
    return ok;
}
// The mesh traversal of the per-lane kernels that suspend nothing (PACKET == 2): every needing lane by itself, and the
// few whose short stack did not do by quad service. Called with all 64 lanes active.
template <bool ANY, bool COUNT>
__device__ __forceinline__ void mesh_serve(const DScene& S, uint32_t root, bool need, const Ray& ray, float tmax,
                                           QuadHit& out, uint32_t* lds, Work& wk) {
    if (!RT_LANE_TRAVERSE) { serve_traverse<ANY, COUNT>(S, root, need, ray, tmax, out, lds, wk); return; }
    out.face = 0xFFFFFFFFu; out.t = 0; out.u = 0; out.v = 0;
    if (__builtin_amdgcn_ballot_w64(need) == 0) return;                                // wave-uniform
    const bool ok = lane_traverse<ANY, COUNT>(S, root, need, ray, tmax, out, lds, wk);
    const bool redo = need && !ok;
    if (__builtin_amdgcn_ballot_w64(redo)) {                                           // wave-uniform, rare
        QuadHit q;
        serve_traverse<ANY, COUNT>(S, root, redo, ray, tmax, q, lds, wk);
        if (redo) out = q;
    }
}

// ---------------------------------------------------------------------------------------------------------
// The two object loops of trace(), each in two forms:
//   QUAD = false  one ray per lane (64 per wave). Spheres and planes are tested in place; a ray that passes a mesh's
//                 bounding-sphere gate SUSPENDS at that object: the loop reports the object index and the caller
//                 saves the ray's state and queues it for the quad form.
//   QUAD = true   one ray per quad (16 per wave; the four lanes hold the same values and compute the sphere / plane
//                 tests redundantly). Meshes are traversed in place. Starts at object k0, so it can resume a
//                 suspended ray (or run a whole ray with k0 = 0).
// A wave that carries its 64 rays through a mesh itself would need four 16-ray batches one after the other; queueing
// them makes those batches independent waves, so the longest dependent chain on the chip is ONE batch.
// Object table entry k = two uint4: {type, index, material, sphere 1/r bits} and the float4 parameters of a sphere
// (cx cy cz r) or plane (a b c d). k is wave-uniform, so both come in by scalar loads; the next entry is requested
// before the current one is worked on.
// ---------------------------------------------------------------------------------------------------------
struct HitState { float dist; V3 n; int cur; uint32_t mat; float tmpdist; };

// Closest-hit loop (reference src/raytracer.cpp:292-300): objects in scene order, strict <. Returns -1 when the loop
// ran to its end, else (QUAD = false only) the index of the mesh object the ray suspended at, with `h` as it was
// before that object.
template <bool QUAD, bool COUNT, int PACKET = 0>      // PACKET (per-lane form): 0 suspend at a mesh, 1 packets, 2 quad service
__device__ __forceinline__ int closest_loop(const DScene& S, const Ray& ray, bool alive, uint32_t k0, HitState& h,
                                            uint32_t* qs, Work& wk, bool coherent = false) {
    int susp = -1;
    V3 tmpnormal = mk(0, 0, 0);
    if (S.n_objects == 0) return susp;
    const bool counter = !QUAD || (lane_id() & 3u) == 0;
    // (the table ends with an entry of zeros, so entry k + 1 always exists; a walking pointer: two scalar instructions
    // per object where index, clamp, shift and add were seven)
    const RT_CONSTANT u32x4* op = (const RT_CONSTANT u32x4*)(const void*)S.objects;
    u32x4 ob = op[0], pb = op[1];
    for (uint32_t k = 0; k < S.n_objects; ++k) {
        op += 2;
        const u32x4 ob_n = op[0], pb_n = op[1];
        const float4 prm = as_f4(pb);
        const bool go = alive && k >= k0 && susp < 0;
        bool hit = false;
        if (ob.x == RT_OBJ_SPHERE) {
            RT_UNIFORM_BRANCH;
            if (COUNT) wk.spheres += go && counter;
            if (go && !(RT_DBG(S.debug) & 16u)) hit = sphere_isect(prm, __uint_as_float(ob.w), ray, h.tmpdist, tmpnormal, coherent);
        }
        if (ob.x == RT_OBJ_PLANE) {
            RT_UNIFORM_BRANCH;
            if (COUNT) wk.planes += go && counter;
            if (go && !(RT_DBG(S.debug) & 16u)) hit = plane_isect(prm, ray, h.tmpdist, tmpnormal);
        }
        if (ob.x != RT_OBJ_SPHERE && ob.x != RT_OBJ_PLANE) {      // (separate tests: see occluded_loop)
            RT_UNIFORM_BRANCH;
            // Model::intersect (reference src/primitives.cpp:364-384): dist = FLT_MAX, bounding-sphere gate, closest
            // face, normal = (1-u-v)*n0 + u*n1 + v*n2 (primitives.cpp:80, not normalised)
            const ModelRef m = load_model(S.models, ob.y);
            bool need = go && !(RT_DBG(S.debug) & 8u);
            if (m.bb.w >= 0) {
                if (COUNT) wk.spheres += need && counter;
                need = need && sphere_gate(m.bb, ray, coherent);
            }
            if (!QUAD && !PACKET) {
                if (need) susp = (int)k;
                else if (go) h.tmpdist = FLT_MAX;
            } else {
                if (go) h.tmpdist = FLT_MAX;
                QuadHit q;
                q.face = 0xFFFFFFFFu; q.t = 0; q.u = 0; q.v = 0;
                if (PACKET == 2) {
                    mesh_serve<false, COUNT>(S, m.root, need, ray, FLT_MAX, q, qs, wk);
                } else if (PACKET) {
                    if (__ballot(need)) packet_traverse<false, COUNT>(S, m.root, need, ray, FLT_MAX, q, wk);
                } else if (need) {
                    quad_traverse<false, COUNT>(S, m.root, true, ray, FLT_MAX, q, qs, wk);
                }
                if (need) {
                    if (q.face != 0xFFFFFFFFu) {
                        const float4* np = reinterpret_cast<const float4*>(S.normals + q.face);
                        const V3 n0 = f4xyz(np[0]), n1 = f4xyz(np[1]), n2 = f4xyz(np[2]);
                        h.tmpdist = q.t;
                        tmpnormal = (n0 * (1 - q.u - q.v) + n1 * q.u) + n2 * q.v;
                        hit = true;   // every accepted face already satisfied dist > EPS
                    }
                }
            }
        }
        if (hit && h.tmpdist < h.dist) {
            h.dist = h.tmpdist;
            h.n = tmpnormal;
            h.cur = (int)k;
            h.mat = ob.z;
        }
        ob = ob_n; pb = pb_n;
    }
    return susp;
}

// Shadow loop (reference src/raytracer.cpp:319-325): any object with hit && tmpdist < ldist; the loop breaks at the
// first one, so later objects no longer touch tmpdist. `susp` as in closest_loop.
template <bool QUAD, bool COUNT, int PACKET = 0>
__device__ __forceinline__ bool occluded_loop(const DScene& S, const Ray& ray, bool alive, uint32_t k0, float ldist,
                                              float& tmpdist, int& susp, uint32_t* qs, Work& wk, bool coherent = false) {
    V3 tmpnormal;
    bool in = false;
    susp = -1;
    if (S.n_objects == 0) return in;
    const bool counter = !QUAD || (lane_id() & 3u) == 0;
    // (the table ends with an entry of zeros, so entry k + 1 always exists; a walking pointer: two scalar instructions
    // per object where index, clamp, shift and add were seven)
    const RT_CONSTANT u32x4* op = (const RT_CONSTANT u32x4*)(const void*)S.objects;
    u32x4 ob = op[0], pb = op[1];
    for (uint32_t k = 0; k < S.n_objects; ++k) {
        op += 2;
        const u32x4 ob_n = op[0], pb_n = op[1];
        const float4 prm = as_f4(pb);
        const bool go = alive && k >= k0 && !in && susp < 0;
        // (three separate tests of the wave-uniform type, not if / else if / else: the structurised chain carries flag
        // registers and undefined values from one branch to the next)
        if (ob.x == RT_OBJ_SPHERE) {
            RT_UNIFORM_BRANCH;
            if (COUNT) wk.spheres += go && counter;
            if (go) {
                const bool hit = sphere_isect(prm, __uint_as_float(ob.w), ray, tmpdist, tmpnormal, coherent);
                in = hit && tmpdist < ldist;
            }
        }
        if (ob.x == RT_OBJ_PLANE) {
            RT_UNIFORM_BRANCH;
            if (COUNT) wk.planes += go && counter;
            if (go) {
                const bool hit = plane_isect(prm, ray, tmpdist, tmpnormal);
                in = hit && tmpdist < ldist;
            }
        }
        if (ob.x != RT_OBJ_SPHERE && ob.x != RT_OBJ_PLANE) {
            RT_UNIFORM_BRANCH;
            const ModelRef m = load_model(S.models, ob.y);
            bool need = go;
            if (m.bb.w >= 0) {
                if (COUNT) wk.spheres += need && counter;
                need = need && sphere_gate(m.bb, ray, coherent);
            }
            if (!QUAD && !PACKET) {
                if (need) susp = (int)k;
                else if (go) tmpdist = FLT_MAX;   // Model::intersect leaves FLT_MAX in dist on a miss (primitives.cpp:368)
            } else {
                QuadHit q;
                q.face = 0xFFFFFFFFu;
                if (PACKET == 2) {
                    mesh_serve<true, COUNT>(S, m.root, need, ray, ldist, q, qs, wk);
                } else if (PACKET) {
                    if (__ballot(need)) packet_traverse<true, COUNT>(S, m.root, need, ray, ldist, q, wk, qs);
                } else if (need) {
                    quad_traverse<true, COUNT>(S, m.root, true, ray, ldist, q, qs, wk);
                }
                if (need) in = q.face != 0xFFFFFFFFu;
                if (go && !in) tmpdist = FLT_MAX;
            }
        }
        ob = ob_n; pb = pb_n;
    }
    return in;
}

// ---------------------------------------------------------------------------------------------------------
// Node records and the per-level clamp
// ---------------------------------------------------------------------------------------------------------
#define NODE_FOLD 1u     // has at least one child ray: colour is finished by the fold pass
#define NODE_VALID 2u

// pow(cosAlpha, specRefExp) of trace() (reference src/raytracer.cpp:343), evaluated in double as there. For the
// small whole exponents scene files use (8 in every shipped material) the power is formed by squaring: at most a few
// double roundings (relative error < 1e-15), which the narrowing to float that follows absorbs except when the value
// sits within that distance of a float rounding boundary (probability ~1e-8 per call; libm's and the device
// library's pow differ from each other by as much). Other exponents go to the device pow.
// GPOW = false: the scene holds integer exponents only (DScene::generic_pow, decided when the scene is created); the
// device pow is then not part of the kernel, which halves k_shade_wide's registers (119 -> 63 VGPRs).
template <bool GPOW>
__device__ __forceinline__ double spec_pow(double x, float e) {
    const int n = (int)e;
    if (!GPOW || ((float)n == e && n >= 0 && n <= RT_POW_MAX_INT)) {
        double r = 1.0, b = x;
        for (int k = n; k; k >>= 1) { if (k & 1) r *= b; b *= b; }
        return r;
    }
    return pow(x, (double)e);
}

// reference src/raytracer.cpp:399-406. refl/refr are the children's colours already scaled by kr / kt
// (raytracer.cpp:387,395), zero when the child does not exist (raytracer.cpp:384,390).
__device__ __forceinline__ V3 finish_color(V3 color, const rt_material& m, float rCoef, V3 refl, V3 refr) {
    color = color + mk(m.color[0] * m.ka, m.color[1] * m.ka, m.color[2] * m.ka);
    if (m.refractive) {
        const float t = 1 - rCoef;
        color = color + (refl * rCoef + refr * t);
    } else {
        color = color + refl;
    }
    color.x = color.x < 0 ? 0 : color.x; color.x = color.x > 1 ? 1 : color.x;
    color.y = color.y < 0 ? 0 : color.y; color.y = color.y > 1 ? 1 : color.y;
    color.z = color.z < 0 ? 0 : color.z; color.z = color.z > 1 ? 1 : color.z;
    return color;
}

__device__ __forceinline__ uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// Wave-aggregated append: one atomic per wave for all lanes with `want`; returns the lane's slot.
__device__ __forceinline__ uint32_t wave_append(uint32_t* counter, bool want) {
    const unsigned long long mask = __ballot(want);
    if (mask == 0) return 0;
    const uint32_t lane = lane_id();
    const int leader = __ffsll((long long)mask) - 1;
    uint32_t base = 0;
    if ((int)lane == leader) base = atomicAdd(counter, (uint32_t)__popcll(mask));
    base = __shfl(base, leader);
    return base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
}

// Both child kinds with ONE atomic per wave: reflection slots first, refraction slots behind them.
__device__ __forceinline__ void wave_append2(uint32_t* counter, bool a, bool b, uint32_t& slot_a, uint32_t& slot_b) {
    const unsigned long long ma = __ballot(a), mb = __ballot(b);
    slot_a = slot_b = 0;
    if ((ma | mb) == 0) return;
    const uint32_t lane = lane_id();
    const uint32_t na = (uint32_t)__popcll(ma);
    const int leader = __ffsll((long long)(ma | mb)) - 1;
    uint32_t base = 0;
    if ((int)lane == leader) base = atomicAdd(counter, na + (uint32_t)__popcll(mb));
    base = __shfl(base, leader);
    const unsigned long long below = (1ull << lane) - 1ull;
    slot_a = base + (uint32_t)__popcll(ma & below);
    slot_b = base + na + (uint32_t)__popcll(mb & below);
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// Sample index -> pixel. Pixels are numbered so that 64 consecutive ones form an 8x8 block (bit-interleaved
// inside the block): a wave's primary rays stay close together for the BVH.
// n / d and the remainder with z = rt_recip32(d) (rt_launch.h): floor(n z / 2^32) is the quotient or one less.
__device__ __forceinline__ uint32_t udivmod32(uint32_t n, uint32_t d, uint32_t z, uint32_t& r) {
    uint32_t q = __umulhi(n, z);
    r = n - q * d;
    if (r >= d) { ++q; r -= d; }
    return q;
}
// FAST: q fits 32 bits (LevelArgs::fast32, wave-uniform) - the same integers by reciprocal multiplication
template <bool FAST>
__device__ __forceinline__ bool map_pixel(const PixelMap& pm, uint64_t q, uint32_t& x, uint32_t& y, uint64_t& out_off, bool& in_buf) {
    const uint32_t per_tile = pm.tile_w * pm.tile_h;
    const uint32_t sub_x = pm.tile_w >> 3;               // 8x8 sub-blocks per tile row
    uint32_t j, r;                                       // ordinal of the owned tile, pixel within it
    if (FAST) j = udivmod32((uint32_t)q, per_tile, pm.z_per_tile, r);
    else { j = (uint32_t)(q / per_tile); r = (uint32_t)(q % per_tile); }
    const uint32_t sb = r >> 6, w = r & 63u;
    const uint32_t mx = (w & 1u) | ((w >> 1) & 2u) | ((w >> 2) & 4u);
    const uint32_t my = ((w >> 1) & 1u) | ((w >> 2) & 2u) | ((w >> 3) & 4u);
    uint32_t sbx, sby;
    if (FAST) sby = udivmod32(sb, sub_x, pm.z_sub_x, sbx);
    else { sbx = sb % sub_x; sby = sb / sub_x; }
    const uint32_t lx = sbx * 8u + mx, ly = sby * 8u + my;
    const uint32_t T = pm.part + j * pm.n_parts;         // global tile id, row-major
    uint32_t tx, ty;
    if (FAST) ty = udivmod32(T, pm.tiles_x, pm.z_tiles_x, tx);
    else { tx = T % pm.tiles_x; ty = T / pm.tiles_x; }
    x = pm.x0 + tx * pm.tile_w + lx;
    y = pm.y0 + ty * pm.tile_h + ly;
    const bool valid = j < pm.n_owned && ty < pm.tiles_y && x < pm.x0 + pm.w && y < pm.y0 + pm.h;
    in_buf = j < pm.n_owned;
    if (pm.tiled_out) out_off = (((uint64_t)j * pm.tile_h + ly) * pm.tile_w + lx) * 3u;
    else out_off = ((uint64_t)y * pm.stride_px + x) * 3u;
    return valid;
}

// ---------------------------------------------------------------------------------------------------------
// One recursion level of trace() is two stages, each in a per-lane ("wide") and a per-quad kernel:
//   extend    closest hit of every ray of the level (raytracer.cpp:292-308), Fresnel term and the child rays
//             (raytracer.cpp:347-396) appended to level+1. Needs ~half the registers of a fused kernel.
//   shade     the light loop (raytracer.cpp:311-345) for the hits of one or several levels. Nothing downstream of
//             extend waits for it, so the host runs it on other streams, next to the extend kernels of the deeper
//             levels: the dependent chain of a level is ONE traversal instead of 1 + n_lights.
// The `return red` exit (raytracer.cpp:330-332) is decided by k_shade, after the children were appended: a hit with a
// NaN normal has at most one child, the reflection ray, whose direction is all NaN; such a ray misses every primitive
// (all comparisons false) and shoots no shadow rays, so it costs one wasted background node. k_shade then clears
// the parent's NODE_FOLD flag (the child is ignored) and counts the child in `cancelled` so that the ray statistics
// equal the reference's.
//   SRC 0: level 0, rays generated from the camera     SRC 1: level >= 1, rays read from the pool
//   SRC 2: level 0, rays supplied by the caller (rt_trace)
// ---------------------------------------------------------------------------------------------------------
// Waves per SIMD the register allocator must leave room for (512 VGPRs per lane: 4 waves = 128 VGPRs each, ...). The
// traversal is a chain of dependent loads, so more resident waves pay even where a few values spill to scratch.
#ifndef RT_WPE_SQ
#define RT_WPE_SQ 5      // 96 VGPRs: a few values spill; measured with batched frames 0.571 -> 0.548 ms (bunny), 82.3 -> 78.8 ms (c4)
#endif
#ifndef RT_WPE_EQ
#define RT_WPE_EQ 5
#endif
#ifndef RT_WPE_SW
#define RT_WPE_SW 4
#endif
#ifndef RT_WPE_EW
#define RT_WPE_EW 7
#endif
#ifndef RT_WPE_ES
#define RT_WPE_ES 7      // k_extend_serve: 72 VGPRs with a few values in scratch. Measured after the packet kernels' diet (5 / 6 / 7 / 8
                         // with RT_WPE_SS 5): bunny_1080p 0.2346 / 0.2290 / 0.2271 / 0.2291 ms, field at 1080p 4.26 / 4.14 / 4.12 / 4.14,
                         // at 8K 454 / 441 / 428 / 417, 4K glass scene 28.2 / 27.7 / 28.3 / 29.1 (its deep levels are latency-bound and
                         // pay for the spills)
#endif
#ifndef RT_WPE_SS
#define RT_WPE_SS 5      // k_shade_serve: 96 VGPRs with 11 spilled (4 -> 5: bunny_1080p -1 %, field -1 %; 6: 4K glass scene +2.5 %)
#endif
#ifndef RT_WPE_PK
#define RT_WPE_PK 7      // the per-lane kernels when they carry the packet traversal (5/6: bunny_1080p 0.292 ms, 7: 0.281, 8: 0.301; 4K glass scene 43.8 / 42.4 / 43.5)
#endif
#ifndef RT_WPE_PKS
#define RT_WPE_PKS RT_WPE_PK      // ... the light-loop kernel of the two
#endif
#define RT_RUN_LOG2 3u    // see group_of: runs of 8 groups (measured: 1, 2, 4 and 16 within 3 %; 64+ and one strip per CU 5-20 % slower). Experiments: RT_DEBUG_SKIP = run_log2 << 8
#define NODE_HIT 4u      // has a hit record: k_shade computes its colour
#define NODE_PARKED 8u   // the ray was suspended by k_extend_wide and finished by k_extend_quad: its hit is shaded by
                         // k_shade_quad<2>, not by k_shade_wide, which may run before k_extend_quad is done
#define NODE_SPARKED 16u // the hit's light loop was suspended by k_shade_wide (level 0: the mark k_shade_quad<1> looks for)

// ---- shards (rt_launch.h): every kernel walks its items shard by shard, in groups that lie in ONE shard, so that a
// wave's appends go to that shard's counters ----
__device__ __forceinline__ uint32_t lane_read(uint32_t v, uint32_t lane) {      // lane: wave-uniform
    return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)__builtin_amdgcn_readfirstlane((int)lane));
}

// Lane s (mod RT_SHARDS): what sub-pool s holds of levels lo..hi (lo >= 1). lo = hi = 0 gives the empty view at the
// start of the sub-pool (where level 1 is appended). 64-bit sums: after an overflow the counters hold the ATTEMPTED
// appends (flagged by the producing level).
struct ShardView { uint32_t start, cnt; };
__device__ __forceinline__ ShardView shard_view(const LevelArgs& A, int lo, int hi) {
    const uint32_t s = lane_id() & (RT_SHARDS - 1u);
    const ShardCounters* c = A.sc + s;
    uint64_t b = 0, e = 0;
    for (int l = 1; l <= hi; ++l) {
        if (l == lo) b = e;
        e += c->level[l];
    }
    if (b > A.sub_cap) b = A.sub_cap;
    if (e > A.sub_cap) e = A.sub_cap;
    ShardView v;
    v.start = A.n0 + s * A.sub_cap + (uint32_t)b;
    v.cnt = (uint32_t)(e - b);
    return v;
}
// Level 0 has no queues: a suspended entry is MARKED (NODE_PARKED / NODE_SPARKED in node_b.w) and the quad kernels pick
// the marked entries of each 64-entry block in sample order. (Queues filled by atomics hold the entries in the order
// the waves happened to finish; with a scene far larger than the L2 - 10 M triangles - traversing the primary rays
// in that order took 1.6x as long as in sample order.) Group `pos` = block (pos & ~63), run (pos & 63) / R of R marked
// entries. `mask` = the block's marks (lane l = entry block + l). Returns whether this quad has an entry.
__device__ __forceinline__ uint32_t select_bit(unsigned long long mask, uint32_t q) {     // index of the q-th set bit
    uint32_t lo = (uint32_t)mask, pos = 0;
    uint32_t c = (uint32_t)__popc(lo);
    if (q >= c) { q -= c; pos = 32; lo = (uint32_t)(mask >> 32); }
#pragma unroll
    for (uint32_t w = 16; w >= 1; w >>= 1) {
        c = (uint32_t)__popc(lo & ((1u << w) - 1u));
        if (q >= c) { q -= c; pos += w; lo >>= w; }
    }
    return pos;
}
__device__ __forceinline__ bool pick_marked(unsigned long long mask, uint32_t pos, uint32_t R, uint32_t quad, uint32_t& i) {
    const uint32_t rank = (pos & 63u) + quad;
    const bool have = quad < R && rank < (uint32_t)__popcll(mask);
    i = (pos & ~63u) + (have ? select_bit(mask, rank) : 0u);
    return have;
}

__device__ __forceinline__ void flush_work(unsigned long long* work, uint32_t kernel, const Work& wk) {
    unsigned long long* out = work + ((blockIdx.x & (RT_WORK_SHARDS - 1)) * RT_N_KINDS + kernel) * RT_WORK_WORDS;   // sharded like shadow_count
    const uint32_t a = wave_sum(wk.nodes), b = wave_sum(wk.tris), c = wave_sum(wk.spheres), d = wave_sum(wk.planes);
    const uint32_t e = wave_sum(wk.wave_nodes), f = wave_sum(wk.wave_tris);
    if (lane_id() == 0) {
        const unsigned long long life = wall_clock64() - wk.t_start;
        atomicAdd(out + 12, (unsigned long long)(clock64() - wk.c_start));      // shader cycles of the same span: the clock this launch ran at
        atomicAdd(out + 6, wk.trav_ticks);
        atomicMax(out + 7, wk.trav_ticks);
        atomicAdd(out + 8, (unsigned long long)wk.batches);
        atomicAdd(out + 9, wk.batches ? 1ull : 0ull);
        atomicAdd(out + 10, life);
        atomicMax(out + 11, life);
        atomicAdd(out + 0, (unsigned long long)a);
        atomicAdd(out + 1, (unsigned long long)b);
        atomicAdd(out + 2, (unsigned long long)c);
        atomicAdd(out + 3, (unsigned long long)d);
        atomicAdd(out + 4, (unsigned long long)e);
        atomicAdd(out + 5, (unsigned long long)f);
    }
}

// The ray of pool entry i (level range starts at `begin`). Returns false for a padding sample outside the image.
template <int SRC>
__device__ __forceinline__ bool fetch_ray(const LevelArgs& A, uint32_t begin, uint32_t i, Ray& ray) {
    if (SRC == 0) {
        uint32_t frame = 0, x, y, k;
        uint64_t off; bool inb;
        if (A.fast32) {             // (wave-uniform) every index of the pass fits 32 bits: divisions by reciprocal
            uint32_t g = (uint32_t)A.sample0 + (i - begin), rem;
            if (A.frame_samples) {  // batch: the frame is the same for the whole 64-entry block (frame_samples % 64 == 0)
                frame = (uint32_t)__builtin_amdgcn_readfirstlane((int)udivmod32((uint32_t)A.sample0 + ((i - begin) & ~63u), A.frame_samples, A.z_frame_samples, rem));
                g -= frame * A.frame_samples;
            }
            const uint32_t px = udivmod32(g, A.spp, A.z_spp, k);
            if (!map_pixel<true>(A.pm, px, x, y, off, inb)) return false;
        } else {
            uint64_t g = A.sample0 + (i - begin);
            if (A.frame_samples) {
                frame = (uint32_t)__builtin_amdgcn_readfirstlane((int)((A.sample0 + ((i - begin) & ~63u)) / A.frame_samples));
                g -= (uint64_t)frame * A.frame_samples;
            }
            if (!map_pixel<false>(A.pm, g / A.spp, x, y, off, inb)) return false;
            k = (uint32_t)(g % A.spp);
        }
        DCamera cam;                                                      // wave-uniform: scalar loads
        {
            const RT_CONSTANT uint32_t* src = (const RT_CONSTANT uint32_t*)(const void*)(&A.fp[frame].cam);
            uint32_t* dst = reinterpret_cast<uint32_t*>(&cam);
#pragma unroll
            for (int k = 0; k < (int)(sizeof(DCamera) / 4); ++k) dst[k] = src[k];
        }
        ray = cast_ray(cam, A.aa, x, y, k);
    } else if (SRC == 1) {
        const float4 o = pool_ld(A.rays + 2 * (size_t)(i - A.n0)), d = pool_ld(A.rays + 2 * (size_t)(i - A.n0) + 1);
        ray.o = f4xyz(o); ray.d = f4xyz(d);
    } else {
        const float* p = A.user_rays + 6 * (size_t)i;
        ray.o = ld3(p); ray.d = ld3(p + 3);
    }
    return true;
}

// What trace() does with the closest hit apart from the light loop: Fresnel term and child rays
// (reference src/raytracer.cpp:302-308, 347-396), the node record and the hit record for k_shade.
// active: this lane / quad holds pool entry i; valid: the entry has a ray; writer: the lane that stores (all lanes in
// the per-lane kernels, lane 0 of the quad in the quad kernels). Called with the whole wave converged.
// level: the entry's recursion level (A.level in the level-synchronous kernels; per quad in k_tail). kids (or null):
// the child rays and their pool entries, for a caller that goes on with them itself.
// node_a.w of an entry whose colour is FINAL (a miss, a hit without children, a folded node): a NaN no arithmetic produces.
// node_color - and with it k_resolve, which reads every level-0 entry of the frame - then needs no look at node_b: 32 -> 16 B
// per entry for the four pixels in five whose primary ray has no children. (Entries that still wait for their children keep
// rCoef there, suspended light loops their tmpdist.)
#ifndef RT_FINAL_MARK
#define RT_FINAL_MARK 1      // (0: A/B builds)
#endif
#define RT_FINAL_W_BITS 0x7FFFA5A5u
__device__ __forceinline__ float final_w(float otherwise) { return RT_FINAL_MARK ? __uint_as_float(RT_FINAL_W_BITS) : otherwise; }
struct Children { bool refl, refr; uint32_t i_refl, i_refr; Ray r_refl, r_refr; };
struct FinishOut { uint4 rec_b; V3 hitPt; float rCoef; };      // what extend_finish stored for the entry (k_trace_wide shades it at once)
#define NODE_LEVEL_SHIFT 8u      // node_b.w bits 8..11: the entry's recursion level (the `return red` exit un-counts children by level)
__device__ __forceinline__ void extend_finish(const DScene& S, const LevelArgs& A, uint32_t i, bool active, bool valid,
                                              bool writer, const Ray& ray, const HitState& h, uint32_t next_base,
                                              uint32_t next_end, uint32_t* next_count, uint32_t extra_flags,
                                              int level, Children* kids = nullptr, FinishOut* fo = nullptr) {
    uint4 rec_b = make_uint4(RT_NO_CHILD, RT_NO_CHILD, 0, 0);
    bool want_refl = false, want_refr = false;
    Ray r_refl, r_refr;
    r_refl.o = r_refl.d = r_refr.o = r_refr.d = mk(0, 0, 0);
    V3 hitPt = mk(0, 0, 0);
    float rCoef = 0;
    const int cur = (RT_DBG(A.debug) & 4u) ? -1 : h.cur;
    if (valid) {
        rec_b.w = NODE_VALID;
        if (cur >= 0) {
            const rt_material* mp = S.materials + h.mat;
            const float refractiveIdx = mp->refractiveIdx;
            const bool reflective = mp->reflective != 0, refractive = mp->refractive != 0;
            const float bias = 0.0001f;                                                  // raytracer.cpp:286
            hitPt = ray_point(ray, h.dist);
            const int depth = level + 1;                                                 // raytracer.cpp:347
            V3 refractNormal = h.n;
            float cosI = 0, cosT = 0, sinSqrPhit = 0, n1 = 0, n2 = 0, n = 0;
            if (refractive && depth <= A.max_depth) {                                    // raytracer.cpp:357-381
                cosI = dot(ray.d, h.n);
                if (cosI < 0.0f) { n1 = 1.0f; n2 = refractiveIdx; cosI = -cosI; }
                else { n1 = refractiveIdx; n2 = 1.0f; refractNormal = neg(refractNormal); }
                n = n1 / n2;
                sinSqrPhit = n * n * (1 - cosI * cosI);
                if (sinSqrPhit <= 1) {
                    cosT = sqrtf(1 - sinSqrPhit);
                    const float Rorth = (n1 * cosI - n2 * cosT) / (n1 * cosI + n2 * cosT);
                    const float Rpar = (n2 * cosI - n1 * cosT) / (n2 * cosI + n1 * cosT);
                    rCoef = (Rorth * Rorth + Rpar * Rpar) / 2;
                } else {
                    rCoef = 1;
                }
            }
            if (reflective && depth <= A.max_depth) {                                    // raytracer.cpp:385-388
                want_refl = true;
                r_refl.d = (neg(h.n) * 2.0f) * dot(h.n, ray.d) + ray.d;
                r_refl.o = hitPt + h.n * bias;
            }
            if (refractive && depth <= A.max_depth && sinSqrPhit <= 1) {                 // raytracer.cpp:391-396
                want_refr = true;
                r_refr.d = ray.d * n + refractNormal * (n * cosI - cosT);
                r_refr.o = hitPt - refractNormal * bias;
            }
            rec_b.z = h.mat;
            rec_b.w = NODE_VALID | NODE_HIT | ((want_refl || want_refr) ? NODE_FOLD : 0u);
        }
    }

    // append child rays to the next level of the wave's shard: one atomic per wave
    const bool any_refl = want_refl, any_refr = want_refr;      // (in every lane of a quad)
    want_refl = want_refl && writer;
    want_refr = want_refr && writer;
    uint32_t s1, s2;
    wave_append2(next_count, want_refl, want_refr, s1, s2);
    if (want_refl) {
        const uint64_t slot = (uint64_t)next_base + s1;
        if (slot < next_end) {
            pool_st(A.rays + 2 * (size_t)(slot - A.n0), make_float4(r_refl.o.x, r_refl.o.y, r_refl.o.z, 0));
            pool_st(A.rays + 2 * (size_t)(slot - A.n0) + 1, make_float4(r_refl.d.x, r_refl.d.y, r_refl.d.z, 0));
            rec_b.x = (uint32_t)slot;
        } else {
            *A.overflow = 1;
        }
    }
    if (want_refr) {
        const uint64_t slot = (uint64_t)next_base + s2;
        if (slot < next_end) {
            pool_st(A.rays + 2 * (size_t)(slot - A.n0), make_float4(r_refr.o.x, r_refr.o.y, r_refr.o.z, 0));
            pool_st(A.rays + 2 * (size_t)(slot - A.n0) + 1, make_float4(r_refr.d.x, r_refr.d.y, r_refr.d.z, 0));
            rec_b.y = (uint32_t)slot;
        } else {
            *A.overflow = 1;
        }
    }
    if (kids) {      // (a child that did not fit the pool does not exist: the pass is flagged as overflowed)
        // (the quad kernels' writer is lane 0 of the quad: its slots go to the quad's other lanes, which hold the same rays)
        kids->i_refl = dpp_u<QP_BCAST(0)>(rec_b.x); kids->refl = any_refl && kids->i_refl != RT_NO_CHILD; kids->r_refl = r_refl;
        kids->i_refr = dpp_u<QP_BCAST(0)>(rec_b.y); kids->refr = any_refr && kids->i_refr != RT_NO_CHILD; kids->r_refr = r_refr;
    }
    if (active && writer) {
        rec_b.w |= extra_flags | ((uint32_t)level << NODE_LEVEL_SHIFT);
        pool_st(A.node_b + i, rec_b);
        if (rec_b.w & NODE_HIT) {
            if (!fo) {      // (k_trace_wide runs the light loop from registers)
                pool_st(A.hit_n + i, make_float4(h.n.x, h.n.y, h.n.z, h.tmpdist));
                pool_st(A.hit_p + i, make_float4(hitPt.x, hitPt.y, hitPt.z, rCoef));
            }
        } else if (rec_b.w & NODE_VALID) {
            const u32x4 bg = ldc4(A.fp->background, 0);
            pool_st(A.node_a + i, make_float4(__uint_as_float(bg.x), __uint_as_float(bg.y), __uint_as_float(bg.z), final_w(0)));   // raytracer.cpp:302-304
        } else {
            A.node_a[i] = make_float4(0, 0, 0, final_w(0));
        }
    }
    if (fo) { fo->rec_b = rec_b; fo->hitPt = hitPt; fo->rCoef = rCoef; }
}

__device__ __forceinline__ HitState fresh_hit() {
    HitState h;
    h.dist = FLT_MAX; h.n = mk(0, 0, 0); h.cur = -1; h.mat = 0;
    h.tmpdist = 0;     // uninitialised in the reference (raytracer.cpp:285); defined as 0
    return h;
}

// Rays per wave of a quad kernel: 16, or fewer when the items would not fill the chip's wave slots anyway (a wave
// lasts as long as its longest ray; with everything resident at once, thinner waves end sooner).
__device__ __forceinline__ uint32_t quads_per_wave(uint32_t n_items, uint32_t vgrid) {
    const uint32_t slots = vgrid < 4096u ? vgrid : 4096u;
    const uint32_t per = (n_items + slots - 1) / slots;
    return per < 1u ? 1u : (per < 16u ? per : 16u);
}

// The groups (R items each, one wave's worth) a kernel's waves loop over.
struct Groups {
    uint32_t first, cnt, gend;   // lane s < RT_SHARDS: index range of shard s (pool or queue indices), inclusive group count
    uint32_t total;              // groups in all
    uint32_t R;                  // items per group
    uint32_t n_dense;            // != 0: level 0 in pool order: group g = items [g*R, g*R + R) of n_dense
    uint32_t chunk_log2;
    bool blocks;                 // level 0, marked entries: group g = the whole 64-entry block g (its runs of R one after the other)
};
template <bool QUAD>
__device__ __forceinline__ Groups make_groups(uint32_t first, uint32_t cnt, uint32_t vgrid = gridDim.x) {
    Groups G;
    const uint32_t lane = lane_id();
    if (lane >= RT_SHARDS) cnt = 0;
    G.first = first; G.cnt = cnt; G.n_dense = 0; G.chunk_log2 = 0; G.blocks = false;
    G.R = QUAD ? quads_per_wave(wave_sum(cnt), vgrid) : (uint32_t)RT_BLOCK;
    uint32_t v = (cnt + G.R - 1) / G.R;
    for (uint32_t off = 1; off < RT_SHARDS; off <<= 1) {
        const uint32_t t = __shfl_up(v, off);
        if (lane >= off) v += t;
    }
    G.gend = v;
    G.total = lane_read(v, RT_SHARDS - 1u);
    return G;
}
template <bool QUAD>
__device__ __forceinline__ Groups make_groups_dense(uint32_t n, uint32_t chunk_log2, uint32_t vgrid = gridDim.x) {
    Groups G;
    G.chunk_log2 = chunk_log2;
    G.first = G.cnt = G.gend = 0;
    G.n_dense = n;
    uint32_t R = RT_BLOCK;
    if (QUAD) { R = quads_per_wave(n, vgrid); R = 1u << (31 - __clz((int)R)); }     // a group must not straddle two shards
    G.R = R;
    G.total = (n + R - 1) / R;
    G.blocks = false;
    return G;
}
// Level 0, marked entries (pick_marked). With enough blocks to keep the chip busy a wave takes a whole block and works
// its runs off one after the other: an empty block costs one look instead of 64/R, a full one loads its marks once.
template <bool QUAD>
__device__ __forceinline__ Groups make_groups_marked(uint32_t n, uint32_t chunk_log2, uint32_t vgrid = gridDim.x) {
    Groups G = make_groups_dense<QUAD>(n, chunk_log2, vgrid);
    const uint32_t n_blocks = (n + 63u) >> 6;
    if (n_blocks >= 65536u) { G.blocks = true; G.R = 16; G.total = n_blocks; }     // (at 32 k blocks, one 1080p frame, the serial runs cost more than the looks: 0.11 -> 0.21 ms)
    return G;
}
// Workgroup -> group. Workgroups whose ids are congruent mod 256 tend to share a CU (ids go round-robin over the 8
// XCDs, then over an XCD's CUs): they get RUNS of consecutive groups (neighbouring rays, the same part of the tree,
// in the same 32-KB L1), and the runs go round the CUs so that a part of the frame with more work than the rest is
// spread over all of them. run_log2 = 31: one run per CU (a strip of the frame each).
__device__ __forceinline__ uint32_t group_of(uint32_t vb, uint32_t per_cu, uint32_t run_log2) {
    const uint32_t cu = vb & 255u, k = vb >> 8;
    if (run_log2 >= 31u) return cu * per_cu + k;
    return ((((k >> run_log2) << 8) + cu) << run_log2) + (k & ((1u << run_log2) - 1u));
}

// Group grp (wave-uniform, < G.total): its shard, the index of its first item and the end of the shard's range.
__device__ __forceinline__ void locate(const Groups& G, uint32_t grp, uint32_t& shard, uint32_t& pos, uint32_t& lim) {
    if (G.n_dense) { pos = G.blocks ? grp * 64u : grp * G.R; lim = G.n_dense; shard = rt_shard_of_level0(pos, G.chunk_log2); return; }
    shard = (uint32_t)__popcll(__ballot(lane_id() < RT_SHARDS && G.gend <= grp));
    const uint32_t prev = shard ? lane_read(G.gend, shard - 1u) : 0u;
    const uint32_t f = lane_read(G.first, shard);
    pos = f + (grp - prev) * G.R;
    lim = f + lane_read(G.cnt, shard);
}

// The kernel's LevelArgs (second argument, after DScene) read again from the kernel-argument segment through a pointer
// the compiler cannot see through. The packet kernels run at 7 waves per SIMD = 96 SGPRs, the packet loop needs most of
// them (32 for a node, the sort, the masks), and what is used only AFTER the traversal - a dozen pointers of LevelArgs -
// was kept across it in spill lanes: a v_writelane and a v_readlane each per 64 rays, on the pipe this path is short of.
// Loaded again after the traversal they cost scalar loads instead.
#ifndef RT_RELOAD_ARGS
#define RT_RELOAD_ARGS 2      // 0 never, 1 the packet kernels, 2 the quad-service kernels too
#endif
#define RT_RELOAD_FOR(PACKET) (RT_RELOAD_ARGS >= 2 ? (PACKET) != 0 : RT_RELOAD_ARGS == 1 && (PACKET) == 1)
__device__ __forceinline__ LevelArgs reload_level_args() {
#if __HIP_DEVICE_COMPILE__      // (the host pass of hipcc parses this too, and has no address spaces)
    const RT_CONSTANT char* p = (const RT_CONSTANT char*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return *(const RT_CONSTANT LevelArgs*)(p + ((sizeof(DScene) + alignof(LevelArgs) - 1) / alignof(LevelArgs)) * alignof(LevelArgs));
#else
    return LevelArgs{};
#endif
}

// k_extend_wide: one ray per lane. Rays that never reach a mesh are finished here; the others are queued.
// ORDERED (camera rays as packets only): the launch logs what each group cost (LevelArgs::cost0) and, when the previous frame
// left an order (LevelArgs::order0), hands the groups out in it - a kernel of its own, so that every other launch keeps
// the registers of the plain loop (as one kernel: 8 to 20 more spilled SGPRs in each per-lane kernel, batched frames 2-5 % slower).
template <int SRC, bool COUNT, int PACKET, bool ORDERED = false>
__device__ __forceinline__ void extend_wide_body(const DScene& S, const LevelArgs& A_, uint32_t* lds) {
    const LevelArgs& A = A_;
    const bool level0 = SRC != 1;
    const ShardView v = shard_view(A, A.level, A.level);
    const Groups G = level0 ? make_groups_dense<false>(A.n0, A.chunk_log2) : make_groups<false>(v.start, v.cnt);
    const uint32_t next_base_l = v.start + v.cnt;     // lane s: children of shard s are appended behind its level
    Work wk = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (COUNT) { wk.t_start = wall_clock64(); wk.c_start = clock64(); }
    else if (RT_LOG_PTR(A.wave_log)) wk.t_start = wall_clock64();      // diagnostics (RT_WAVE_LOG): the wave's start and end, nothing else

    // a workgroup is 1..4 independent waves (RT_WIDE_BLOCK: the dispatcher starts workgroups at a limited rate, and these
    // waves are short-lived), wave w of workgroup b takes groups b * waves + w, ...
    const uint32_t nwv = blockDim.x >> 6;
    const bool ordered = ORDERED && level0 && A.order0;      // (wave-uniform)
    const bool costs = ORDERED && level0 && A.cost0;
    const uint32_t n_w = ordered ? RT_ORDER_SLOTS(G.total) : G.total;
    for (uint32_t w = blockIdx.x * nwv + (threadIdx.x >> 6); w < n_w; w += gridDim.x * nwv) {
        // (level 0 of a small frame: the groups in descending order of last frame's cost, the dearest in parts: k_order_groups)
        const uint32_t code = ordered ? ((const RT_CONSTANT uint32_t*)(const void*)A.order0)[w] : w;
        if (ORDERED && code == RT_ORDER_NONE) continue;                                        // wave-uniform
        const uint32_t grp = ordered ? (code & 0x0FFFFFFFu) : code, parts_log2 = ordered ? ((code >> 28) & 3u) : 0u, part = ordered ? (code >> 30) : 0u;
        if (ORDERED && grp >= G.total) continue;                                               // (an order is only ever a hint: never out of range)
        const unsigned long long t_grp = costs ? wall_clock64() : 0ull;
        uint32_t shard, pos, lim;
        locate(G, grp, shard, pos, lim);
        const uint32_t i = pos + (threadIdx.x & 63u);
        const bool active = i < lim && (!ORDERED || ((threadIdx.x & 63u) >> (6u - parts_log2)) == part);
        Ray ray;
        ray.o = mk(0, 0, 0); ray.d = mk(0, 0, 0);
        bool valid;
        uint32_t dbg;
        {   // (camera, pixel map, sample pattern: used once per group, before the traversal)
            const LevelArgs Af = (RT_RELOAD_FOR(PACKET)) ? reload_level_args() : A_;
            valid = active && fetch_ray<SRC>(Af, 0, i, ray);
            dbg = RT_DBG(Af.debug);
        }
        HitState h = fresh_hit();
        int susp = -1;
        if (!(dbg & 2u)) susp = closest_loop<false, COUNT, PACKET>(S, ray, valid, 0, h, lds, wk, SRC == 0);      // camera rays: neighbours
        const bool parked = susp >= 0;
        const LevelArgs A = (RT_RELOAD_FOR(PACKET)) ? reload_level_args() : A_;
        ShardCounters* sc = A.sc + shard;
        uint32_t qslot = 0;
        if (!level0) qslot = wave_append(sc->qe + A.level, parked);
        if (parked) {
            A.node_b[i] = make_uint4(RT_NO_CHILD, RT_NO_CHILD, (uint32_t)susp, NODE_PARKED);   // .z: where to resume
            if (!level0) A.queue_e[lane_read(v.start, shard) + qslot] = make_uint2(i, (uint32_t)susp);
            A.hit_n[i] = make_float4(h.n.x, h.n.y, h.n.z, h.tmpdist);
            A.hit_p[i] = make_float4(h.dist, __int_as_float(h.cur), __uint_as_float(h.mat), 0);
        }
        extend_finish(S, A, i, active && !parked, valid && !parked, true, ray, h, lane_read(next_base_l, shard),
                      A.n0 + (shard + 1u) * A.sub_cap, sc->level + (A.level + 1), 0u, A.level);
        if (ORDERED && level0 && A.cost0 && lane_id() == 0)      // (the word of the wave's first quarter; a group that was not split: word 0)
            A.cost0[4u * grp + (part << (2u - parts_log2))] = (uint32_t)min(wall_clock64() - t_grp, 0xFFFFull);
    }
    if (COUNT) {
        const uint32_t steps = wave_sum(wk.wave_nodes + wk.wave_tris);
        const uint32_t wid = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        if (A.wave_log && lane_id() == 0 && wid < 4u * 8192u)
            A.wave_log[wid] = make_uint4((uint32_t)wk.t_start, (uint32_t)wall_clock64(), steps, (uint32_t)(clock64() - wk.c_start));
        flush_work(A.work, PACKET == 2 ? 2 : 0, wk);
    } else if (RT_LOG_PTR(A.wave_log) && lane_id() == 0) {
        const uint32_t wid = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        if (wid < 4u * 8192u) A.wave_log[wid] = make_uint4((uint32_t)wk.t_start, (uint32_t)wall_clock64(), 0u, 0u);
    }
}
template <int SRC, bool COUNT, bool PACKET, bool ORDERED = false>
__global__ __launch_bounds__(RT_WIDE_BLOCK_MAX) __attribute__((amdgpu_waves_per_eu(PACKET ? RT_WPE_PK : RT_WPE_EW))) void k_extend_wide(DScene S, LevelArgs A) {
    extend_wide_body<SRC, COUNT, PACKET ? 1 : 0, ORDERED>(S, A, nullptr);
}
// the rays of a level >= 1, one per lane, meshes by quad service
// (SRC 2: rt_trace_form's caller-supplied rays, for the parity tests)
template <int SRC, bool COUNT>
__global__ __launch_bounds__(RT_BLOCK) __attribute__((amdgpu_waves_per_eu(RT_WPE_ES))) void k_extend_serve(DScene S, LevelArgs A) {
    __shared__ uint32_t lds_stack[RT_SERVE_LDS];
    extend_wide_body<SRC, COUNT, 2>(S, A, lds_stack);
}

// k_extend_quad: one ray per quad. FRESH: all rays of the level from the start (levels with few rays: no per-lane
// kernel runs for them). !FRESH: the rays k_extend_wide queued, resumed at the mesh they stopped at.
// The body works for "virtual" workgroup vblock of vgrid single-wave workgroups (the kernel: blockIdx.x of gridDim.x; the
// caller's choice) with `lds_stack` = this wave's 16-column traversal stack. `direct`: group = virtual block instead of
// the CU-aware mapping of group_of.
template <int SRC, bool FRESH, bool COUNT>
__device__ __forceinline__ void extend_quad_body(const DScene& S, const LevelArgs& A, uint32_t* lds_stack, uint32_t vblock, uint32_t vgrid, bool direct) {
    const uint32_t quad = (threadIdx.x & 63u) >> 2, j = threadIdx.x & 3u;
    uint32_t* qs = lds_stack + quad;
    // (s_setprio(3) for these waves - the frame's chain of dependent launches, which shares the SIMDs with the light loops
    // of the levels above - measured no different: 0.684 / 0.689 against 0.679 / 0.688 ms one frame at a time)

    const bool level0 = SRC != 1;
    const ShardView v = shard_view(A, A.level, A.level);
    const uint32_t next_base_l = v.start + v.cnt;
    Groups G;
    if (FRESH) {
        G = level0 ? make_groups_dense<true>(A.n0, A.chunk_log2, vgrid) : make_groups<true>(v.start, v.cnt, vgrid);
    } else if (level0) {      // the marked entries, block by block
        G = make_groups_marked<true>(A.n0, A.chunk_log2, vgrid);
    } else {                  // the shard's queue
        const uint32_t q = A.sc[lane_id() & (RT_SHARDS - 1u)].qe[A.level];
        G = make_groups<true>(v.start, q < v.cnt ? q : v.cnt, vgrid);
    }
    Work wk = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (COUNT) { wk.t_start = wall_clock64(); wk.c_start = clock64(); }

    const uint32_t R = G.R;
    // Workgroup -> group mapping: see group_of
    const uint32_t run_log2 = (RT_DBG(A.debug) >> 8) ? ((RT_DBG(A.debug) >> 8) & 31u) : RT_RUN_LOG2;
    const uint32_t rl = run_log2 >= 31u ? 0u : run_log2;
    const uint32_t per_cu = ((((G.total + 255u) >> 8) + (1u << rl) - 1u) >> rl) << rl;
    for (uint32_t vb = vblock; vb < (direct ? G.total : per_cu * 256u); vb += vgrid) {
        const uint32_t grp = direct ? vb : group_of(vb, per_cu, run_log2);
        if (grp >= G.total) continue;
        uint32_t shard, pos0, lim;
        locate(G, grp, shard, pos0, lim);
        unsigned long long marks = 0;
        uint32_t n_runs = 1;
        if (!FRESH && level0) {
            const uint32_t il = (pos0 & ~63u) + lane_id();
            marks = __ballot(il < lim && (A.node_b[il].w & NODE_PARKED) != 0);
            const uint32_t have = (uint32_t)__popcll(marks);
            if (G.blocks) n_runs = (have + R - 1u) / R;
            else if (have <= (pos0 & 63u)) continue;                                             // wave-uniform
        }
        for (uint32_t run = 0; run < n_runs; ++run) {
        const uint32_t pos = pos0 + run * R;
        bool active = quad < R && pos + quad < lim;
        uint32_t i = pos + quad, k0 = 0;
        HitState h = fresh_hit();
        if (!FRESH && level0) {
            active = pick_marked(marks, pos, R, quad, i);
            if (active) k0 = A.node_b[i].z;
        } else if (!FRESH && active) {
            const uint2 e = A.queue_e[pos + quad];
            i = e.x; k0 = e.y;
        }
        if (!FRESH && active) {
            const float4 a = A.hit_n[i], b = A.hit_p[i];
            h.n = f4xyz(a); h.tmpdist = a.w; h.dist = b.x; h.cur = __float_as_int(b.y); h.mat = __float_as_uint(b.z);
        }
        Ray ray;
        ray.o = mk(0, 0, 0); ray.d = mk(0, 0, 0);
        const bool valid = active && fetch_ray<SRC>(A, 0, i, ray);
        if (COUNT) { wk.batches++; }
        unsigned long long c0 = 0;
        if (COUNT) c0 = wall_clock64();
        if (!(RT_DBG(A.debug) & 2u)) closest_loop<true, COUNT>(S, ray, valid, k0, h, qs, wk);
        if (COUNT) wk.trav_ticks += wall_clock64() - c0;
        extend_finish(S, A, i, active, valid, j == 0, ray, h, lane_read(next_base_l, shard), A.n0 + (shard + 1u) * A.sub_cap,
                      A.sc[shard].level + (A.level + 1), FRESH ? 0u : NODE_PARKED, A.level);
        }
    }
    if (COUNT) flush_work(A.work, 2, wk);
    if (COUNT && A.wave_log && lane_id() == 0)
        A.wave_log[vblock] = make_uint4((uint32_t)wk.t_start, (uint32_t)wall_clock64(), (uint32_t)wk.trav_ticks, (uint32_t)(clock64() - wk.c_start));
}
template <int SRC, bool FRESH, bool COUNT>
__global__ __launch_bounds__(RT_BLOCK) __attribute__((amdgpu_waves_per_eu(RT_WPE_EQ))) void k_extend_quad(DScene S, LevelArgs A) {
    __shared__ uint32_t lds_stack[RT_QSTACK * 16];
    extend_quad_body<SRC, FRESH, COUNT>(S, A, lds_stack, blockIdx.x, gridDim.x, false);
}

// Light loop of trace() (reference src/raytracer.cpp:311-345) for one hit, starting at light li0 / object k0 of its
// shadow loop (0/0, or where the per-lane kernel suspended). Writes node_a[i]; on suspension (QUAD = false) node_a[i]
// holds {colour so far, tmpdist} and `parked` gets (li << 16) | k. Convergent.
template <bool QUAD, bool COUNT, bool GPOW, int PACKET = 0>
__device__ __forceinline__ void shade_ray(const DScene& S, const LevelArgs& A_, uint32_t i, uint4 b, bool hit, bool writer,
                                          V3 hitNormal, V3 hitPt, float rCoef, float tmpdist, uint32_t li0, uint32_t k0,
                                          bool resumed, V3 color, uint32_t* qs, Work& wk, uint32_t& n_shadow, int& parked, bool coherent = false) {
    // the material is fetched where it is used (after the shadow test) rather than held in registers across the
    // traversal: the quad kernels are register-bound
    const rt_material* mp = S.materials + b.z;
    const float bias = 0.0001f;                                                          // raytracer.cpp:286
    bool red = false;
    parked = -1;
    const uint32_t dbg = RT_DBG(A_.debug);
    // the loop runs for the whole wave; `live` says whether this lane is still in ITS light loop
    for (uint32_t li = 0; li < S.n_lights; ++li) {                                       // raytracer.cpp:311-345
        const bool live = hit && !red && parked < 0 && li >= li0;
        const float4 lp = as_f4(ldc4(S.lights, 2 * li)), lc = as_f4(ldc4(S.lights, 2 * li + 1));
        V3 lv = f4xyz(lp) - hitPt;
        const float ldist = length(lv);
        lv = normalize(lv);
        Ray sr;
        sr.o = hitPt + hitNormal * bias;
        sr.d = lv;
        n_shadow += live && writer && !(resumed && li == li0);      // a resumed ray was counted when it was shot
        int sk;
        const bool occ = occluded_loop<QUAD, COUNT, PACKET>(S, sr, live && !(dbg & 1u), li == li0 ? k0 : 0u, ldist, tmpdist, sk, qs, wk, coherent);
        if (live && sk >= 0) { parked = (int)((li << 16) | (uint32_t)sk); continue; }
        if (!live || occ) continue;
        if (isnan3(hitNormal)) { red = true; continue; }                                 // raytracer.cpp:330-332
        const float4 mc = *reinterpret_cast<const float4*>(mp);            // color[3], ka
        const float4 mk4 = *(reinterpret_cast<const float4*>(mp) + 1);      // kd, ks, kr, kt
        const float specRefExp = mp->specRefExp;
        const float cosPhi = max_std(dot(hitNormal, lv), 0.0f);
        const float fd = cosPhi / (ldist * ldist);
        color = color + mk(((mc.x * lc.x) * mk4.x) * fd, ((mc.y * lc.y) * mk4.x) * fd, ((mc.z * lc.z) * mk4.x) * fd);
        const V3 reflectRay = (hitNormal * 2.0f) * dot(hitNormal, lv) - lv;
        const float cosAlpha = max_std(dot(reflectRay, lv), 0.0f);
        const float fs = (float)(spec_pow<GPOW>((double)cosAlpha, specRefExp) / (double)(ldist * ldist));
        color = color + mk((lc.x * mk4.y) * fs, (lc.y * mk4.y) * fs, (lc.z * mk4.y) * fs);
    }
    if (!hit || !writer) return;
    // (packet kernels: what the stores below need is read again rather than kept in spill lanes across the light loop)
    const LevelArgs A = (!QUAD && RT_RELOAD_FOR(PACKET)) ? reload_level_args() : A_;
    if (parked >= 0) {
        A.node_a[i] = make_float4(color.x, color.y, color.z, tmpdist);
    } else if (red) {
        A.node_a[i] = make_float4(1, 0, 0, final_w(0));
        if (b.w & NODE_FOLD) {      // children were appended speculatively: ignore them, and un-count them
            A.node_b[i].w = b.w & ~NODE_FOLD;      // (the marks the level-0 quad kernels select by stay as they are)
            const int lv = (int)((b.w >> NODE_LEVEL_SHIFT) & 15u);      // the level of entry i
            atomicAdd(A.cancelled + lv + 1, (b.x != RT_NO_CHILD ? 1u : 0u) + (b.y != RT_NO_CHILD ? 1u : 0u));
        }
    } else if (b.w & NODE_FOLD) {
        pool_st(A.node_a + i, make_float4(color.x, color.y, color.z, rCoef));
    } else {
        const rt_material m = *mp;
        const V3 c = finish_color(color, m, rCoef, mk(0, 0, 0), mk(0, 0, 0));
        pool_st(A.node_a + i, make_float4(c.x, c.y, c.z, final_w(rCoef)));
    }
}

__device__ __forceinline__ void flush_shadow(const LevelArgs& A, uint32_t n_shadow) {
    // statistics: one atomic per wave, spread over RT_STAT_SHARDS words
    const uint32_t ns = wave_sum(n_shadow);
    if (lane_id() == 0 && ns) atomicAdd(A.shadow_count + (blockIdx.x & (RT_STAT_SHARDS - 1)), (unsigned long long)ns);
}

// k_shade_wide: light loop, one hit per lane, for the hits of levels A.level..A.level_hi. Hits whose shadow rays never
// reach a mesh are finished here; the others are queued (queue of A.level) at the point they stopped.
// LEVEL0: 0 any level(s); 1 the launch is over level 0 and the kernel knows it (no range of levels, no queue, no
// permutation: 6 SGPR spills and 6 VGPRs less - bunny_1080p in batches 0.2111 -> 0.2094 ms, shadows-only bunny 0.135 -> 0.133,
// field at 1080p 3.94 -> 3.90, 4K glass scene 27.3 -> 26.6; without packets 54 instead of 65 VGPRs, i.e. 8 waves per SIMD:
// analytic 1080p 4 spp 0.361 -> 0.330); 2 level 0, expensive groups first (ORDERED: see extend_wide_body).
template <bool COUNT, bool GPOW, int PACKET, int LEVEL0 = 0>
__device__ __forceinline__ void shade_wide_body(const DScene& S, const LevelArgs& A_, uint32_t* lds) {
    const LevelArgs& A = A_;
    constexpr bool ORDERED = LEVEL0 == 2;
    const bool level0 = LEVEL0 != 0 || A.level == 0;      // then level_hi == 0: level 0 is not part of a range
    const ShardView v = shard_view(A, A.level, A.level_hi);
    const Groups G = level0 ? make_groups_dense<false>(A.n0, A.chunk_log2) : make_groups<false>(v.start, v.cnt);
    Work wk = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (COUNT) { wk.t_start = wall_clock64(); wk.c_start = clock64(); }
    else if (RT_LOG_PTR(A.wave_log)) wk.t_start = wall_clock64();
    uint32_t n_shadow = 0;
    const uint32_t nwv = blockDim.x >> 6;      // see extend_wide_body
    // Sorted light loops (packets over levels >= 1, A.perm): the first n_perm groups take their 64 hits from the
    // permutation - neighbours in space, whose shadow rays towards a light travel together like those of camera rays' hits -
    // instead of 64 consecutive pool entries, which after a bounce lie all over the scene. The permutation covers the
    // first n_perm groups of the pool order (hits only); groups behind it (a level that grew past the host's estimate)
    // are walked in pool order as ever.
    const uint32_t n_perm = (PACKET == 1 && !level0 && A.perm) ? (A.perm_n >> 6) : 0u;
    const uint32_t n_groups = G.total > n_perm ? G.total : n_perm;
    const bool ordered = ORDERED && A.order0;     // (wave-uniform)
    const uint32_t n_w = ordered ? RT_ORDER_SLOTS(n_groups) : n_groups;
    for (uint32_t w = blockIdx.x * nwv + (threadIdx.x >> 6); w < n_w; w += gridDim.x * nwv) {
        // (level 0 of a small frame: the groups in descending order of last frame's cost, the dearest in parts: k_order_groups)
        const uint32_t code = ordered ? ((const RT_CONSTANT uint32_t*)(const void*)A.order0)[w] : w;
        if (ORDERED && code == RT_ORDER_NONE) continue;                                        // wave-uniform
        const uint32_t grp = ordered ? (code & 0x0FFFFFFFu) : code, parts_log2 = ordered ? ((code >> 28) & 3u) : 0u, part = ordered ? (code >> 30) : 0u;
        if (ORDERED && grp >= n_groups) continue;                                              // (an order is only ever a hint: never out of range)
        const uint32_t cost_word = 4u * grp + (part << (2u - parts_log2));
        const unsigned long long t_grp = (ORDERED && A.cost0) ? wall_clock64() : 0ull;
        uint32_t shard = 0, pos = 0, lim = 0;
        uint32_t i;
        bool have;
        if (PACKET == 1 && grp < n_perm) {                                                    // wave-uniform
            const LevelArgs Af = reload_level_args();
            i = Af.perm[grp * 64u + (threadIdx.x & 63u)];
            have = i != 0xFFFFFFFFu;
        } else {
            locate(G, grp, shard, pos, lim);
            i = pos + (threadIdx.x & 63u);
            have = i < lim && (!ORDERED || ((threadIdx.x & 63u) >> (6u - parts_log2)) == part);
        }
        uint4 b = make_uint4(0, 0, 0, 0);
        float4 hn = make_float4(0, 0, 0, 0), hp = make_float4(0, 0, 0, 0);
        bool hit;
        {
            const LevelArgs Af = (RT_RELOAD_FOR(PACKET)) ? reload_level_args() : A_;
            if (have) b = pool_ld(Af.node_b + i);
            hit = (b.w & (NODE_HIT | NODE_PARKED)) == NODE_HIT;
            if (!__any(hit)) {                                                               // wave-uniform
                if (ORDERED && Af.cost0 && lane_id() == 0) Af.cost0[cost_word] = 0u;
                continue;
            }
            if (hit) { hn = pool_ld(Af.hit_n + i); hp = pool_ld(Af.hit_p + i); }
        }
        int parked;
        shade_ray<false, COUNT, GPOW, PACKET>(S, A, i, b, hit, true, f4xyz(hn), f4xyz(hp), hp.w, hn.w, 0, 0, false, mk(0, 0, 0), lds, wk,
                                n_shadow, parked, level0 || n_perm != 0u);      // the hits of camera rays / sorted hits: neighbours
        const LevelArgs A = (RT_RELOAD_FOR(PACKET)) ? reload_level_args() : A_;
        if (ORDERED && A.cost0 && lane_id() == 0) A.cost0[cost_word] = (uint32_t)min(wall_clock64() - t_grp, 0xFFFFull);
        if (level0) {       // mark; hit_n.w (tmpdist at the start of the light loop) is not needed again: keep the resume point there
            if (parked >= 0) { A.node_b[i].w = b.w | NODE_SPARKED; A.hit_n[i].w = __uint_as_float((uint32_t)parked); }
        } else {
            const uint32_t qslot = wave_append(A.sc[shard].qs + A.level, parked >= 0);
            if (parked >= 0) A.queue_s[lane_read(v.start, shard) + qslot] = make_uint2(i, (uint32_t)parked);
        }
    }
    flush_shadow(A, n_shadow);
    if (COUNT) {
        const uint32_t steps = wave_sum(wk.wave_nodes + wk.wave_tris);
        const uint32_t wid = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        if (A.wave_log && lane_id() == 0 && wid < 4u * 8192u)
            A.wave_log[wid] = make_uint4((uint32_t)wk.t_start, (uint32_t)wall_clock64(), steps, (uint32_t)(clock64() - wk.c_start));
        flush_work(A.work, PACKET == 2 ? 3 : 1, wk);
    } else if (RT_LOG_PTR(A.wave_log) && lane_id() == 0) {
        const uint32_t wid = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        if (wid < 4u * 8192u) A.wave_log[wid] = make_uint4((uint32_t)wk.t_start, (uint32_t)wall_clock64(), 0u, 0u);
    }
}
template <bool COUNT, bool GPOW, bool PACKET, int LEVEL0 = 0>
__global__ __launch_bounds__(RT_WIDE_BLOCK_MAX) __attribute__((amdgpu_waves_per_eu(PACKET ? RT_WPE_PKS : RT_WPE_SW))) void k_shade_wide(DScene S, LevelArgs A) {
    extern __shared__ uint32_t pk_lds[];      // packet_traverse: the rays of a compacted leaf, RT_PK_LDS_WORDS per wave (dynamic: sized for the launch's waves)
    shade_wide_body<COUNT, GPOW, PACKET ? 1 : 0, LEVEL0>(S, A, PACKET ? pk_lds + (threadIdx.x >> 6) * RT_PK_LDS_WORDS : nullptr);
}
// k_trace_wide: one level of a scene WITHOUT meshes in one launch. Such a scene never suspends a ray, so nothing separates
// the closest hit from its light loop: closest hit, Fresnel and child rays as k_extend_wide (extend_finish), then the light
// loop of the hit as k_shade_wide (shade_ray) with the hit still in registers - the hit record (32 B written, 48 B read per
// sample, the node record read again) never goes through memory, and the level is one launch instead of two on two
// streams. The host launches it only for scenes that hold no Model (rt_api.cpp: option "fuse").
// Measured (analytic scene, 1080p, 4 spp, shadows only, 8 frames per call): see DESIGN.md section 4.
#ifndef RT_WPE_TW
#define RT_WPE_TW 7
#endif
// PACKET (level 0 only): a scene WITH meshes none of whose materials bounces - level 0 is the whole frame, and as packets
// level 0 suspends nothing either (closest hits and shadow rays traverse the mesh in the per-lane kernel itself).
// ORDERED (with PACKET): expensive groups first as in extend_wide_body - a group's cost is the whole of its wave here.
template <int SRC, bool GPOW, bool PACKET = false, bool ORDERED = false>
__global__ __launch_bounds__(RT_WIDE_BLOCK_MAX) __attribute__((amdgpu_waves_per_eu(GPOW ? 4 : RT_WPE_TW))) void k_trace_wide(DScene S, LevelArgs A) {
    extern __shared__ uint32_t pk_lds[];      // (packet_traverse: see k_shade_wide)
    uint32_t* lds = PACKET ? pk_lds + (threadIdx.x >> 6) * RT_PK_LDS_WORDS : nullptr;
    const bool level0 = SRC != 1;
    const ShardView v = shard_view(A, A.level, A.level);
    const Groups G = level0 ? make_groups_dense<false>(A.n0, A.chunk_log2) : make_groups<false>(v.start, v.cnt);
    const uint32_t next_base_l = v.start + v.cnt;     // lane s: children of shard s are appended behind its level
    Work wk = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t n_shadow = 0;
    const uint32_t nwv = blockDim.x >> 6;      // see extend_wide_body
    const bool ordered = ORDERED && A.order0;      // (wave-uniform)
    const uint32_t n_w = ordered ? RT_ORDER_SLOTS(G.total) : G.total;
    for (uint32_t w = blockIdx.x * nwv + (threadIdx.x >> 6); w < n_w; w += gridDim.x * nwv) {
        const uint32_t code = ordered ? ((const RT_CONSTANT uint32_t*)(const void*)A.order0)[w] : w;
        if (ORDERED && code == RT_ORDER_NONE) continue;                                        // wave-uniform
        const uint32_t grp = ordered ? (code & 0x0FFFFFFFu) : code, parts_log2 = ordered ? ((code >> 28) & 3u) : 0u, part = ordered ? (code >> 30) : 0u;
        if (ORDERED && grp >= G.total) continue;                                               // (an order is only ever a hint: never out of range)
        const unsigned long long t_grp = (ORDERED && A.cost0) ? wall_clock64() : 0ull;
        uint32_t shard, pos, lim;
        locate(G, grp, shard, pos, lim);
        const uint32_t i = pos + (threadIdx.x & 63u);
        const bool active = i < lim && (!ORDERED || ((threadIdx.x & 63u) >> (6u - parts_log2)) == part);
        Ray ray;
        ray.o = mk(0, 0, 0); ray.d = mk(0, 0, 0);
        const bool valid = active && fetch_ray<SRC>(A, 0, i, ray);
        HitState h = fresh_hit();
        (void)closest_loop<false, false, PACKET ? 1 : 0>(S, ray, valid, 0, h, lds, wk, SRC == 0);      // (no Model in the scene, or packets: nothing suspends)
        FinishOut fo;
        extend_finish(S, A, i, active, valid, true, ray, h, lane_read(next_base_l, shard), A.n0 + (shard + 1u) * A.sub_cap,
                      A.sc[shard].level + (A.level + 1), 0u, A.level, nullptr, &fo);
        const bool hit = active && (fo.rec_b.w & NODE_HIT) != 0;
        if (__any(hit)) {                                                                   // wave-uniform
            int parked;
            shade_ray<false, false, GPOW, PACKET ? 1 : 0>(S, A, i, fo.rec_b, hit, true, h.n, fo.hitPt, fo.rCoef, h.tmpdist, 0, 0, false, mk(0, 0, 0), lds, wk,
                                                          n_shadow, parked, SRC == 0);
        }
        if (ORDERED && A.cost0 && lane_id() == 0)      // (the word of the wave's first quarter; a group that was not split: word 0)
            A.cost0[4u * grp + (part << (2u - parts_log2))] = (uint32_t)min(wall_clock64() - t_grp, 0xFFFFull);
    }
    flush_shadow(A, n_shadow);
}

// the hits of levels >= 1, one per lane, shadow rays through meshes by quad service
template <bool COUNT, bool GPOW>
__global__ __launch_bounds__(RT_BLOCK) __attribute__((amdgpu_waves_per_eu(RT_WPE_SS))) void k_shade_serve(DScene S, LevelArgs A) {
    __shared__ uint32_t lds_stack[RT_SERVE_LDS];
    shade_wide_body<COUNT, GPOW, 2>(S, A, lds_stack);      // (an instantiation that knows it is never level 0: no spills, measured no different)
}

// k_shade_quad: light loop, one hit per quad.
//   MODE 0  all hits of levels A.level..A.level_hi, from the start
//   MODE 1  the hits k_shade_wide queued (queue_s of A.level), resumed where they stopped
//   MODE 2  the hits of the rays that went through the extend queue (queue_e of A.level), from the start
template <int MODE, bool COUNT, bool GPOW>
__device__ __forceinline__ void shade_quad_body(const DScene& S, const LevelArgs& A, uint32_t* lds_stack, uint32_t vblock, uint32_t vgrid, bool direct) {
    constexpr bool FRESH = MODE != 1;
    const uint32_t quad = (threadIdx.x & 63u) >> 2, j = threadIdx.x & 3u;
    uint32_t* qs = lds_stack + quad;

    const bool level0 = A.level == 0;                 // then level_hi == 0
    const ShardView v = shard_view(A, A.level, A.level_hi);
    Groups G;
    if (MODE != 0 && level0) {      // MODE 1, 2 at level 0: the marked entries, block by block
        G = make_groups_marked<true>(A.n0, A.chunk_log2, vgrid);
    } else if (MODE == 0) {
        G = level0 ? make_groups_dense<true>(A.n0, A.chunk_log2, vgrid) : make_groups<true>(v.start, v.cnt, vgrid);
    } else {
        const ShardCounters* c = A.sc + (lane_id() & (RT_SHARDS - 1u));
        const uint32_t q = MODE == 1 ? c->qs[A.level] : c->qe[A.level];
        G = make_groups<true>(v.start, q < v.cnt ? q : v.cnt, vgrid);
    }
    Work wk = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (COUNT) { wk.t_start = wall_clock64(); wk.c_start = clock64(); }
    uint32_t n_shadow = 0;
    const uint32_t R = G.R;
    // Workgroup -> group mapping: see group_of
    const uint32_t run_log2 = (RT_DBG(A.debug) >> 8) ? ((RT_DBG(A.debug) >> 8) & 31u) : RT_RUN_LOG2;
    const uint32_t rl = run_log2 >= 31u ? 0u : run_log2;
    const uint32_t per_cu = ((((G.total + 255u) >> 8) + (1u << rl) - 1u) >> rl) << rl;
    for (uint32_t vb = vblock; vb < (direct ? G.total : per_cu * 256u); vb += vgrid) {
        const uint32_t grp = direct ? vb : group_of(vb, per_cu, run_log2);
        if (grp >= G.total) continue;
        uint32_t shard, pos0, lim;
        locate(G, grp, shard, pos0, lim);
        unsigned long long marks = 0;
        uint32_t n_runs = 1;
        if (MODE != 0 && level0) {
            const uint32_t il = (pos0 & ~63u) + lane_id();
            const uint32_t want = MODE == 1 ? NODE_SPARKED : (NODE_HIT | NODE_PARKED);
            marks = __ballot(il < lim && (A.node_b[il].w & want) == want);
            const uint32_t have = (uint32_t)__popcll(marks);
            if (G.blocks) n_runs = (have + R - 1u) / R;
            else if (have <= (pos0 & 63u)) continue;                                             // wave-uniform
        }
        for (uint32_t run = 0; run < n_runs; ++run) {
        const uint32_t pos = pos0 + run * R;
        bool active = quad < R && pos + quad < lim;
        uint32_t i = pos + quad, li0 = 0, k0 = 0;
        if (MODE != 0 && level0) {
            active = pick_marked(marks, pos, R, quad, i);
        } else if (MODE == 1 && active) {
            const uint2 e = A.queue_s[pos + quad];
            i = e.x; li0 = e.y >> 16; k0 = e.y & 0xFFFFu;
        } else if (MODE == 2 && active) {
            i = A.queue_e[pos + quad].x;
        }
        uint4 b = make_uint4(0, 0, 0, 0);
        if (active) b = pool_ld(A.node_b + i);
        const bool hit = (b.w & NODE_HIT) != 0;
        if (!__any(hit)) continue;                                                           // wave-uniform
        float4 hn = make_float4(0, 0, 0, 0), hp = make_float4(0, 0, 0, 0);
        V3 color = mk(0, 0, 0);
        float tmpdist = 0;
        if (hit) {
            hn = pool_ld(A.hit_n + i); hp = pool_ld(A.hit_p + i);
            tmpdist = hn.w;
            if (!FRESH) {
                const float4 c = A.node_a[i];
                color = f4xyz(c); tmpdist = c.w;
                if (level0) { const uint32_t code = __float_as_uint(hn.w); li0 = code >> 16; k0 = code & 0xFFFFu; }
            }
        }
        if (COUNT) wk.batches++;
        int parked;
        shade_ray<true, COUNT, GPOW>(S, A, i, b, hit, j == 0, f4xyz(hn), f4xyz(hp), hp.w, tmpdist, li0, k0, !FRESH, color, qs, wk,
                               n_shadow, parked);
        }
    }
    flush_shadow(A, n_shadow);
    if (COUNT) flush_work(A.work, MODE == 2 ? 4 : 3, wk);
}
template <int MODE, bool COUNT, bool GPOW>
__global__ __launch_bounds__(RT_BLOCK) __attribute__((amdgpu_waves_per_eu(RT_WPE_SQ))) void k_shade_quad(DScene S, LevelArgs A) {
    __shared__ uint32_t lds_stack[RT_QSTACK * 16];
    shade_quad_body<MODE, COUNT, GPOW>(S, A, lds_stack, blockIdx.x, gridDim.x, false);
}

// ---- expensive groups first, the most expensive in parts -------------------------------------------------------------
// A level-0 launch over ONE small frame is mostly drain: every wave of a 1080p frame has STARTED a third of the way into
// the closest-hit launch, and the rest of it is the chip emptying while the packets that cross the mesh - ten times the
// average wave's life - finish (tools/exp_wavelog2.py: 32,400 waves, mean life 9 us, longest 89 us, last start at 39 us
// of 116). Their groups come wherever the picture has them. So every wave leaves behind how long its group took
// (LevelArgs::cost0), one workgroup orders the group numbers (256 cost buckets of 0.64 us), and the next frame's launch
// of the same kind hands the groups out in that order (LevelArgs::order0):
//   * the expensive eighth (whole buckets, the most expensive first) goes to the front, the rest follows IN POOL ORDER - the
//     long waves start first, the short ones fill in behind them, and the bulk of the launch keeps walking the pool as it
//     lies (a frame whose groups all cost the same is left alone);
//   * with the long waves first a launch is as long as its longest wave (89 us of closest hits, 139 us of light loops,
//     on a chip that has been empty for half of that), so the most expensive groups are worked on in PARTS by several
//     waves: the top 1/256 as four quarters (4x4 pixels each), up to the top 1/128 as two halves. A part's packet visits
//     what its own rays visit: a wave of 16 rays executes fewer steps than the wave of 64 did.
// Only WHICH wave works on WHICH entries, and when, changes - pool positions, shards and results do not - and a stale or
// missing order (first frame, camera cut) costs time, never pixels. The costs are cleared here for the next frame (a wave
// writes the word of its first quarter; a group's cost is the sum of its four words).
// (measured, level-0 launches of one 1080p frame, closest hits / light loops in us: quarters for the top 1/32 and halves up to
// 1/8: 74 / 155; 1/64, 1/16: 68 / 133; 1/128, 1/32: 67 / 122; 1/256, 1/64: 67 / 119; 1/256, 1/128: 67 / 117; 1/512, 1/128: 69 / 115;
// 1/1024, 1/256: 70 / 114 - a part's waves cost more in all than the whole did, so only the few that set the launch's length)
#ifndef RT_ORDER_Q_DIV
#define RT_ORDER_Q_DIV 256u
#endif
#ifndef RT_ORDER_H_DIV
#define RT_ORDER_H_DIV 128u
#endif
__global__ __launch_bounds__(1024) void k_order_groups(uint32_t* __restrict__ cost_e, uint32_t* __restrict__ cost_s,
                                                       uint32_t* __restrict__ order_e, uint32_t* __restrict__ order_s,
                                                       uint8_t* __restrict__ key_e, uint8_t* __restrict__ key_s, uint32_t n) {
    __shared__ uint32_t bins[256];
    __shared__ uint32_t scan[1024];
    __shared__ uint32_t s_q, s_h, s_thr, s_front;
    uint32_t* cost = blockIdx.x == 0 ? cost_e : cost_s;
    uint32_t* order = blockIdx.x == 0 ? order_e : order_s;
    const uint32_t t = threadIdx.x;
    // every group's key is formed ONCE and kept (n bytes): whatever the costs hold, or did while this ran, every group lands
    // in the order exactly once
    uint8_t* keys = blockIdx.x == 0 ? key_e : key_s;
    const uint4* cost4 = reinterpret_cast<const uint4*>(cost);
    auto key_of = [&](uint32_t g) { return (uint32_t)keys[g]; };      // 0 = the most expensive bucket
    // (a histogram per wave, summed afterwards: nearly all groups fall into the few cheapest buckets. The kernel takes ~70 us
    // for a 1080p frame's 32,400 groups - four walks of 32 dependent steps over arrays that other CUs wrote - on a stream
    // where nothing waits for it but the next frame's light loops, which start later than that.)
    __shared__ uint32_t wbins[16][256];
    for (uint32_t k = t; k < 16u * 256u; k += 1024u) (&wbins[0][0])[k] = 0u;
    __syncthreads();
    for (uint32_t g = t; g < n; g += 1024u) {
        const uint4 c = cost4[g];
        const uint32_t key = 255u - min((c.x + c.y + c.z + c.w) >> 6, 255u);
        keys[g] = (uint8_t)key;
        atomicAdd(&wbins[t >> 6][key], 1u);
    }
    __threadfence_block();
    __syncthreads();
    if (t < 256u) {
        uint32_t sum = 0;
        for (uint32_t w = 0; w < 16u; ++w) sum += wbins[w][t];
        bins[t] = sum;
    }
    __syncthreads();
    if (t == 0) {
        // keys < q: quarters; < h: halves; < thr: whole, in front. bins[] becomes each front bucket's first SLOT.
        uint32_t groups = 0, slots = 0, q = 0, h = 0, thr = 0;
        bool open_q = true, open_h = true, open = true;
        for (uint32_t b = 0; b < 256u; ++b) {
            const uint32_t c = bins[b];
            bins[b] = slots;
            open_q = open_q && b < 255u && groups + c <= n / RT_ORDER_Q_DIV;
            open_h = open_h && b < 255u && groups + c <= n / RT_ORDER_H_DIV;
            open = open && groups < n / 8u && groups + c <= n / 4u;
            if (open_q) { q = b + 1u; slots += 4u * c; groups += c; }
            else if (open_h) { h = b + 1u; slots += 2u * c; groups += c; }
            else if (open) { thr = b + 1u; slots += c; groups += c; }
            else open = false;
        }
        h = h < q ? q : h;
        thr = thr < h ? h : thr;
        s_q = q; s_h = h; s_thr = thr; s_front = slots;
    }
    __syncthreads();
    const uint32_t q = s_q, h = s_h, thr = s_thr, front = s_front;
    // Wave v of the 16 owns groups [v * per, v * per + per), 64 at a time (coalesced). First how many of them stay behind
    // the front, then one walk that scatters the front groups into their buckets' slots and writes the rest in pool order
    // (ranks by ballot).
    const uint32_t lane = t & 63u, wv = t >> 6;
    const uint32_t per = ((n + 15u) / 16u + 63u) & ~63u, g0 = min(n, wv * per), g1 = min(n, g0 + per);
    uint32_t mine = 0;
    for (uint32_t g = g0; g < g1; g += 64u)
        mine += (uint32_t)__popcll(__ballot(g + lane < g1 && key_of(g + lane) >= thr));
    if (lane == 0) scan[wv] = mine;
    __syncthreads();
    if (t == 0) {
        uint32_t run = front;
        for (uint32_t k = 0; k < 16u; ++k) { const uint32_t c = scan[k]; scan[k] = run; run += c; }
        scan[16] = run;
    }
    __syncthreads();
    uint32_t at = scan[wv];
    for (uint32_t g = g0; g < g1; g += 64u) {
        const bool in = g + lane < g1;
        const uint32_t key = in ? key_of(g + lane) : 255u;
        const bool sel = in && key >= thr;
        const unsigned long long m = __ballot(sel);
        if (sel) order[at + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = g + lane;
        at += (uint32_t)__popcll(m);
        if (in && key < q) {
            const uint32_t a4 = atomicAdd(&bins[key], 4u);
            for (uint32_t p = 0; p < 4u; ++p) order[a4 + p] = (g + lane) | (2u << 28) | (p << 30);
        } else if (in && key < h) {
            const uint32_t a2 = atomicAdd(&bins[key], 2u);
            for (uint32_t p = 0; p < 2u; ++p) order[a2 + p] = (g + lane) | (1u << 28) | (p << 30);
        } else if (in && key < thr) {
            order[atomicAdd(&bins[key], 1u)] = g + lane;
        }
    }
    const uint32_t used = scan[16];
    for (uint32_t k = used + t; k < RT_ORDER_SLOTS(n); k += 1024u) order[k] = RT_ORDER_NONE;
    __syncthreads();      // (every thread has read what it needs of the costs)
    uint4* cost16 = reinterpret_cast<uint4*>(cost);
    for (uint32_t k = t; k < n; k += 1024u) cost16[k] = make_uint4(0u, 0u, 0u, 0u);
}

// ---- sorted light loops: the hits of a bounce level ordered by where they lie -------------------------------------
// After a bounce the 64 consecutive pool entries a wave takes are hits all over the scene (the reflection rays of one
// pixel's samples land on different objects), and their shadow rays as ONE packet are 64 unrelated traversals behind a
// single chain of scalar fetches: the light loops of level 1 are the slowest launches of the 10 M-triangle field's frame
// (350 of 896 serial ms). Shadow rays depend on nothing but the hit, so the hits may be shaded in any order: a 24-bit
// Morton key of the hit point (axis_bits per axis over the box of the scene's meshes), a radix sort of (key, entry), and
// k_shade_wide walks the sorted entry numbers (LevelArgs::perm) - a wave's hits are then neighbours in space and their
// shadow rays towards a light travel together. Entries without a hit get no slot's worth of work (key past the valid
// range, entry 0xFFFFFFFF). Results do not depend on the order: every hit is shaded exactly once, by the same code.
__device__ __forceinline__ uint32_t spread10(uint32_t v) {      // bit k -> bit 3k (10 bits)
    v &= 0x3FFu;
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}
// axis_bits (1..10) bits per axis: the key has 3 * axis_bits bits, the all-ones key is "no hit"
__global__ __launch_bounds__(256) void k_sort_keys(DScene S, LevelArgs A, SortBox bx, uint32_t axis_bits, uint32_t* __restrict__ keys,
                                                   uint32_t* __restrict__ vals, uint32_t n_slots) {
    const ShardView v = shard_view(A, A.level, A.level_hi);
    const Groups G = make_groups<false>(v.start, v.cnt);
    const uint32_t lane = threadIdx.x & 63u, nwv = blockDim.x >> 6;
    const uint32_t none = (1u << (3u * axis_bits)) - 1u;
    const float cells = (float)(1u << axis_bits), top = cells - 1.0f, down = cells * (1.0f / 256.0f);      // (SortBox::scale is for 256 cells)
    for (uint32_t grp = blockIdx.x * nwv + (threadIdx.x >> 6); grp < (n_slots >> 6); grp += gridDim.x * nwv) {
        uint32_t key = none, e = 0xFFFFFFFFu;
        if (grp < G.total) {                                                                  // wave-uniform
            uint32_t shard, pos, lim;
            locate(G, grp, shard, pos, lim);
            if (pos + lane < lim) {
                const uint4 b = pool_ld(A.node_b + pos + lane);
                if ((b.w & (NODE_HIT | NODE_PARKED)) == NODE_HIT) {
                    const float4 p = pool_ld(A.hit_p + pos + lane);
                    // (fmaxf / fminf return the non-NaN operand: a NaN coordinate lands in cell 0)
                    const uint32_t cx = (uint32_t)fminf(fmaxf((p.x - bx.lo[0]) * bx.scale[0] * down, 0.0f), top);
                    const uint32_t cy = (uint32_t)fminf(fmaxf((p.y - bx.lo[1]) * bx.scale[1] * down, 0.0f), top);
                    const uint32_t cz = (uint32_t)fminf(fmaxf((p.z - bx.lo[2]) * bx.scale[2] * down, 0.0f), top);
                    key = spread10(cx) | (spread10(cy) << 1) | (spread10(cz) << 2);      // (cells < 2^axis_bits: 3 * axis_bits key bits)
                    if (key >= none) key = none - 1u;
                    e = pos + lane;
                }
            }
        }
        keys[grp * 64u + lane] = key;
        vals[grp * 64u + lane] = e;
    }
}

// Colour of pool node i once its children (one level deeper) are final.
__device__ __forceinline__ V3 node_color(const DScene& S, const float4* __restrict__ node_a, const uint4* __restrict__ node_b,
                                         uint32_t i) {
    const float4 a = pool_ld(node_a + i);
    if (RT_FINAL_MARK && __float_as_uint(a.w) == RT_FINAL_W_BITS) return mk(a.x, a.y, a.z);
    const uint4 b = pool_ld(node_b + i);
    if (!(b.w & NODE_FOLD)) return mk(a.x, a.y, a.z);
    const rt_material m = S.materials[b.z];
    V3 refl = mk(0, 0, 0), refr = mk(0, 0, 0);
    if (b.x != RT_NO_CHILD) { const float4 c = node_a[b.x]; refl = mk(c.x * m.kr, c.y * m.kr, c.z * m.kr); }   // kr * trace(...)
    if (b.y != RT_NO_CHILD) { const float4 c = node_a[b.y]; refr = mk(c.x * m.kt, c.y * m.kt, c.z * m.kt); }   // kt * trace(...)
    return finish_color(mk(a.x, a.y, a.z), m, a.w, refl, refr);
}

// Frame parameters by value -> device memory, and the frame's pass counters cleared (one small launch per frame
// instead of a launch and a memset; see FrameParams).
// (At most RT_FP_CHUNK frames per launch, read word by word through the kernel-argument pointer. Measured on the 16-pass
// frame of config 4, tools/exp_ab.sh: 32 frames by value, indexed with threadIdx = 2.8 KB of arguments that the compiler
// copies to private scratch: 66.8 ms/frame; 16 frames: 61.3; this form: 57.6 - the launch cost 0.3-0.6 ms per pass.)
#define RT_FP_CHUNK 8u
struct FrameParamsChunk { FrameParams f[RT_FP_CHUNK]; };
static_assert(sizeof(FrameParams) % 4 == 0 && sizeof(FrameParamsChunk) / 4 <= 256, "one word per thread of block 0");
__global__ __launch_bounds__(256) void k_set_frame_params(FrameParamsChunk p, uint32_t n_frames, FrameParams* dst, uint4* clear, uint32_t n_clear16) {
    if (blockIdx.x == 0 && threadIdx.x < n_frames * uint32_t(sizeof(FrameParams) / 4)) {
        const RT_CONSTANT uint32_t* words = (const RT_CONSTANT uint32_t*)__builtin_amdgcn_kernarg_segment_ptr();   // p is the first argument
        reinterpret_cast<uint32_t*>(dst)[threadIdx.x] = words[threadIdx.x];
    }
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n_clear16; i += gridDim.x * 256) clear[i] = make_uint4(0, 0, 0, 0);
}

// The anti-aliasing pattern by value -> device memory (when it changed): as kernel arguments it needs no copy from the
// caller's pageable memory and no staging buffer that a frame in flight might still read.
struct AAPattern { float xy[2 * RT_MAX_AA_POINTS]; };
static_assert(sizeof(AAPattern) <= 2048, "the pattern travels in the kernel-argument segment");
__global__ __launch_bounds__(256) void k_set_aa(AAPattern p, uint32_t n_floats, float* dst) {
    const RT_CONSTANT float* words = (const RT_CONSTANT float*)__builtin_amdgcn_kernarg_segment_ptr();   // p is the first argument
    for (uint32_t i = threadIdx.x; i < n_floats; i += 256) dst[i] = words[i];
}

// Fold of one level (>= 1): finishes every node whose children were traced one level deeper. Wave vwave of vwaves.
__device__ __forceinline__ void fold_body(const DScene& S, const LevelArgs& A, uint32_t vwave, uint32_t vwaves) {
    const ShardView v = shard_view(A, A.level, A.level);
    const Groups G = make_groups<false>(v.start, v.cnt);
    const uint32_t lane = threadIdx.x & 63u;
    for (uint32_t grp = vwave; grp < G.total; grp += vwaves) {
        uint32_t shard, pos, lim;
        locate(G, grp, shard, pos, lim);
        const uint32_t i = pos + lane;
        if (i >= lim) continue;
        const uint4 b = A.node_b[i];
        if (b.w & NODE_FOLD) {
            const V3 c = node_color(S, A.node_a, A.node_b, i);
            A.node_a[i] = make_float4(c.x, c.y, c.z, final_w(0));
            A.node_b[i].w = b.w & ~NODE_FOLD;
        }
    }
}
__global__ __launch_bounds__(256) void k_fold(DScene S, LevelArgs A) {
    fold_body(S, A, blockIdx.x * 4u + (threadIdx.x >> 6), gridDim.x * 4u);
}

// ---- the tail of the recursion in ONE launch, NOT level-synchronous --------------------------------------------------
// The deep levels of a frame hold a few thousand rays, then a few dozen (bunny_1080p: 4753, 3327, 636, 174, 40, 20, 17,
// 8 at levels 3..10). One closest-hit launch per level makes a chain of dependent launches, each as long as the LONGEST
// ray of its level (12-35 us) plus a launch boundary, while the chip is idle: 217 us of a 0.63 ms frame, and 9 fold
// launches behind them. (Round 2 put these levels into one launch WITH a grid barrier per level - same critical path,
// the barrier no cheaper than a launch boundary: measured slower, removed.) Here each quad takes one ray of level
// A.level (the tail's root level) and follows that ray's OWN subtree depth first: closest hit, Fresnel and children
// exactly as k_extend_quad (closest_loop + extend_finish), then it goes on with the reflection child and keeps the
// refraction child (a node has at most these two) on a per-quad stack in LDS - at most one entry per level. No barrier,
// no wave ever waits for another: a wave lasts as long as the longest TREE among its quads, not the sum of the longest
// ray of every level. Every entry of the subtrees is appended to the shard's region behind the root level and counted
// in the counter of level A.level + 1, whatever its own level (the level-synchronous layout - level k + 1 behind level k
// - cannot be kept without a barrier); its true level goes into node_b.w (NODE_LEVEL_SHIFT) and into tail_levels[] for
// the ray statistics. The light loops of all these entries are one k_shade_quad launch over "levels" A.level ..
// A.level + 1; k_fold_tree then folds each root's subtree.
#define RT_TAIL_STACK 16      // pending refraction children per quad: one per level at most
template <bool COUNT>
__global__ __launch_bounds__(RT_BLOCK) __attribute__((amdgpu_waves_per_eu(4))) void k_tail(DScene S, LevelArgs A, uint32_t* tail_levels) {
    __shared__ uint32_t lds_stack[RT_QSTACK * 16];
    __shared__ float pend[RT_TAIL_STACK][8][16];      // [depth][entry, level, o.xyz, d.xyz][quad]
    const uint32_t quad = (threadIdx.x & 63u) >> 2, j = threadIdx.x & 3u;
    uint32_t* qs = lds_stack + quad;
    const ShardView v = shard_view(A, A.level, A.level);
    const uint32_t next_base_l = v.start + v.cnt;
    const Groups G = make_groups<true>(v.start, v.cnt);
    Work wk = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (COUNT) { wk.t_start = wall_clock64(); wk.c_start = clock64(); }
    for (uint32_t grp = blockIdx.x; grp < G.total; grp += gridDim.x) {
        uint32_t shard, pos, lim;
        locate(G, grp, shard, pos, lim);
        const uint32_t next_base = lane_read(next_base_l, shard), next_end = A.n0 + (shard + 1u) * A.sub_cap;
        uint32_t* next_count = A.sc[shard].level + (A.level + 1);
        bool have = quad < G.R && pos + quad < lim;
        uint32_t i = pos + quad;
        int level = A.level;
        int sp = 0;
        Ray ray;
        ray.o = mk(0, 0, 0); ray.d = mk(0, 0, 0);
        if (have) fetch_ray<1>(A, 0, i, ray);
        while (__any(have)) {                                                                // wave-uniform
            HitState h = fresh_hit();
            closest_loop<true, COUNT>(S, ray, have, 0, h, qs, wk);
            Children kids;
            extend_finish(S, A, i, have, have, j == 0, ray, h, next_base, next_end, next_count, 0u, level, &kids);
            if (have) {
                const uint32_t n_kids = (kids.refl ? 1u : 0u) + (kids.refr ? 1u : 0u);
                if (j == 0 && n_kids) atomicAdd(tail_levels + level + 1, n_kids);
                if (kids.refl && kids.refr && sp < RT_TAIL_STACK) {      // (sp < RT_TAIL_STACK always: one pending child per level)
                    if (j == 0) {
                        pend[sp][0][quad] = __uint_as_float(kids.i_refr); pend[sp][1][quad] = __int_as_float(level + 1);
                        pend[sp][2][quad] = kids.r_refr.o.x; pend[sp][3][quad] = kids.r_refr.o.y; pend[sp][4][quad] = kids.r_refr.o.z;
                        pend[sp][5][quad] = kids.r_refr.d.x; pend[sp][6][quad] = kids.r_refr.d.y; pend[sp][7][quad] = kids.r_refr.d.z;
                    }
                    ++sp;
                }
                if (kids.refl) { i = kids.i_refl; ray = kids.r_refl; ++level; }
                else if (kids.refr) { i = kids.i_refr; ray = kids.r_refr; ++level; }
                else if (sp > 0) {      // (LDS operations of one wave complete in order: the quad's own words are there)
                    --sp;
                    i = __float_as_uint(pend[sp][0][quad]); level = __float_as_int(pend[sp][1][quad]);
                    ray.o = mk(pend[sp][2][quad], pend[sp][3][quad], pend[sp][4][quad]);
                    ray.d = mk(pend[sp][5][quad], pend[sp][6][quad], pend[sp][7][quad]);
                } else have = false;
            }
        }
    }
    if (COUNT) flush_work(A.work, 2, wk);
}

// Folds the subtree under every entry of level A.level (the tail's roots): one thread per root walks its tree depth
// first with an explicit stack (entries + which child is next) and applies node_color - the reference's expression
// (raytracer.cpp:399-406) - to a node once its children are final. The trees of a tail are chains with an occasional
// fork, a dozen nodes: one launch of a few dependent loads per node instead of a launch per level.
__global__ __launch_bounds__(64) void k_fold_tree(DScene S, LevelArgs A) {
    __shared__ uint32_t st_i[RT_MAX_LEVELS + 1][64];
    __shared__ uint32_t st_k[RT_MAX_LEVELS + 1][64];
    const ShardView v = shard_view(A, A.level, A.level);
    const Groups G = make_groups<false>(v.start, v.cnt);
    const uint32_t lane = threadIdx.x & 63u;
    for (uint32_t grp = blockIdx.x; grp < G.total; grp += gridDim.x) {
        uint32_t shard, pos, lim;
        locate(G, grp, shard, pos, lim);
        if (pos + lane >= lim) continue;
        int sp = 0;
        st_i[0][lane] = pos + lane; st_k[0][lane] = 0;
        while (sp >= 0) {
            const uint32_t i = st_i[sp][lane], k = st_k[sp][lane];
            const uint4 b = A.node_b[i];
            if (!(b.w & NODE_FOLD)) { --sp; continue; }                       // final already (or no children)
            const uint32_t child = k == 0 ? b.x : b.y;
            if (k < 2) {
                st_k[sp][lane] = k + 1;
                if (child != RT_NO_CHILD && sp < RT_MAX_LEVELS && (A.node_b[child].w & NODE_FOLD)) { ++sp; st_i[sp][lane] = child; st_k[sp][lane] = 0; }
                continue;
            }
            const V3 c = node_color(S, A.node_a, A.node_b, i);
            A.node_a[i] = make_float4(c.x, c.y, c.z, final_w(0));
            A.node_b[i].w = b.w & ~NODE_FOLD;
            --sp;
        }
    }
}

// render(): px += trace(...) / numRays per sample, then (uchar)(c*255) as B,G,R (reference src/raytracer.cpp:436-448).
// One thread per pixel of the pass; the level-0 nodes of a pixel are consecutive (spp of them, 32 B each in node_a /
// node_b). The sum over a pixel's samples is a chain of float additions in sample order, so it stays with one lane;
// the loads do not: a lane that walks its own pixel's samples touches a different cache line than its neighbours in
// every load (64 lines per wave-instruction, of which 16 B each are used, evicted from the 32-KB L1 before the lane
// comes back for the next 16 B: measured 8.5x the algorithmic bytes from HBM at 16 spp, a tenth of the 4K and 8K
// frames' time). So the wave reads its 64 pixels' samples 8 at a time with lane l on pixel l / 8, sample l % 8 of
// the chunk - every 8 lanes read one whole 128-B line of each array - and the colours change hands through LDS
// ([sample of the chunk][pixel], rows 72 words apart: conflict-free both ways).
#define RT_RESOLVE_CHUNK 8u
__global__ __launch_bounds__(256) void k_resolve(DScene S, LevelArgs A, uint8_t* __restrict__ out) {
    __shared__ float lds[4][3][RT_RESOLVE_CHUNK][72];
    const uint32_t npix = A.n0 / A.spp;
    const uint32_t p = blockIdx.x * 256 + threadIdx.x;
    if (p == 0 && A.host_overflow && *A.overflow)       // tell the host without a copy: the pass ran out of pool
        __hip_atomic_store(A.host_overflow, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const float inv = 1 / (float)A.spp;        // color.h:27-30
    V3 px = mk(0, 0, 0);
    if (A.spp == 1) {
        if (p < npix) px = node_color(S, A.node_a, A.node_b, p) * inv;
    } else {
        const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
        const uint32_t p0 = p - lane;                       // the wave's first pixel (npix is a multiple of 64)
        float (*w)[RT_RESOLVE_CHUNK][72] = lds[wave];
        for (uint32_t s0 = 0; s0 < A.spp; s0 += RT_RESOLVE_CHUNK) {      // uniform over the workgroup
            const uint32_t chunk = A.spp - s0 < RT_RESOLVE_CHUNK ? A.spp - s0 : RT_RESOLVE_CHUNK;
            const uint32_t k = lane & 7u;
#pragma unroll
            for (uint32_t j = 0; j < 8; ++j) {
                const uint32_t q = j * 8u + (lane >> 3);    // pixel of the wave
                V3 c = mk(0, 0, 0);
                if (k < chunk && p0 + q < npix) c = node_color(S, A.node_a, A.node_b, (p0 + q) * A.spp + s0 + k) * inv;
                w[0][k][q] = c.x; w[1][k][q] = c.y; w[2][k][q] = c.z;
            }
            __syncthreads();
            for (uint32_t kk = 0; kk < chunk; ++kk) px = px + mk(w[0][kk][lane], w[1][kk][lane], w[2][kk][lane]);
            __syncthreads();
        }
    }
    if (p >= npix) return;
    uint32_t x, y; uint64_t off; bool inb;
    bool valid;
    if (A.fast32) {
        uint32_t rem, q = udivmod32((uint32_t)A.sample0, A.spp, A.z_spp, rem) + p;
        if (A.frame_samples) {      // batch: frame b's buffer starts b * out_stride bytes further
            const uint32_t b = udivmod32(q, A.frame_px, A.z_frame_px, q);
            out += (uint64_t)b * A.out_stride;
        }
        valid = map_pixel<true>(A.pm, q, x, y, off, inb);
    } else {
        uint64_t q = A.sample0 / A.spp + p;
        if (A.frame_samples) {
            const uint32_t frame_px = A.frame_samples / A.spp;
            const uint32_t b = (uint32_t)(q / frame_px);
            q -= (uint64_t)b * frame_px;
            out += (uint64_t)b * A.out_stride;
        }
        valid = map_pixel<false>(A.pm, q, x, y, off, inb);
    }
    if (!valid) {
        if (A.pm.tiled_out && inb) { out[off] = 0; out[off + 1] = 0; out[off + 2] = 0; }
        return;
    }
    out[off] = (uint8_t)(px.z * 255);
    out[off + 1] = (uint8_t)(px.y * 255);
    out[off + 2] = (uint8_t)(px.x * 255);
}

// rt_trace: float colours of the level-0 nodes.
__global__ __launch_bounds__(256) void k_resolve_rgb(DScene S, LevelArgs A, float* __restrict__ rgb) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= A.n0) return;
    const V3 c = node_color(S, A.node_a, A.node_b, i);
    rgb[3 * (size_t)i] = c.x; rgb[3 * (size_t)i + 1] = c.y; rgb[3 * (size_t)i + 2] = c.z;
}

// Tile-major parts -> row-major BGR8 frame.
__global__ __launch_bounds__(256) void k_untile(UntileArgs U, const uint8_t* __restrict__ tiles, uint8_t* __restrict__ out) {
    const uint64_t p = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= (uint64_t)U.W * U.H) return;
    tiles += (uint64_t)blockIdx.y * U.in_stride;        // frame of a batch
    out += (uint64_t)blockIdx.y * U.out_stride;
    const uint32_t x = (uint32_t)(p % U.W), y = (uint32_t)(p / U.W);
    const uint32_t tx = x / U.tile_w, ty = y / U.tile_h;
    const uint32_t T = ty * U.tiles_x + tx;
    const uint32_t part = T % U.n_parts, j = T / U.n_parts;
    // tiles of parts 0..part-1 come first: part q owns ceil((n_tiles - q) / n_parts) tiles
    uint64_t before = (uint64_t)part * U.part_stride;
    if (U.part_stride == 0)
        for (uint32_t q = 0; q < part; ++q) before += (U.n_tiles - q + U.n_parts - 1) / U.n_parts;
    const uint64_t src = (((before + j) * U.tile_h + (y % U.tile_h)) * U.tile_w + (x % U.tile_w)) * 3u;
    const uint64_t dst = p * 3u;
    out[dst] = tiles[src]; out[dst + 1] = tiles[src + 1]; out[dst + 2] = tiles[src + 2];
}

// ---- unit kernels for the parity tests (same device functions) ----
__global__ __launch_bounds__(RT_BLOCK) void k_cast_rays(DCamera cam, const float* __restrict__ aa, uint32_t spp,
                                                        const uint32_t* __restrict__ xy, uint32_t n, float* __restrict__ out) {
    const uint32_t i = blockIdx.x * RT_BLOCK + threadIdx.x;
    if (i >= n * spp) return;
    const uint32_t p = i / spp, k = i % spp;
    const Ray r = cast_ray(cam, aa, xy[2 * p], xy[2 * p + 1], k);
    float* o = out + 6 * (size_t)i;
    o[0] = r.o.x; o[1] = r.o.y; o[2] = r.o.z; o[3] = r.d.x; o[4] = r.d.y; o[5] = r.d.z;
}

// One ray per quad (16 rays per workgroup), the same loops the render kernels use.
__global__ __launch_bounds__(RT_BLOCK) void k_intersect(DScene S, int object, const float* __restrict__ rays, uint32_t n,
                                                        int32_t* __restrict__ hit_obj, float* __restrict__ dist_out,
                                                        float* __restrict__ normal_out) {
    __shared__ uint32_t lds_stack[RT_QSTACK * 16];
    const uint32_t quad = threadIdx.x >> 2, j = threadIdx.x & 3u;
    const uint32_t i = blockIdx.x * 16 + quad;
    const bool alive = i < n;
    Ray ray;
    ray.o = mk(0, 0, 0); ray.d = mk(0, 0, 0);
    if (alive) { ray.o = ld3(rays + 6 * (size_t)i); ray.d = ld3(rays + 6 * (size_t)i + 3); }
    Work wk = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    HitState h = fresh_hit();
    float d = 0;
    V3 nrm = mk(0, 0, 0);
    int hit = -1;
    if (object >= 0) {
        // that object only: Primitive::intersect(ray, dist, normal) with dist = 0 on entry
        const u32x4 ob = ldc4(S.objects, 2 * object), pb = ldc4(S.objects, 2 * object + 1);
        const float4 prm = as_f4(pb);
        bool got = false;
        if (ob.x == RT_OBJ_SPHERE) { if (alive) got = sphere_isect(prm, __uint_as_float(ob.w), ray, d, nrm); }
        else if (ob.x == RT_OBJ_PLANE) { if (alive) got = plane_isect(prm, ray, d, nrm); }
        else {
            const ModelRef m = load_model(S.models, ob.y);
            bool need = alive;
            if (alive) d = FLT_MAX;
            if (m.bb.w >= 0) need = alive && sphere_gate(m.bb, ray);
            if (need) {
                QuadHit q;
                quad_traverse<false, false>(S, m.root, true, ray, FLT_MAX, q, lds_stack + quad, wk);
                if (q.face != 0xFFFFFFFFu) {
                    const float4* np = reinterpret_cast<const float4*>(S.normals + q.face);
                    d = q.t;
                    nrm = (f4xyz(np[0]) * (1 - q.u - q.v) + f4xyz(np[1]) * q.u) + f4xyz(np[2]) * q.v;
                    got = true;
                }
            }
        }
        if (got) hit = object;
    } else {
        closest_loop<true, false>(S, ray, alive, 0, h, lds_stack + quad, wk);
        hit = h.cur; d = h.dist; nrm = h.n;
    }
    if (!alive || j != 0) return;
    hit_obj[i] = hit;
    dist_out[i] = d;
    normal_out[3 * (size_t)i] = nrm.x; normal_out[3 * (size_t)i + 1] = nrm.y; normal_out[3 * (size_t)i + 2] = nrm.z;
}

__global__ __launch_bounds__(RT_BLOCK) void k_occluded(DScene S, const float* __restrict__ rays, const float* __restrict__ max_dist,
                                                       uint32_t n, uint8_t* __restrict__ occ) {
    __shared__ uint32_t lds_stack[RT_QSTACK * 16];
    const uint32_t quad = threadIdx.x >> 2, j = threadIdx.x & 3u;
    const uint32_t i = blockIdx.x * 16 + quad;
    const bool alive = i < n;
    Ray ray;
    ray.o = mk(0, 0, 0); ray.d = mk(0, 0, 0);
    float md = 0;
    if (alive) { ray.o = ld3(rays + 6 * (size_t)i); ray.d = ld3(rays + 6 * (size_t)i + 3); md = max_dist[i]; }
    Work wk = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    float tmp = 0;
    int sk;
    const bool in = occluded_loop<true, false>(S, ray, alive, 0, md, tmp, sk, lds_stack + quad, wk);
    if (alive && j == 0) occ[i] = in ? 1 : 0;
}

// The same two unit kernels with one ray per LANE, through the traversal forms the render kernels use for most of
// their rays: PACKET 1 = the wave's 64 rays walk the tree together (packet_traverse), 2 = quad service
// (serve_traverse). Rays are taken 64 at a time in the order given, so what a packet holds is the caller's choice
// (rt_intersect_form / rt_occluded_form: the parity tests feed coherent camera rays AND unrelated random rays).
template <int PACKET>
__global__ __launch_bounds__(RT_BLOCK) void k_intersect_lane(DScene S, int object, const float* __restrict__ rays, uint32_t n,
                                                             int32_t* __restrict__ hit_obj, float* __restrict__ dist_out,
                                                             float* __restrict__ normal_out) {
    __shared__ uint32_t lds_stack[RT_SERVE_LDS];
    const uint32_t i = blockIdx.x * RT_BLOCK + threadIdx.x;
    const bool alive = i < n;
    Ray ray;
    ray.o = mk(0, 0, 0); ray.d = mk(0, 0, 0);
    if (alive) { ray.o = ld3(rays + 6 * (size_t)i); ray.d = ld3(rays + 6 * (size_t)i + 3); }
    Work wk = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    HitState h = fresh_hit();
    float d = 0;
    V3 nrm = mk(0, 0, 0);
    int hit = -1;
    if (object >= 0) {
        const u32x4 ob = ldc4(S.objects, 2 * object), pb = ldc4(S.objects, 2 * object + 1);
        const float4 prm = as_f4(pb);
        bool got = false;
        if (ob.x == RT_OBJ_SPHERE) { if (alive) got = sphere_isect(prm, __uint_as_float(ob.w), ray, d, nrm); }
        else if (ob.x == RT_OBJ_PLANE) { if (alive) got = plane_isect(prm, ray, d, nrm); }
        else {
            const ModelRef m = load_model(S.models, ob.y);
            bool need = alive;
            if (alive) d = FLT_MAX;
            if (m.bb.w >= 0) need = alive && sphere_gate(m.bb, ray);
            QuadHit q;
            q.face = 0xFFFFFFFFu; q.t = 0; q.u = 0; q.v = 0;
            if (PACKET == 2) mesh_serve<false, false>(S, m.root, need, ray, FLT_MAX, q, lds_stack, wk);
            else if (__ballot(need)) packet_traverse<false, false>(S, m.root, need, ray, FLT_MAX, q, wk);
            if (need && q.face != 0xFFFFFFFFu) {
                const float4* np = reinterpret_cast<const float4*>(S.normals + q.face);
                d = q.t;
                nrm = (f4xyz(np[0]) * (1 - q.u - q.v) + f4xyz(np[1]) * q.u) + f4xyz(np[2]) * q.v;
                got = true;
            }
        }
        if (got) hit = object;
    } else {
        closest_loop<false, false, PACKET>(S, ray, alive, 0, h, lds_stack, wk);
        hit = h.cur; d = h.dist; nrm = h.n;
    }
    if (!alive) return;
    hit_obj[i] = hit;
    dist_out[i] = d;
    normal_out[3 * (size_t)i] = nrm.x; normal_out[3 * (size_t)i + 1] = nrm.y; normal_out[3 * (size_t)i + 2] = nrm.z;
}

template <int PACKET>
__global__ __launch_bounds__(RT_BLOCK) void k_occluded_lane(DScene S, const float* __restrict__ rays, const float* __restrict__ max_dist,
                                                            uint32_t n, uint8_t* __restrict__ occ) {
    __shared__ uint32_t lds_stack[RT_SERVE_LDS];
    const uint32_t i = blockIdx.x * RT_BLOCK + threadIdx.x;
    const bool alive = i < n;
    Ray ray;
    ray.o = mk(0, 0, 0); ray.d = mk(0, 0, 0);
    float md = 0;
    if (alive) { ray.o = ld3(rays + 6 * (size_t)i); ray.d = ld3(rays + 6 * (size_t)i + 3); md = max_dist[i]; }
    Work wk = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    float tmp = 0;
    int sk;
    const bool in = occluded_loop<false, false, PACKET>(S, ray, alive, 0, md, tmp, sk, lds_stack, wk);
    if (alive) occ[i] = in ? 1 : 0;
}

// ---------------------------------------------------------------------------------------------------------
// Workspace of rt_render_device_noalloc (include/raytracer_amd.h): for callers that cannot allocate, the node pool of
// one small pass lives in this file-scope array instead of hipMalloc'ed memory. Every GPU the library runs on holds
// its own copy (20.5 MiB) at its own address until the process ends. `scratch` as host code sees it is a stand-in, not
// a device address: host code only ever uses the pointer hipGetSymbolAddress returned for the device in question,
// and the kernels get it as an argument (the LevelArgs pool pointers), exactly as they get hipMalloc'ed memory.
// ---------------------------------------------------------------------------------------------------------
static __device__ __align__(256) unsigned char scratch[RT_NOALLOC_SCRATCH_BYTES];

#include <mutex>
#define RT_NOALLOC_MAX_DEVICES 64
hipError_t noalloc_scratch(int device, void** ptr) {
    static std::mutex mu;
    static void* per_device[RT_NOALLOC_MAX_DEVICES] = {};
    *ptr = nullptr;
    if (device < 0 || device >= RT_NOALLOC_MAX_DEVICES) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lock(mu);
    if (!per_device[device]) {
        int cur = -1;
        hipError_t e = hipGetDevice(&cur);
        if (e != hipSuccess) return e;
        if (cur != device) return hipErrorInvalidDevice;      // the symbol's address is per device: ask with that device current
        void* p = nullptr;
        e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return e;
        if (!p) return hipErrorInvalidSymbol;
        per_device[device] = p;
    }
    *ptr = per_device[device];
    return hipSuccess;
}

// ---------------------------------------------------------------------------------------------------------
// Launchers (called from rt_api.cpp, which is plain C++)
// ---------------------------------------------------------------------------------------------------------
extern "C++" {

// Waves per workgroup of the per-lane kernels (RT_WIDE_BLOCK in the environment: threads, for experiments).
static uint32_t wide_waves_default() {
    static const uint32_t n = [] { const char* e = std::getenv("RT_WIDE_BLOCK"); const int t = e ? std::atoi(e) : RT_WIDE_BLOCK_DEFAULT; return (uint32_t)std::max(1, std::min(RT_WIDE_BLOCK_MAX, t) / 64); }();
    return n;
}
// ... of this launch: the pass' choice (LevelArgs::wide_waves: rt_api.cpp picks 4 for the passes of frames that come alone) or
// the process default above.
static uint32_t wide_waves_of(const LevelArgs& A) {
    const uint32_t w = A.wide_waves ? A.wide_waves : wide_waves_default();
    return std::max(1u, std::min<uint32_t>(w, RT_WIDE_BLOCK_MAX / 64));
}
template <int SRC>
static void launch_extend_src(int form, bool count, const DScene& S, const LevelArgs& A, dim3 g, hipStream_t st) {
    dim3 b(RT_BLOCK);
    if (form == RT_FORM_WIDE) {
        b = dim3(64u * wide_waves_of(A));
        g = dim3((g.x + wide_waves_of(A) - 1u) / wide_waves_of(A));
        if ((SRC == 0 && S.packet) || (SRC == 1 && S.packet >= 2) || (SRC == 2 && S.user_packet)) {     // primary rays: coherent, traversed as packets in the per-lane kernel itself (2: deeper levels too; user_packet: rt_trace_form)
            if (count) hipLaunchKernelGGL((k_extend_wide<SRC, true, true>), g, b, 0, st, S, A);
            else if (SRC == 0 && A.cost0) hipLaunchKernelGGL((k_extend_wide<0, false, true, true>), g, b, 0, st, S, A);      // expensive groups first
            else hipLaunchKernelGGL((k_extend_wide<SRC, false, true>), g, b, 0, st, S, A);
        } else {
            if (count) hipLaunchKernelGGL((k_extend_wide<SRC, true, false>), g, b, 0, st, S, A);
            else hipLaunchKernelGGL((k_extend_wide<SRC, false, false>), g, b, 0, st, S, A);
        }
    } else if (form == RT_FORM_WIDE_FUSED) {
        if constexpr (SRC != 2) {       // (a scene without meshes: the level's light loops in the same launch)
            b = dim3(64u * wide_waves_of(A));
            g = dim3((g.x + wide_waves_of(A) - 1u) / wide_waves_of(A));
            if (SRC == 0 && S.packet && !S.generic_pow && A.cost0) hipLaunchKernelGGL((k_trace_wide<0, false, true, true>), g, b, RT_LEAF_COMPACT ? wide_waves_of(A) * RT_PK_LDS_WORDS * 4u : 0u, st, S, A);
            else if (SRC == 0 && S.packet && !S.generic_pow) hipLaunchKernelGGL((k_trace_wide<0, false, true>), g, b, RT_LEAF_COMPACT ? wide_waves_of(A) * RT_PK_LDS_WORDS * 4u : 0u, st, S, A);
            else if (S.generic_pow) hipLaunchKernelGGL((k_trace_wide<SRC, true>), g, b, 0, st, S, A);
            else hipLaunchKernelGGL((k_trace_wide<SRC, false>), g, b, 0, st, S, A);
        }
    } else if (form == RT_FORM_WIDE_SERVE) {
        if constexpr (SRC != 0) {       // (camera rays are never served: level 0 of a frame is packets or marks + quads)
            if (count) hipLaunchKernelGGL((k_extend_serve<SRC, true>), g, b, 0, st, S, A);
            else hipLaunchKernelGGL((k_extend_serve<SRC, false>), g, b, 0, st, S, A);
        }
    } else if (form == RT_FORM_QUAD_FRESH) {
        if (count) hipLaunchKernelGGL((k_extend_quad<SRC, true, true>), g, b, 0, st, S, A);
        else hipLaunchKernelGGL((k_extend_quad<SRC, true, false>), g, b, 0, st, S, A);
    } else {
        if (count) hipLaunchKernelGGL((k_extend_quad<SRC, false, true>), g, b, 0, st, S, A);
        else hipLaunchKernelGGL((k_extend_quad<SRC, false, false>), g, b, 0, st, S, A);
    }
}
void launch_extend(int src, int form, bool count, const DScene& S, const LevelArgs& A, uint32_t grid, hipStream_t st) {
    if (src == 0) launch_extend_src<0>(form, count, S, A, dim3(grid), st);
    else if (src == 1) launch_extend_src<1>(form, count, S, A, dim3(grid), st);
    else launch_extend_src<2>(form, count, S, A, dim3(grid), st);
}
template <bool COUNT, bool GPOW>
static void launch_shade_t(int form, const DScene& S, const LevelArgs& A, dim3 g, dim3 b, hipStream_t st) {
    if (form == RT_FORM_WIDE) {
        b = dim3(64u * wide_waves_of(A));
        g = dim3((g.x + wide_waves_of(A) - 1u) / wide_waves_of(A));
        // the hits of primary rays: their shadow rays are coherent too (rt_trace's caller-supplied rays need not be)
        const uint32_t pk_lds = RT_LEAF_COMPACT ? wide_waves_of(A) * RT_PK_LDS_WORDS * 4u : 0u;
        if (S.packet && (A.level == 0 || S.packet >= 2) && (!A.user_rays || S.user_packet)) {
            if (!COUNT && !GPOW && A.level == 0 && A.level_hi == 0 && A.cost0) hipLaunchKernelGGL((k_shade_wide<false, false, true, 2>), g, b, pk_lds, st, S, A);      // expensive groups first
            else if (!COUNT && !GPOW && A.level == 0 && A.level_hi == 0) hipLaunchKernelGGL((k_shade_wide<false, false, true, 1>), g, b, pk_lds, st, S, A);
            else hipLaunchKernelGGL((k_shade_wide<COUNT, GPOW, true>), g, b, pk_lds, st, S, A);
        }
        else if (!COUNT && !GPOW && A.level == 0 && A.level_hi == 0 && !A.user_rays) hipLaunchKernelGGL((k_shade_wide<false, false, false, 1>), g, b, 0, st, S, A);
        else hipLaunchKernelGGL((k_shade_wide<COUNT, GPOW, false>), g, b, 0, st, S, A);
    }
    else if (form == RT_FORM_WIDE_SERVE) hipLaunchKernelGGL((k_shade_serve<COUNT, GPOW>), g, b, 0, st, S, A);
    else if (form == RT_FORM_QUAD_FRESH) hipLaunchKernelGGL((k_shade_quad<0, COUNT, GPOW>), g, b, 0, st, S, A);
    else if (form == RT_FORM_QUAD_RESUME) hipLaunchKernelGGL((k_shade_quad<1, COUNT, GPOW>), g, b, 0, st, S, A);
    else hipLaunchKernelGGL((k_shade_quad<2, COUNT, GPOW>), g, b, 0, st, S, A);
}
void launch_shade(int form, bool count, const DScene& S, const LevelArgs& A, uint32_t grid, hipStream_t st) {
    dim3 g(grid), b(RT_BLOCK);
    if (S.generic_pow) { if (count) launch_shade_t<true, true>(form, S, A, g, b, st); else launch_shade_t<false, true>(form, S, A, g, b, st); }
    else { if (count) launch_shade_t<true, false>(form, S, A, g, b, st); else launch_shade_t<false, false>(form, S, A, g, b, st); }
}
void launch_order_groups(uint32_t* cost_e, uint32_t* cost_s, uint32_t* order_e, uint32_t* order_s, uint8_t* key_e, uint8_t* key_s, uint32_t n, hipStream_t st) {
    hipLaunchKernelGGL(k_order_groups, dim3(2), dim3(1024), 0, st, cost_e, cost_s, order_e, order_s, key_e, key_s, n);
}
size_t sort_hits_temp_bytes(uint32_t n_slots) {
    size_t bytes = 0;
    hipcub::DoubleBuffer<uint32_t> dk(nullptr, nullptr), dv(nullptr, nullptr);
    if (hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, dk, dv, int(n_slots), 0, 30) != hipSuccess) return 0;
    return bytes;
}
const uint32_t* launch_sort_hits(const DScene& S, const LevelArgs& A, const SortBox& box, uint32_t axis_bits, uint32_t* const keys[2],
                                 uint32_t* const vals[2], void* temp, size_t temp_bytes, uint32_t n_slots, hipStream_t st) {
    axis_bits = std::max(1u, std::min(10u, axis_bits));
    hipLaunchKernelGGL(k_sort_keys, dim3(std::min<uint32_t>(2048u, (n_slots / 64u + 3u) / 4u)), dim3(256), 0, st, S, A, box, axis_bits, keys[0], vals[0], n_slots);
    hipcub::DoubleBuffer<uint32_t> dk(keys[0], keys[1]), dv(vals[0], vals[1]);
    if (hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, dk, dv, int(n_slots), 0, int(3u * axis_bits), st) != hipSuccess) return nullptr;
    return dv.Current();
}
void launch_set_frame_params(const FrameParamsBatch& p, uint32_t n_frames, FrameParams* dst, void* clear, size_t clear_bytes, hipStream_t st) {
    const uint32_t n16 = uint32_t(clear_bytes / 16);
    const uint32_t grid = n16 ? std::min<uint32_t>((n16 + 255u) / 256u, 1024u) : 1u;
    for (uint32_t f0 = 0; f0 < n_frames; f0 += RT_FP_CHUNK) {
        FrameParamsChunk c;
        const uint32_t n = std::min(RT_FP_CHUNK, n_frames - f0);
        for (uint32_t i = 0; i < n; ++i) c.f[i] = p.f[f0 + i];
        for (uint32_t i = n; i < RT_FP_CHUNK; ++i) c.f[i] = p.f[f0];
        hipLaunchKernelGGL(k_set_frame_params, dim3(f0 ? 1u : grid), dim3(256), 0, st, c, n, dst + f0, static_cast<uint4*>(clear), f0 ? 0u : n16);
    }
}
void launch_set_aa(const float* aa_xy, uint32_t n_points, float* dst, hipStream_t st) {
    AAPattern p;
    std::memset(&p, 0, sizeof p);
    std::memcpy(p.xy, aa_xy, sizeof(float) * 2 * std::min<uint32_t>(n_points, RT_MAX_AA_POINTS));
    hipLaunchKernelGGL(k_set_aa, dim3(1), dim3(256), 0, st, p, 2u * std::min<uint32_t>(n_points, RT_MAX_AA_POINTS), dst);
}
void launch_tail(bool count, const DScene& S, const LevelArgs& A, uint32_t* tail_levels, uint32_t grid, hipStream_t st) {
    if (count) hipLaunchKernelGGL((k_tail<true>), dim3(grid), dim3(RT_BLOCK), 0, st, S, A, tail_levels);
    else hipLaunchKernelGGL((k_tail<false>), dim3(grid), dim3(RT_BLOCK), 0, st, S, A, tail_levels);
}
void launch_fold_tree(const DScene& S, const LevelArgs& A, uint32_t grid, hipStream_t st) {
    hipLaunchKernelGGL(k_fold_tree, dim3(grid), dim3(64), 0, st, S, A);
}
void launch_fold(const DScene& S, const LevelArgs& A, uint32_t grid, hipStream_t st) {
    hipLaunchKernelGGL(k_fold, dim3(grid), dim3(256), 0, st, S, A);
}
void launch_resolve(const DScene& S, const LevelArgs& A, uint8_t* out, hipStream_t st) {
    const uint32_t npix = A.n0 / A.spp;
    hipLaunchKernelGGL(k_resolve, dim3((npix + 255) / 256), dim3(256), 0, st, S, A, out);
}
void launch_resolve_rgb(const DScene& S, const LevelArgs& A, float* rgb, hipStream_t st) {
    hipLaunchKernelGGL(k_resolve_rgb, dim3((A.n0 + 255) / 256), dim3(256), 0, st, S, A, rgb);
}
void launch_untile(const UntileArgs& U, const uint8_t* tiles, uint8_t* out, hipStream_t st) {
    const uint64_t n = (uint64_t)U.W * U.H;
    hipLaunchKernelGGL(k_untile, dim3((uint32_t)((n + 255) / 256), U.n_frames ? U.n_frames : 1u), dim3(256), 0, st, U, tiles, out);
}
void launch_cast_rays(const DCamera& cam, const float* aa, uint32_t spp, const uint32_t* xy, uint32_t n, float* out,
                      hipStream_t st) {
    const uint32_t total = n * spp;
    hipLaunchKernelGGL(k_cast_rays, dim3((total + RT_BLOCK - 1) / RT_BLOCK), dim3(RT_BLOCK), 0, st, cam, aa, spp, xy, n, out);
}
void launch_intersect(const DScene& S, int form, int object, const float* rays, uint32_t n, int32_t* hit, float* dist, float* nrm,
                      hipStream_t st) {
    const dim3 gl((n + RT_BLOCK - 1) / RT_BLOCK), b(RT_BLOCK);
    if (form == 1) hipLaunchKernelGGL((k_intersect_lane<1>), gl, b, 0, st, S, object, rays, n, hit, dist, nrm);
    else if (form == 2) hipLaunchKernelGGL((k_intersect_lane<2>), gl, b, 0, st, S, object, rays, n, hit, dist, nrm);
    else hipLaunchKernelGGL(k_intersect, dim3((n + 15) / 16), b, 0, st, S, object, rays, n, hit, dist, nrm);
}
void launch_occluded(const DScene& S, int form, const float* rays, const float* max_dist, uint32_t n, uint8_t* occ, hipStream_t st) {
    const dim3 gl((n + RT_BLOCK - 1) / RT_BLOCK), b(RT_BLOCK);
    if (form == 1) hipLaunchKernelGGL((k_occluded_lane<1>), gl, b, 0, st, S, rays, max_dist, n, occ);
    else if (form == 2) hipLaunchKernelGGL((k_occluded_lane<2>), gl, b, 0, st, S, rays, max_dist, n, occ);
    else hipLaunchKernelGGL(k_occluded, dim3((n + 15) / 16), b, 0, st, S, rays, max_dist, n, occ);
}

}  // extern "C++"
