// bvh_device.hip — BVH build on the GPU: the same tree FORMAT as the host builder's
// (bvh_build.cpp: Bvh4Node / LeafBlock of rt_device.h), from a linear BVH:
//   1. centroid bounds of the model's faces, 63-bit Morton key per face (21 bits per axis)
//   2. radix sort of (key, face)                                   hipcub::DeviceRadixSort
//   3. binary radix tree over the sorted faces (Karras 2012: one thread per internal node, no synchronisation)
//   4. leaf blocks = the maximal subtrees with at most 4 faces (each a run of sorted faces, so a block follows the
//      tree instead of cutting across it), with the floats Triangle::intersect forms per call (reference
//      src/primitives.cpp:52-53) and their exact boxes; numbered by a prefix sum
//   5. boxes of the nodes above the blocks, bottom-up (the second thread to arrive at a node continues upwards)
//   6. every second level of the binary tree becomes a 4-wide node: a node's children are its grandchildren (or its
//      children where those are blocks); nodes at even depth are kept, numbered by a prefix sum
// Like the host tree, it only decides WHICH triangles get tested (see bvh_build.cpp); a Morton-order tree culls less
// well than the SAH tree (slower frames), but builds in milliseconds instead of seconds. The worst-case traversal
// stack follows from the depth of the binary tree; a model whose tree would be too deep for the traversal's LDS stack
// is reported as not applicable and the caller falls back to the host builder.
// The reference has no acceleration structure; this replaces, with bvh_build.cpp, the loop of Model::intersect
// (reference src/primitives.cpp:376-382).
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <chrono>
#include <cstdio>
#include <vector>

#include "raytracer_amd.h"
#include "rt_device.h"
#include "bvh_device.h"

void rt_set_error(const char* fmt, ...);   // rt_api.cpp

namespace {

#define NONE 0xFFFFFFFFu
#define BLOCK 256

// order-preserving float <-> uint (for atomicMin/Max on floats of either sign)
__device__ __forceinline__ uint32_t f2o(float f) { const uint32_t u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__host__ __device__ __forceinline__ float o2f(uint32_t o) {
    const uint32_t u = (o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o;
#ifdef __HIP_DEVICE_COMPILE__
    return __uint_as_float(u);
#else
    float f; memcpy(&f, &u, 4); return f;
#endif
}

struct Ctx {
    const float* v0; const float* v1; const float* v2;   // whole scene, 3 floats per face
    uint32_t first, n;       // the model's faces
    uint32_t node_base, leaf_base;
    uint32_t* bounds;        // 6 ordered-uint: centroid min xyz, max xyz
    uint64_t* key; uint32_t* face;          // sorted by key
    // binary radix tree over the n sorted faces: internal nodes 0..n-2 (0 = root), leaves = sorted positions
    uint32_t* child;         // 2 per internal node: internal index, or RT_LEAF_BIT | sorted position
    uint32_t* parent_int; uint32_t* parent_leaf;
    uint32_t* cnt; uint32_t* lo;            // per internal node: faces below it, first sorted position
    // leaf blocks = maximal subtrees with <= 4 faces (a run of sorted positions each)
    uint32_t* start_flag;    // per sorted position: a block starts here
    uint32_t* run_len; uint32_t* block_parent; uint32_t* block_ref;   // per block start: faces, parent node, own tree reference
    uint32_t* blk;           // exclusive scan of start_flag: block number
    float* leaf_box;         // 6 per block
    float* ibox;             // 6 per internal node with more than 4 faces
    uint32_t* flag;          // per internal node: arrivals
    uint32_t* mark; uint32_t* idx4;
    uint32_t* max_depth;     // depth of the deepest internal node with more than 4 faces (root = 0)
    Bvh4Node* nodes; LeafBlock* leaves;
};

__global__ void k_bounds(Ctx c) {
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (uint32_t i = blockIdx.x * BLOCK + threadIdx.x; i < c.n; i += gridDim.x * BLOCK) {
        const size_t f = 3 * size_t(c.first + i);
        for (int a = 0; a < 3; ++a) {
            const float x = c.v0[f + a], y = c.v1[f + a], z = c.v2[f + a];
            const float m = 0.5f * (fminf(x, fminf(y, z)) + fmaxf(x, fmaxf(y, z)));
            lo[a] = fminf(lo[a], m); hi[a] = fmaxf(hi[a], m);
        }
    }
    for (int a = 0; a < 3; ++a) {
        for (int off = 32; off > 0; off >>= 1) { lo[a] = fminf(lo[a], __shfl_xor(lo[a], off)); hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], off)); }
        if ((threadIdx.x & 63) == 0) { atomicMin(c.bounds + a, f2o(lo[a])); atomicMax(c.bounds + 3 + a, f2o(hi[a])); }
    }
}

__device__ __forceinline__ uint64_t spread21(uint64_t x) {      // 21 bits -> every third bit
    x &= 0x1FFFFFull;
    x = (x | (x << 32)) & 0x1F00000000FFFFull;
    x = (x | (x << 16)) & 0x1F0000FF0000FFull;
    x = (x | (x << 8)) & 0x100F00F00F00F00Full;
    x = (x | (x << 4)) & 0x10C30C30C30C30C3ull;
    x = (x | (x << 2)) & 0x1249249249249249ull;
    return x;
}

__global__ void k_morton(Ctx c) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= c.n) return;
    const size_t f = 3 * size_t(c.first + i);
    uint64_t key = 0;
    for (int a = 0; a < 3; ++a) {
        const float lo = o2f(c.bounds[a]), hi = o2f(c.bounds[3 + a]);
        const float x = c.v0[f + a], y = c.v1[f + a], z = c.v2[f + a];
        const float m = 0.5f * (fminf(x, fminf(y, z)) + fmaxf(x, fmaxf(y, z)));
        const float ext = hi - lo;
        float t = ext > 0 ? (m - lo) / ext : 0.0f;
        t = fminf(fmaxf(t, 0.0f), 1.0f);
        const uint64_t q = (uint64_t)fminf(t * 2097152.0f, 2097151.0f);
        key |= spread21(q) << (2 - a);
    }
    c.key[i] = key;
    c.face[i] = c.first + i;
}

// length of the common prefix of sorted faces i and j (keys made distinct by the position); -1 outside the range
__device__ __forceinline__ int delta(const uint64_t* key, uint32_t n, int i, int j) {
    if (j < 0 || j >= (int)n) return -1;
    const uint64_t a = key[i], b = key[j];
    if (a != b) return __clzll((long long)(a ^ b));
    return 64 + __clz((int)((uint32_t)i ^ (uint32_t)j));
}

__global__ void k_radix_tree(Ctx c) {
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    const uint32_t n = c.n;
    if (i >= (int)n - 1) return;
    const uint64_t* key = c.key;
    const int d = delta(key, n, i, i + 1) - delta(key, n, i, i - 1) >= 0 ? 1 : -1;
    const int dmin = delta(key, n, i, i - d);
    int lmax = 2;
    while (delta(key, n, i, i + lmax * d) > dmin) lmax *= 2;
    int l = 0;
    for (int t = lmax / 2; t >= 1; t /= 2)
        if (delta(key, n, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = delta(key, n, i, j);
    int s = 0, t = l;
    do {
        t = (t + 1) / 2;
        if (delta(key, n, i, i + (s + t) * d) > dnode) s += t;
    } while (t > 1);
    const int gamma = i + s * d + min(d, 0);
    const int lo = min(i, j), hi = max(i, j);
    const bool left_leaf = lo == gamma, right_leaf = hi == gamma + 1;
    c.child[2 * size_t(i)] = left_leaf ? (RT_LEAF_BIT | (uint32_t)gamma) : (uint32_t)gamma;
    c.child[2 * size_t(i) + 1] = right_leaf ? (RT_LEAF_BIT | (uint32_t)(gamma + 1)) : (uint32_t)(gamma + 1);
    if (left_leaf) c.parent_leaf[gamma] = (uint32_t)i; else c.parent_int[gamma] = (uint32_t)i;
    if (right_leaf) c.parent_leaf[gamma + 1] = (uint32_t)i; else c.parent_int[gamma + 1] = (uint32_t)i;
    if (i == 0) c.parent_int[0] = NONE;
    c.cnt[i] = (uint32_t)(hi - lo + 1);
    c.lo[i] = (uint32_t)lo;
}

// A tree reference below which at most 4 faces lie is the root of a leaf block.
__device__ __forceinline__ bool is_block(const Ctx& c, uint32_t ref) { return (ref & RT_LEAF_BIT) || c.cnt[ref] <= RT_MAX_LEAF_TRIS; }
__device__ __forceinline__ uint32_t block_start(const Ctx& c, uint32_t ref) { return (ref & RT_LEAF_BIT) ? (ref & ~RT_LEAF_BIT) : c.lo[ref]; }

// n <= 4: the whole model is one block
__global__ void k_single_block(Ctx c) {
    if (blockIdx.x == 0 && threadIdx.x == 0) { c.start_flag[0] = 1; c.run_len[0] = c.n; c.block_parent[0] = NONE; c.block_ref[0] = NONE; }
}

__global__ void k_mark_blocks(Ctx c) {
    const uint32_t t = blockIdx.x * BLOCK + threadIdx.x;
    if (t >= c.n) return;
    {   // the face at sorted position t, if its parent holds more than 4 faces
        const uint32_t p = c.parent_leaf[t];
        if (c.cnt[p] > RT_MAX_LEAF_TRIS) { c.start_flag[t] = 1; c.run_len[t] = 1; c.block_parent[t] = p; c.block_ref[t] = RT_LEAF_BIT | t; }
    }
    if (t + 1 < c.n && c.cnt[t] <= RT_MAX_LEAF_TRIS) {      // internal node t with <= 4 faces under a parent with more
        const uint32_t p = c.parent_int[t];
        if (p != NONE && c.cnt[p] > RT_MAX_LEAF_TRIS) {
            const uint32_t s = c.lo[t];
            c.start_flag[s] = 1; c.run_len[s] = c.cnt[t]; c.block_parent[s] = p; c.block_ref[s] = t;
        }
    }
}

__global__ void k_leaf_blocks(Ctx c) {
    const uint32_t s = blockIdx.x * BLOCK + threadIdx.x;
    if (s >= c.n || !c.start_flag[s]) return;
    const uint32_t j = c.blk[s], cnt = c.run_len[s];
    LeafBlock lb;
    memset(&lb, 0, sizeof lb);
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (uint32_t t = 0; t < cnt; ++t) {
        const uint32_t f = c.face[s + t];
        const float* a = c.v0 + 3 * size_t(f);
        const float* b = c.v1 + 3 * size_t(f);
        const float* d = c.v2 + 3 * size_t(f);
        for (int k = 0; k < 3; ++k) {
            lb.a[t].v0[k] = a[k];
            lb.b[t].e1[k] = b[k] - a[k];      // v0v1, reference src/primitives.cpp:52
            lb.c[t].e2[k] = d[k] - a[k];      // v0v2, reference src/primitives.cpp:53
            lo[k] = fminf(lo[k], fminf(a[k], fminf(b[k], d[k])));
            hi[k] = fmaxf(hi[k], fmaxf(a[k], fmaxf(b[k], d[k])));
        }
        lb.a[t].face = f;
    }
    c.leaves[c.leaf_base + j] = lb;
    for (int k = 0; k < 3; ++k) { c.leaf_box[6 * size_t(j) + k] = lo[k]; c.leaf_box[6 * size_t(j) + 3 + k] = hi[k]; }
}

__device__ __forceinline__ float ld_coherent(const float* p) {
    return __uint_as_float(__hip_atomic_load(reinterpret_cast<const uint32_t*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

// boxes of the internal nodes above the blocks, bottom-up: of the two threads that reach a node the second one owns it
__global__ void k_fit(Ctx c) {
    const uint32_t s = blockIdx.x * BLOCK + threadIdx.x;
    if (s >= c.n || !c.start_flag[s]) return;
    float box[6];
    for (int k = 0; k < 6; ++k) box[k] = c.leaf_box[6 * size_t(c.blk[s]) + k];
    uint32_t me = c.block_ref[s];
    uint32_t p = c.block_parent[s];
    bool internal = false;
    while (p != NONE) {
        if (internal)       // publish this subtree's box before announcing the arrival
            for (int k = 0; k < 6; ++k) __hip_atomic_store(reinterpret_cast<uint32_t*>(c.ibox + 6 * size_t(me) + k), __float_as_uint(box[k]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence();
        if (atomicAdd(c.flag + p, 1u) == 0u) return;       // first to arrive: the sibling's thread carries on
        __threadfence();
        const uint32_t l = c.child[2 * size_t(p)], r = c.child[2 * size_t(p) + 1];
        const uint32_t sib = l == me ? r : l;
        const float* sb = is_block(c, sib) ? c.leaf_box + 6 * size_t(c.blk[block_start(c, sib)]) : c.ibox + 6 * size_t(sib);
        for (int k = 0; k < 3; ++k) {
            box[k] = fminf(box[k], ld_coherent(sb + k));
            box[3 + k] = fmaxf(box[3 + k], ld_coherent(sb + 3 + k));
        }
        me = p;
        internal = true;
        p = c.parent_int[p];
    }
    if (internal) for (int k = 0; k < 6; ++k) c.ibox[6 * size_t(me) + k] = box[k];     // the root's box
}

__global__ void k_depth(Ctx c) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    const bool ok = i + 1 < c.n && c.cnt[i] > RT_MAX_LEAF_TRIS;
    uint32_t d = 0;
    if (i + 1 < c.n) {
        if (ok) for (uint32_t p = c.parent_int[i]; p != NONE; p = c.parent_int[p]) ++d;
        c.mark[i] = (ok && !(d & 1u)) ? 1u : 0u;
    }
    uint32_t m = d;
    for (int off = 32; off > 0; off >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, off));
    if ((threadIdx.x & 63) == 0) atomicMax(c.max_depth, m);
}

__device__ __forceinline__ void pad_box(const float* in, float* lo, float* hi) {     // as bvh_build.cpp's pad()
    for (int a = 0; a < 3; ++a) {
        const float m = fmaxf(fabsf(in[a]), fabsf(in[3 + a]));
        const float d = m * 2e-5f + 1e-7f;
        lo[a] = in[a] - d;
        hi[a] = in[3 + a] + d;
    }
}

__global__ void k_emit4(Ctx c) {
    const uint32_t b = blockIdx.x * BLOCK + threadIdx.x;
    if (b + 1 >= c.n || !c.mark[b]) return;
    uint32_t ref[4];
    int k = 0;
    for (int side = 0; side < 2; ++side) {
        const uint32_t ch = c.child[2 * size_t(b) + side];
        if (is_block(c, ch)) { ref[k++] = ch; continue; }
        ref[k++] = c.child[2 * size_t(ch)];
        ref[k++] = c.child[2 * size_t(ch) + 1];
    }
    Bvh4Node out;
    for (int i = 0; i < 4; ++i) {
        out.b[i].pad = 0;
        if (i >= k) {
            for (int a = 0; a < 3; ++a) { out.a[i].lo[a] = INFINITY; out.b[i].hi[a] = -INFINITY; }
            out.a[i].ref = RT_NO_CHILD;
            continue;
        }
        if (is_block(c, ref[i])) {
            const uint32_t s = block_start(c, ref[i]), j = c.blk[s];
            pad_box(c.leaf_box + 6 * size_t(j), out.a[i].lo, out.b[i].hi);
            out.a[i].ref = RT_LEAF_BIT | ((c.run_len[s] - 1u) << 27) | (c.leaf_base + j);
        } else {
            pad_box(c.ibox + 6 * size_t(ref[i]), out.a[i].lo, out.b[i].hi);
            out.a[i].ref = c.node_base + c.idx4[ref[i]];
        }
    }
    c.nodes[c.node_base + c.idx4[b]] = out;
}

struct Buf {
    void* p = nullptr;
    ~Buf() { if (p) (void)hipFree(p); }
    template <typename T> T* get(size_t n) { if (hipMalloc(&p, std::max<size_t>(n, 1) * sizeof(T)) != hipSuccess) p = nullptr; return static_cast<T*>(p); }
};

#define TRY(expr) do { if ((expr) != hipSuccess) { rt_set_error("device BVH build: %s failed: %s", #expr, hipGetErrorString(hipGetLastError())); return -1; } } while (0)

}  // namespace

int bvh_build_device(const rt_flat_scene* sc, DeviceBvh* out) {
    const auto t0 = std::chrono::steady_clock::now();
    out->nodes = nullptr; out->leaves = nullptr; out->n_nodes = 0; out->n_leaves = 0; out->max_depth = 0; out->max_stack = 1;
    out->model_root.assign(sc->n_models, RT_NO_CHILD);
    const uint32_t nf = sc->n_faces;
    uint32_t max_n = 0;
    for (uint32_t m = 0; m < sc->n_models; ++m) max_n = std::max(max_n, sc->models[m].num_faces);
    Buf bv0, bv1, bv2, bnodes, bleaves;
    float* v0 = bv0.get<float>(3 * size_t(nf));
    float* v1 = bv1.get<float>(3 * size_t(nf));
    float* v2 = bv2.get<float>(3 * size_t(nf));
    // upper bounds (a block per face, a node per block); the exact-size arrays are cut out at the end
    Bvh4Node* nodes = bnodes.get<Bvh4Node>(nf);
    LeafBlock* leaves = bleaves.get<LeafBlock>(nf);
    if (!v0 || !v1 || !v2 || !nodes || !leaves) { rt_set_error("device BVH build: out of memory"); return -1; }
    if (nf) {
        TRY(hipMemcpy(v0, sc->v0, sizeof(float) * 3 * size_t(nf), hipMemcpyHostToDevice));
        TRY(hipMemcpy(v1, sc->v1, sizeof(float) * 3 * size_t(nf), hipMemcpyHostToDevice));
        TRY(hipMemcpy(v2, sc->v2, sizeof(float) * 3 * size_t(nf), hipMemcpyHostToDevice));
    }

    Buf bbounds, bkey, bkey2, bface, bface2, bchild, bpi, bpl, bcnt, blo, bsf, brl, bbp, bbr, bblk, bleafbox, bibox, bflag, bmark, bidx, bdepth, btemp;
    Ctx c{};
    c.v0 = v0; c.v1 = v1; c.v2 = v2;
    c.bounds = bbounds.get<uint32_t>(6);
    uint64_t* key_a = bkey.get<uint64_t>(max_n); uint64_t* key_b = bkey2.get<uint64_t>(max_n);
    uint32_t* face_a = bface.get<uint32_t>(max_n); uint32_t* face_b = bface2.get<uint32_t>(max_n);
    c.child = bchild.get<uint32_t>(2 * size_t(max_n));
    c.parent_int = bpi.get<uint32_t>(max_n); c.parent_leaf = bpl.get<uint32_t>(max_n);
    c.cnt = bcnt.get<uint32_t>(max_n); c.lo = blo.get<uint32_t>(max_n);
    c.start_flag = bsf.get<uint32_t>(max_n); c.run_len = brl.get<uint32_t>(max_n);
    c.block_parent = bbp.get<uint32_t>(max_n); c.block_ref = bbr.get<uint32_t>(max_n); c.blk = bblk.get<uint32_t>(max_n);
    c.leaf_box = bleafbox.get<float>(6 * size_t(max_n));
    c.ibox = bibox.get<float>(6 * size_t(max_n));
    c.flag = bflag.get<uint32_t>(max_n);
    c.mark = bmark.get<uint32_t>(max_n); c.idx4 = bidx.get<uint32_t>(max_n);
    c.max_depth = bdepth.get<uint32_t>(1);
    c.nodes = nodes; c.leaves = leaves;
    if (!c.bounds || !key_a || !key_b || !face_a || !face_b || !c.child || !c.parent_int || !c.parent_leaf || !c.cnt || !c.lo ||
        !c.start_flag || !c.run_len || !c.block_parent || !c.block_ref || !c.blk || !c.leaf_box || !c.ibox || !c.flag || !c.mark ||
        !c.idx4 || !c.max_depth) { rt_set_error("device BVH build: out of memory"); return -1; }
    size_t sort_bytes = 0, scan_bytes = 0;
    if (max_n) {
        hipcub::DoubleBuffer<uint64_t> dk(key_a, key_b);
        hipcub::DoubleBuffer<uint32_t> dv(face_a, face_b);
        if (hipcub::DeviceRadixSort::SortPairs(nullptr, sort_bytes, dk, dv, int(max_n), 0, 63) != hipSuccess) { rt_set_error("device BVH build: sort sizing failed"); return -1; }
        if (hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, c.mark, c.idx4, int(max_n)) != hipSuccess) { rt_set_error("device BVH build: scan sizing failed"); return -1; }
    }
    void* temp = btemp.get<char>(std::max(sort_bytes, scan_bytes));
    if (!temp) { rt_set_error("device BVH build: out of memory"); return -1; }

    uint32_t node_base = 0, leaf_base = 0, depth4_max = 0;
    for (uint32_t m = 0; m < sc->n_models; ++m) {
        const rt_model& md = sc->models[m];
        if (md.num_faces == 0) continue;
        const uint32_t n = md.num_faces;
        c.first = md.first_face; c.n = n;
        c.node_base = node_base; c.leaf_base = leaf_base;
        const dim3 grid_n((n + BLOCK - 1) / BLOCK), block(BLOCK);
        const uint32_t init_bounds[6] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u};
        TRY(hipMemcpy(c.bounds, init_bounds, sizeof init_bounds, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_bounds, dim3(std::min<uint32_t>(grid_n.x, 2048u)), block, 0, nullptr, c);
        c.key = key_a; c.face = face_a;
        hipLaunchKernelGGL(k_morton, grid_n, block, 0, nullptr, c);
        {
            hipcub::DoubleBuffer<uint64_t> dk(key_a, key_b);
            hipcub::DoubleBuffer<uint32_t> dv(face_a, face_b);
            size_t bytes = sort_bytes;
            if (hipcub::DeviceRadixSort::SortPairs(temp, bytes, dk, dv, int(n), 0, 63) != hipSuccess) { rt_set_error("device BVH build: sort failed"); return -1; }
            c.key = dk.Current(); c.face = dv.Current();
        }
        TRY(hipMemsetAsync(c.start_flag, 0, sizeof(uint32_t) * n, nullptr));
        if (n <= RT_MAX_LEAF_TRIS) {
            hipLaunchKernelGGL(k_single_block, dim3(1), dim3(64), 0, nullptr, c);
        } else {
            hipLaunchKernelGGL(k_radix_tree, grid_n, block, 0, nullptr, c);
            hipLaunchKernelGGL(k_mark_blocks, grid_n, block, 0, nullptr, c);
        }
        size_t bytes = scan_bytes;
        if (hipcub::DeviceScan::ExclusiveSum(temp, bytes, c.start_flag, c.blk, int(n)) != hipSuccess) { rt_set_error("device BVH build: scan failed"); return -1; }
        uint32_t last_blk = 0, last_flag = 0;
        TRY(hipMemcpy(&last_blk, c.blk + (n - 1), sizeof(uint32_t), hipMemcpyDeviceToHost));
        TRY(hipMemcpy(&last_flag, c.start_flag + (n - 1), sizeof(uint32_t), hipMemcpyDeviceToHost));
        const uint32_t n_blocks = last_blk + last_flag;
        if (uint64_t(leaf_base) + n_blocks > RT_LEAF_START_MASK) return 1;
        hipLaunchKernelGGL(k_leaf_blocks, grid_n, block, 0, nullptr, c);
        uint32_t n4 = 0;
        if (n <= RT_MAX_LEAF_TRIS) {
            out->model_root[m] = RT_LEAF_BIT | ((n - 1u) << 27) | leaf_base;
        } else {
            TRY(hipMemsetAsync(c.flag, 0, sizeof(uint32_t) * n, nullptr));
            TRY(hipMemsetAsync(c.max_depth, 0, sizeof(uint32_t), nullptr));
            hipLaunchKernelGGL(k_fit, grid_n, block, 0, nullptr, c);
            hipLaunchKernelGGL(k_depth, grid_n, block, 0, nullptr, c);
            bytes = scan_bytes;
            if (hipcub::DeviceScan::ExclusiveSum(temp, bytes, c.mark, c.idx4, int(n - 1)) != hipSuccess) { rt_set_error("device BVH build: scan failed"); return -1; }
            uint32_t last_idx = 0, last_mark = 0, depth2 = 0;
            TRY(hipMemcpy(&last_idx, c.idx4 + (n - 2), sizeof(uint32_t), hipMemcpyDeviceToHost));
            TRY(hipMemcpy(&last_mark, c.mark + (n - 2), sizeof(uint32_t), hipMemcpyDeviceToHost));
            TRY(hipMemcpy(&depth2, c.max_depth, sizeof(uint32_t), hipMemcpyDeviceToHost));
            n4 = last_idx + last_mark;
            const uint32_t depth4 = depth2 / 2u + 1u;                 // four-wide nodes on the longest path
            if (3u * depth4 + 1u > RT_QSTACK) return 1;               // too deep for the traversal's stack: host builder
            hipLaunchKernelGGL(k_emit4, grid_n, block, 0, nullptr, c);
            out->model_root[m] = node_base;                            // the binary root is node 0 of the model
            depth4_max = std::max(depth4_max, depth4);
        }
        TRY(hipGetLastError());
        node_base += n4;
        leaf_base += n_blocks;
    }
    TRY(hipDeviceSynchronize());
    // exact-size copies for the scene to keep
    TRY(hipMalloc(reinterpret_cast<void**>(&out->nodes), sizeof(Bvh4Node) * std::max<uint32_t>(1, node_base)));
    if (hipMalloc(reinterpret_cast<void**>(&out->leaves), sizeof(LeafBlock) * std::max<uint32_t>(1, leaf_base)) != hipSuccess) {
        (void)hipFree(out->nodes); out->nodes = nullptr;
        rt_set_error("device BVH build: out of memory");
        return -1;
    }
    if (node_base) TRY(hipMemcpy(out->nodes, nodes, sizeof(Bvh4Node) * size_t(node_base), hipMemcpyDeviceToDevice));
    if (leaf_base) TRY(hipMemcpy(out->leaves, leaves, sizeof(LeafBlock) * size_t(leaf_base), hipMemcpyDeviceToDevice));
    out->n_nodes = node_base;
    out->n_leaves = leaf_base;
    out->max_depth = depth4_max;
    out->max_stack = 3u * depth4_max + 1u;
    out->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return 0;
}
