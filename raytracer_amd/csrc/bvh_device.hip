// bvh_device.hip — BVH build on the GPU: the same tree FORMAT as the host builder's
// (bvh_build.cpp: Bvh4Node / LeafBlock of rt_device.h), from a linear BVH:
//   1. centroid bounds of the model's faces, 63-bit Morton key per face (21 bits per axis)
//   2. radix sort of (key, face)                                   hipcub::DeviceRadixSort
//   3. binary radix tree over the sorted faces (Karras 2012: one thread per internal node, no synchronisation)
//   4. leaf blocks, with the floats Triangle::intersect forms per call (reference src/primitives.cpp:52-53) and their
//      exact boxes, numbered by a prefix sum. A maximal subtree with at most 16 faces (a "treelet": a run of sorted
//      faces) becomes ONE 4-wide node over blocks of 4 faces (the last one the rest), grouped as the host builder groups
//      a range of <= 16 faces: sorted along an axis and cut at the multiples of 4 of least surface-area cost,
//      recursively (k_treelet_order) - so the blocks are 3.5 faces full instead of the 2.9 that "maximal subtrees with
//      <= 4 faces" gave, without the overlap that runs of consecutive Morton-order faces have
//   5. boxes of the nodes above the blocks, bottom-up (the second thread to arrive at a node continues upwards)
//   5b. the TOP of the tree is rebuilt by surface-area heuristic (HLBVH-style): the radix-tree nodes whose faces share
//      fewer than P key bits (P ~ log2(n / 16): cells of ~16 faces; 64 / 16 / 2 measured: frames of the 10 M-triangle field 5.15 / 5.03 / 5.01 ms) are the part of the tree where Morton splits are
//      worst - they cut space by bit planes, through objects; the subtrees hanging below them ("entries": ~n / 16 of
//      them) keep their Morton order, their boxes go to the host, a binned-SAH binary tree over those boxes is built there
//      (milliseconds) and written back over the SAME node slots (a binary tree over K entries has K - 1 inner nodes
//      either way)
//   6. the binary tree above the treelets becomes 4-wide top-down, level by level, as the host builder collapses its
//      tree: a kept node starts with its two children and twice opens the child with the largest surface that can be
//      opened; the inner nodes among the resulting (at most 4) children are kept in turn. Kept nodes are numbered by a
//      prefix sum; the number of levels is the exact depth of the 4-wide tree (traversal stack bound).
// Like the host tree, it only decides WHICH triangles get tested (see bvh_build.cpp); a Morton-order tree culls less
// well than the SAH tree (slower frames), but builds in milliseconds instead of seconds. The worst-case traversal
// stack follows from the depth of the binary tree; a model whose tree would be too deep for the traversal's LDS stack
// is reported as not applicable and the caller falls back to the host builder.
// The reference has no acceleration structure; this replaces, with bvh_build.cpp, the loop of Model::intersect
// (reference src/primitives.cpp:376-382).
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
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
    uint32_t* kids;          // 4 per kept internal node: its children in the 4-wide tree (tree references; NONE: none)
    uint32_t* front_a; uint32_t* front_b; uint32_t* front_count;
    uint32_t* max_depth;     // depth of the deepest internal node with more than 4 faces (root = 0)
    // top rebuild (5b)
    uint32_t* plen;          // per internal node: common key prefix of its faces, in bits (delta())
    uint32_t top_bits;       // nodes with plen <= top_bits (and more than 4 faces) form the top
    uint32_t* top_count;     // [0] entries, [1] top nodes
    uint32_t* ent_ref; float* ent_box; uint32_t* top_node; uint32_t top_cap;
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
    c.plen[i] = (uint32_t)dnode;
}

// A tree reference below which at most 4 faces lie is the root of a leaf block (when it is not inside a treelet).
__device__ __forceinline__ bool is_block(const Ctx& c, uint32_t ref) { return (ref & RT_LEAF_BIT) || c.cnt[ref] <= RT_MAX_LEAF_TRIS; }
__device__ __forceinline__ uint32_t block_start(const Ctx& c, uint32_t ref) { return (ref & RT_LEAF_BIT) ? (ref & ~RT_LEAF_BIT) : c.lo[ref]; }
// Treelets: maximal subtrees with 5..RT_TREELET faces; their root becomes one 4-wide node over blocks of consecutive faces.
#define RT_TREELET 16u
__device__ __forceinline__ bool above_treelets(const Ctx& c, uint32_t node) { return node == NONE || c.cnt[node] > RT_TREELET; }
__device__ __forceinline__ bool treelet_root(const Ctx& c, uint32_t i) {      // i: internal node
    const uint32_t n = c.cnt[i];
    return n > RT_MAX_LEAF_TRIS && n <= RT_TREELET && above_treelets(c, c.parent_int[i]);
}
// block g of a treelet with n faces: first face (relative) and size
__device__ __forceinline__ void treelet_block(uint32_t n, uint32_t g, uint32_t& off, uint32_t& size) {
    off = g * RT_MAX_LEAF_TRIS;
    size = n - off < RT_MAX_LEAF_TRIS ? n - off : RT_MAX_LEAF_TRIS;
}
// Which faces of a treelet share a block: not runs of the Morton order (measured: fuller blocks, but slower frames than the
// subtrees they replaced - their boxes overlap) but what the host builder does with a range of <= 16 faces: the range is
// sorted along the axis and cut at the multiple of 4 that minimise the surface-area cost, recursively. One thread per
// treelet, at most 16 faces in local arrays; the faces are reordered within the treelet's own run of c.face.
__global__ void k_treelet_order(Ctx c) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i + 1 >= c.n || !treelet_root(c, i)) return;
    const uint32_t n = c.cnt[i], base = c.lo[i];
    uint32_t f[RT_TREELET];
    float lo[RT_TREELET][3], hi[RT_TREELET][3];
    for (uint32_t k = 0; k < n; ++k) {
        f[k] = c.face[base + k];
        const float* a = c.v0 + 3 * size_t(f[k]);
        const float* b = c.v1 + 3 * size_t(f[k]);
        const float* d = c.v2 + 3 * size_t(f[k]);
        for (int x = 0; x < 3; ++x) { lo[k][x] = fminf(a[x], fminf(b[x], d[x])); hi[k][x] = fmaxf(a[x], fmaxf(b[x], d[x])); }
    }
    uint32_t perm[RT_TREELET], trial[RT_TREELET];
    for (uint32_t k = 0; k < n; ++k) perm[k] = k;
    uint32_t stack_a[4], stack_b[4];
    int sp = 0;
    stack_a[0] = 0; stack_b[0] = n; sp = 1;
    while (sp > 0) {
        --sp;
        const uint32_t ra = stack_a[sp], rb = stack_b[sp];
        if (rb - ra <= RT_MAX_LEAF_TRIS) continue;
        float best = INFINITY; int best_axis = 0; uint32_t best_k = ra + RT_MAX_LEAF_TRIS;
        for (int axis = 0; axis < 3; ++axis) {
            for (uint32_t k = ra; k < rb; ++k) trial[k] = perm[k];
            for (uint32_t k = ra + 1; k < rb; ++k) {       // insertion sort by centroid
                const uint32_t e = trial[k];
                const float m = lo[e][axis] + hi[e][axis];
                uint32_t q = k;
                while (q > ra && lo[trial[q - 1]][axis] + hi[trial[q - 1]][axis] > m) { trial[q] = trial[q - 1]; --q; }
                trial[q] = e;
            }
            for (uint32_t k = ra + RT_MAX_LEAF_TRIS; k < rb; k += RT_MAX_LEAF_TRIS) {
                float cost = 0;
                for (int side = 0; side < 2; ++side) {
                    const uint32_t a = side ? k : ra, b = side ? rb : k;
                    float bl[3] = {INFINITY, INFINITY, INFINITY}, bh[3] = {-INFINITY, -INFINITY, -INFINITY};
                    for (uint32_t q = a; q < b; ++q) for (int x = 0; x < 3; ++x) { bl[x] = fminf(bl[x], lo[trial[q]][x]); bh[x] = fmaxf(bh[x], hi[trial[q]][x]); }
                    const float dx = bh[0] - bl[0], dy = bh[1] - bl[1], dz = bh[2] - bl[2];
                    cost += (dx * dy + dy * dz + dz * dx) * float((b - a + RT_MAX_LEAF_TRIS - 1) / RT_MAX_LEAF_TRIS);      // blocks on that side
                }
                if (cost < best) { best = cost; best_axis = axis; best_k = k; }
            }
        }
        for (uint32_t k = ra + 1; k < rb; ++k) {           // the winning order
            const uint32_t e = perm[k];
            const float m = lo[e][best_axis] + hi[e][best_axis];
            uint32_t q = k;
            while (q > ra && lo[perm[q - 1]][best_axis] + hi[perm[q - 1]][best_axis] > m) { perm[q] = perm[q - 1]; --q; }
            perm[q] = e;
        }
        stack_a[sp] = ra; stack_b[sp] = best_k; ++sp;
        stack_a[sp] = best_k; stack_b[sp] = rb; ++sp;
    }
    for (uint32_t k = 0; k < n; ++k) c.face[base + k] = f[perm[k]];
}

// n <= 4: the whole model is one block
__global__ void k_single_block(Ctx c) {
    if (blockIdx.x == 0 && threadIdx.x == 0) { c.start_flag[0] = 1; c.run_len[0] = c.n; c.block_parent[0] = NONE; c.block_ref[0] = NONE; }
}

__global__ void k_mark_blocks(Ctx c) {
    const uint32_t t = blockIdx.x * BLOCK + threadIdx.x;
    if (t >= c.n) return;
    {   // the face at sorted position t, if it hangs directly below a node that is above the treelets
        const uint32_t p = c.parent_leaf[t];
        if (c.cnt[p] > RT_TREELET) { c.start_flag[t] = 1; c.run_len[t] = 1; c.block_parent[t] = p; c.block_ref[t] = RT_LEAF_BIT | t; }
    }
    if (t + 1 >= c.n) return;
    const uint32_t p = c.parent_int[t], nf = c.cnt[t];
    if (nf <= RT_MAX_LEAF_TRIS) {      // internal node t with <= 4 faces below a node above the treelets: one block
        if (p != NONE && c.cnt[p] > RT_TREELET) {
            const uint32_t s = c.lo[t];
            c.start_flag[s] = 1; c.run_len[s] = nf; c.block_parent[s] = p; c.block_ref[s] = t;
        }
    } else if (treelet_root(c, t)) {    // 5..16 faces: blocks of consecutive faces (block_ref NONE: not a tree reference)
        const uint32_t nb = (nf + RT_MAX_LEAF_TRIS - 1) / RT_MAX_LEAF_TRIS;
        for (uint32_t g = 0; g < nb; ++g) {
            uint32_t off, size;
            treelet_block(nf, g, off, size);
            const uint32_t s = c.lo[t] + off;
            c.start_flag[s] = 1; c.run_len[s] = size; c.block_parent[s] = t; c.block_ref[s] = NONE;
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

// boxes of the internal nodes above the blocks and treelets, bottom-up: of the two threads that reach a node the second
// one owns it. `me`: the subtree this thread has the box of (a tree reference), p: its parent.
__device__ __forceinline__ void fit_walk(const Ctx& c, uint32_t me, uint32_t p, float* box, bool internal) {
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
__global__ void k_fit(Ctx c) {
    const uint32_t s = blockIdx.x * BLOCK + threadIdx.x;
    if (s < c.n && c.start_flag[s] && c.block_ref[s] != NONE) {      // a block that is a subtree of its own
        float box[6];
        for (int k = 0; k < 6; ++k) box[k] = c.leaf_box[6 * size_t(c.blk[s]) + k];
        fit_walk(c, c.block_ref[s], c.block_parent[s], box, false);
    }
    if (s + 1 < c.n && treelet_root(c, s)) {                          // a treelet: the union of its blocks
        float box[6] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY};
        const uint32_t nf = c.cnt[s], nb = (nf + RT_MAX_LEAF_TRIS - 1) / RT_MAX_LEAF_TRIS;
        for (uint32_t g = 0; g < nb; ++g) {
            uint32_t off, size;
            treelet_block(nf, g, off, size);
            const float* lb = c.leaf_box + 6 * size_t(c.blk[c.lo[s] + off]);
            for (int k = 0; k < 3; ++k) { box[k] = fminf(box[k], lb[k]); box[3 + k] = fmaxf(box[3 + k], lb[3 + k]); }
        }
        fit_walk(c, s, c.parent_int[s], box, true);
    }
}

// ---- 5b: the top of the tree ----
__device__ __forceinline__ bool in_top(const Ctx& c, uint32_t ref) {
    return !(ref & RT_LEAF_BIT) && c.cnt[ref] > RT_TREELET && c.plen[ref] <= c.top_bits;
}
// every top node lists itself and those of its children that are not top nodes (the entries), with their boxes
__global__ void k_top_collect(Ctx c) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i + 1 >= c.n || !in_top(c, i)) return;
    const uint32_t t = atomicAdd(c.top_count + 1, 1u);
    if (t < c.top_cap) c.top_node[t] = i;
    for (int side = 0; side < 2; ++side) {
        const uint32_t ch = c.child[2 * size_t(i) + side];
        if (in_top(c, ch)) continue;
        const uint32_t e = atomicAdd(c.top_count, 1u);
        if (e >= c.top_cap) continue;
        c.ent_ref[e] = ch;
        const float* b = is_block(c, ch) ? c.leaf_box + 6 * size_t(c.blk[block_start(c, ch)]) : c.ibox + 6 * size_t(ch);
        for (int k = 0; k < 6; ++k) c.ent_box[6 * size_t(e) + k] = b[k];
    }
}
// the rebuilt top: node[t] gets children l[t], r[t] (tree references) and box[t]; the children learn their parent
__global__ void k_top_apply(Ctx c, uint32_t n_top, const uint32_t* node, const uint32_t* l, const uint32_t* r, const float* box) {
    const uint32_t t = blockIdx.x * BLOCK + threadIdx.x;
    if (t >= n_top) return;
    const uint32_t me = node[t];
    c.child[2 * size_t(me)] = l[t];
    c.child[2 * size_t(me) + 1] = r[t];
    for (int k = 0; k < 6; ++k) c.ibox[6 * size_t(me) + k] = box[6 * size_t(t) + k];
    for (int side = 0; side < 2; ++side) {
        const uint32_t ch = side ? r[t] : l[t];
        if (!(ch & RT_LEAF_BIT)) c.parent_int[ch] = me;      // (a face directly below a top node has no parent link anyone reads)
    }
    if (t == 0) c.parent_int[me] = NONE;
}

// ---- 6: collapse to 4 children per node, one level of the 4-wide tree per launch ----
__device__ __forceinline__ float box_half_area(const float* b) {
    const float dx = b[3] - b[0], dy = b[4] - b[1], dz = b[5] - b[2];
    return dx * dy + dy * dz + dz * dx;
}
__global__ void k_collapse_level(Ctx c, const uint32_t* front, uint32_t n_front, uint32_t* next) {
    const uint32_t t = blockIdx.x * BLOCK + threadIdx.x;
    if (t >= n_front) return;
    const uint32_t b = front[t];
    c.mark[b] = 1u;
    if (treelet_root(c, b)) return;            // its children are its blocks (k_emit4)
    uint32_t ref[4] = {c.child[2 * size_t(b)], c.child[2 * size_t(b) + 1], NONE, NONE};
    int k = 2;
    for (int round = 0; round < 2; ++round) {
        int pick = -1;
        float best = -1.0f;
        for (int i = 0; i < k; ++i) {
            if (is_block(c, ref[i]) || treelet_root(c, ref[i])) continue;
            const float a = box_half_area(c.ibox + 6 * size_t(ref[i]));
            if (a > best) { best = a; pick = i; }
        }
        if (pick < 0) break;
        const uint32_t open = ref[pick];
        ref[pick] = c.child[2 * size_t(open)];
        ref[k++] = c.child[2 * size_t(open) + 1];
    }
    for (int i = 0; i < 4; ++i) {
        c.kids[4 * size_t(b) + i] = ref[i];
        if (i < k && !is_block(c, ref[i])) next[atomicAdd(c.front_count, 1u)] = ref[i];      // an inner node: kept
    }
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
    // children: a block (given by the sorted position it starts at) or a 4-wide node (given by its binary node)
    uint32_t ref[4];
    bool blk[4];
    int k = 0;
    if (treelet_root(c, b)) {
        const uint32_t nf = c.cnt[b], nb = (nf + RT_MAX_LEAF_TRIS - 1) / RT_MAX_LEAF_TRIS;
        for (uint32_t g = 0; g < nb; ++g) {
            uint32_t off, size;
            treelet_block(nf, g, off, size);
            ref[k] = c.lo[b] + off; blk[k] = true; ++k;
        }
    } else {
        for (int i = 0; i < 4; ++i) {
            const uint32_t r = c.kids[4 * size_t(b) + i];
            if (r == NONE) continue;
            if (is_block(c, r)) { ref[k] = block_start(c, r); blk[k] = true; } else { ref[k] = r; blk[k] = false; }
            ++k;
        }
    }
    Bvh4Node out;
    for (int i = 0; i < 4; ++i) {
        out.b[i].pad = 0;
        if (i >= k) {
            for (int a = 0; a < 3; ++a) { out.a[i].lo[a] = INFINITY; out.b[i].hi[a] = -INFINITY; }
            out.a[i].ref = RT_NO_CHILD;
            continue;
        }
        if (blk[i]) {
            const uint32_t s = ref[i], j = c.blk[s];
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

// Binned-SAH binary tree over K boxes (host; K ~ n / 64). Inner node t: children l[t], r[t] - an inner node index, or
// K_LEAF | entry - and its box; node 0 is the root, K - 1 inner nodes in all.
struct TopTree { std::vector<uint32_t> l, r; std::vector<float> box; };
constexpr uint32_t K_LEAF = 0x80000000u;
struct TopBuilder {
    const float* eb;                       // 6 per entry
    std::vector<uint32_t> idx;
    TopTree* out;
    static float half_area(const float* b) {
        const float dx = b[3] - b[0], dy = b[4] - b[1], dz = b[5] - b[2];
        return dx * dy + dy * dz + dz * dx;
    }
    static void grow(float* b, const float* e) { for (int k = 0; k < 3; ++k) { b[k] = std::min(b[k], e[k]); b[3 + k] = std::max(b[3 + k], e[3 + k]); } }
    static void empty(float* b) { for (int k = 0; k < 3; ++k) { b[k] = INFINITY; b[3 + k] = -INFINITY; } }
    // builds the subtree over idx[lo, hi) (hi - lo >= 2) into inner node `me` (already allocated)
    void build(uint32_t me, uint32_t lo, uint32_t hi) {
        struct Job { uint32_t me, lo, hi; };
        std::vector<Job> todo{{me, lo, hi}};
        while (!todo.empty()) {
            const Job j = todo.back(); todo.pop_back();
            float nb[6], cb[6]; empty(nb); empty(cb);
            for (uint32_t k = j.lo; k < j.hi; ++k) {
                const float* e = eb + 6 * size_t(idx[k]);
                grow(nb, e);
                for (int a = 0; a < 3; ++a) { const float m = 0.5f * (e[a] + e[3 + a]); cb[a] = std::min(cb[a], m); cb[3 + a] = std::max(cb[3 + a], m); }
            }
            for (int k = 0; k < 6; ++k) out->box[6 * size_t(j.me) + k] = nb[k];
            uint32_t mid = (j.lo + j.hi) / 2;
            const uint32_t cnt = j.hi - j.lo;
            if (cnt > 2) {
                constexpr int NB = 16;
                float best = INFINITY; int best_axis = -1, best_bin = 0;
                for (int a = 0; a < 3; ++a) {
                    const float ext = cb[3 + a] - cb[a];
                    if (!(ext > 0)) continue;
                    float bb[NB][6]; uint32_t bc[NB] = {};
                    for (int b = 0; b < NB; ++b) empty(bb[b]);
                    const float scale = NB / ext;
                    for (uint32_t k = j.lo; k < j.hi; ++k) {
                        const float* e = eb + 6 * size_t(idx[k]);
                        const int b = std::min(NB - 1, std::max(0, int((0.5f * (e[a] + e[3 + a]) - cb[a]) * scale)));
                        grow(bb[b], e); ++bc[b];
                    }
                    float ra[NB]; uint32_t rc[NB]; float acc[6]; empty(acc); uint32_t n = 0;
                    for (int b = NB - 1; b > 0; --b) { if (bc[b]) grow(acc, bb[b]); n += bc[b]; ra[b] = n ? half_area(acc) : 0.0f; rc[b] = n; }
                    empty(acc); n = 0;
                    for (int b = 0; b + 1 < NB; ++b) {
                        if (bc[b]) grow(acc, bb[b]); n += bc[b];
                        if (!n || !rc[b + 1]) continue;
                        const float cost = half_area(acc) * float(n) + ra[b + 1] * float(rc[b + 1]);
                        if (cost < best) { best = cost; best_axis = a; best_bin = b; }
                    }
                }
                if (best_axis >= 0) {
                    const int a = best_axis;
                    const float scale = 16 / (cb[3 + a] - cb[a]);
                    auto it = std::partition(idx.begin() + j.lo, idx.begin() + j.hi, [&](uint32_t e) {
                        const float* q = eb + 6 * size_t(e);
                        return std::min(15, std::max(0, int((0.5f * (q[a] + q[3 + a]) - cb[a]) * scale))) <= best_bin;
                    });
                    mid = uint32_t(it - idx.begin());
                    if (mid == j.lo || mid == j.hi) mid = (j.lo + j.hi) / 2;
                }
            }
            for (int side = 0; side < 2; ++side) {
                const uint32_t a = side ? mid : j.lo, b = side ? j.hi : mid;
                uint32_t ref;
                if (b - a == 1) ref = K_LEAF | idx[a];
                else { ref = uint32_t(out->l.size()); out->l.push_back(0); out->r.push_back(0); out->box.resize(out->box.size() + 6); todo.push_back({ref, a, b}); }
                (side ? out->r : out->l)[j.me] = ref;
            }
        }
    }
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
    Buf bkids, bfa, bfb, bfc;
    c.kids = bkids.get<uint32_t>(4 * size_t(max_n));
    c.front_a = bfa.get<uint32_t>(max_n); c.front_b = bfb.get<uint32_t>(max_n); c.front_count = bfc.get<uint32_t>(1);
    if (!c.kids || !c.front_a || !c.front_b || !c.front_count) { rt_set_error("device BVH build: out of memory"); return -1; }
    Buf bplen, btc, ber, beb, btn, btl, btr, btb, btnode;
    c.plen = bplen.get<uint32_t>(max_n);
    c.top_count = btc.get<uint32_t>(2);
    c.top_cap = std::min<uint32_t>(max_n, 1u << 22);
    c.ent_ref = ber.get<uint32_t>(c.top_cap); c.ent_box = beb.get<float>(6 * size_t(c.top_cap)); c.top_node = btn.get<uint32_t>(c.top_cap);
    uint32_t* d_tl = btl.get<uint32_t>(c.top_cap); uint32_t* d_tr = btr.get<uint32_t>(c.top_cap);
    float* d_tb = btb.get<float>(6 * size_t(c.top_cap)); uint32_t* d_tnode = btnode.get<uint32_t>(c.top_cap);
    if (!c.plen || !c.top_count || !c.ent_ref || !c.ent_box || !c.top_node || !d_tl || !d_tr || !d_tb || !d_tnode) { rt_set_error("device BVH build: out of memory"); return -1; }
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
            hipLaunchKernelGGL(k_treelet_order, grid_n, block, 0, nullptr, c);
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
            // 5b: rebuild the top by SAH. Cells of ~16 faces: log2(n / 16) key bits, a multiple of 3 (delta() counts from bit
            // 63 of a 63-bit key, hence the + 1). RT_BVH_TOP=0 in the environment keeps the Morton top (experiments).
            {
                static const bool top_on = !(std::getenv("RT_BVH_TOP") && std::atoi(std::getenv("RT_BVH_TOP")) == 0);
                static const uint32_t cell = std::getenv("RT_BVH_CELL") ? uint32_t(std::max(1, std::atoi(std::getenv("RT_BVH_CELL")))) : 16u;   // experiments
                uint32_t bits = 0;
                while ((uint64_t(cell) << bits) < n) bits += 3;
                c.top_bits = bits + 1;
                uint32_t cnts[2] = {0, 0};
                if (top_on && bits > 0) {
                    TRY(hipMemsetAsync(c.top_count, 0, 2 * sizeof(uint32_t), nullptr));
                    hipLaunchKernelGGL(k_top_collect, grid_n, block, 0, nullptr, c);
                    TRY(hipMemcpy(cnts, c.top_count, sizeof cnts, hipMemcpyDeviceToHost));
                }
                const uint32_t K = cnts[0], T = cnts[1];
                if (K >= 3 && K <= c.top_cap && T == K - 1) {
                    std::vector<uint32_t> h_ref(K), h_node(T);
                    std::vector<float> h_box(6 * size_t(K));
                    TRY(hipMemcpy(h_ref.data(), c.ent_ref, sizeof(uint32_t) * K, hipMemcpyDeviceToHost));
                    TRY(hipMemcpy(h_box.data(), c.ent_box, sizeof(float) * 6 * size_t(K), hipMemcpyDeviceToHost));
                    TRY(hipMemcpy(h_node.data(), c.top_node, sizeof(uint32_t) * T, hipMemcpyDeviceToHost));
                    // slot 0 of the new top is the tree's root, which must stay binary node 0
                    for (uint32_t t = 0; t < T; ++t) if (h_node[t] == 0u) { std::swap(h_node[t], h_node[0]); break; }
                    if (h_node[0] == 0u) {
                        TopTree tt;
                        tt.l.assign(1, 0); tt.r.assign(1, 0); tt.box.assign(6, 0.0f);
                        tt.l.reserve(T); tt.r.reserve(T); tt.box.reserve(6 * size_t(T));
                        TopBuilder tb{h_box.data(), {}, &tt};
                        tb.idx.resize(K);
                        for (uint32_t e = 0; e < K; ++e) tb.idx[e] = e;
                        tb.build(0, 0, K);
                        if (tt.l.size() == T) {
                            for (uint32_t t = 0; t < T; ++t) {       // inner index -> binary node, entry -> its tree reference
                                tt.l[t] = (tt.l[t] & K_LEAF) ? h_ref[tt.l[t] & ~K_LEAF] : h_node[tt.l[t]];
                                tt.r[t] = (tt.r[t] & K_LEAF) ? h_ref[tt.r[t] & ~K_LEAF] : h_node[tt.r[t]];
                            }
                            TRY(hipMemcpy(d_tnode, h_node.data(), sizeof(uint32_t) * T, hipMemcpyHostToDevice));
                            TRY(hipMemcpy(d_tl, tt.l.data(), sizeof(uint32_t) * T, hipMemcpyHostToDevice));
                            TRY(hipMemcpy(d_tr, tt.r.data(), sizeof(uint32_t) * T, hipMemcpyHostToDevice));
                            TRY(hipMemcpy(d_tb, tt.box.data(), sizeof(float) * 6 * size_t(T), hipMemcpyHostToDevice));
                            hipLaunchKernelGGL(k_top_apply, dim3((T + BLOCK - 1) / BLOCK), block, 0, nullptr, c, T, d_tnode, d_tl, d_tr, d_tb);
                        }
                    }
                }
            }
            // 6: the 4-wide tree, level by level from the root (binary node 0)
            TRY(hipMemsetAsync(c.mark, 0, sizeof(uint32_t) * n, nullptr));
            uint32_t depth4 = 0, n_front = 1;
            {
                const uint32_t zero = 0;
                TRY(hipMemcpy(c.front_a, &zero, sizeof zero, hipMemcpyHostToDevice));
                uint32_t* cur_f = c.front_a; uint32_t* nxt_f = c.front_b;
                while (n_front) {
                    ++depth4;
                    if (3u * depth4 + 1u > RT_QSTACK) return 1;       // too deep for the traversal's stack: host builder
                    TRY(hipMemsetAsync(c.front_count, 0, sizeof(uint32_t), nullptr));
                    hipLaunchKernelGGL(k_collapse_level, dim3((n_front + BLOCK - 1) / BLOCK), block, 0, nullptr, c, cur_f, n_front, nxt_f);
                    TRY(hipMemcpy(&n_front, c.front_count, sizeof(uint32_t), hipMemcpyDeviceToHost));
                    std::swap(cur_f, nxt_f);
                }
            }
            bytes = scan_bytes;
            if (hipcub::DeviceScan::ExclusiveSum(temp, bytes, c.mark, c.idx4, int(n - 1)) != hipSuccess) { rt_set_error("device BVH build: scan failed"); return -1; }
            uint32_t last_idx = 0, last_mark = 0;
            TRY(hipMemcpy(&last_idx, c.idx4 + (n - 2), sizeof(uint32_t), hipMemcpyDeviceToHost));
            TRY(hipMemcpy(&last_mark, c.mark + (n - 2), sizeof(uint32_t), hipMemcpyDeviceToHost));
            n4 = last_idx + last_mark;
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
