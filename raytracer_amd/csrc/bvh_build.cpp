// bvh_build.cpp — host BVH builder for the Model objects of a flat scene: binned-SAH binary tree, then collapsed to
// the 4-children-per-node layout the quad traversal uses (rt_device.h).
//
// Replaces the reference's O(#faces) loop in Model::intersect (reference src/primitives.cpp:376-382) with a
// hierarchy the GPU can cull with; the reference has no acceleration structure (its bounding-sphere gate,
// src/primitives.cpp:370, is kept separately and applied as is). The hierarchy only decides WHICH triangles get
// tested; every accepted hit still comes from the reference's Moeller-Trumbore arithmetic, and ties are broken
// by original face index, so closest hits equal the brute-force loop's (see DESIGN.md "BVH == brute force").
//
// Boxes are padded outward (relative 2e-5 of the coordinate magnitude + 1e-7) so that a hit the float
// Moeller-Trumbore test accepts slightly outside the exact triangle is still inside its leaf box.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "raytracer_amd.h"
#include "rt_device.h"

namespace {

struct Box {
    float lo[3], hi[3];
    void reset() { for (int a = 0; a < 3; ++a) { lo[a] = INFINITY; hi[a] = -INFINITY; } }
    void grow(const float* p) { for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], p[a]); hi[a] = std::max(hi[a], p[a]); } }
    void grow(const Box& b) { for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], b.lo[a]); hi[a] = std::max(hi[a], b.hi[a]); } }
    float half_area() const {
        const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        return dx * dy + dy * dz + dz * dx;
    }
};

struct Builder {
    const rt_flat_scene* sc;
    std::vector<Box> tri_box;        // per original face
    std::vector<float> centroid;     // 3 per face
    std::vector<uint32_t> order;     // permutation of faces, leaf order at the end
    struct Node2 { Box box[2]; uint32_t child[2]; };   // binary build tree: unpadded child boxes
    std::vector<Node2> nodes2;
    std::atomic<uint32_t> n_nodes2{0};
    bool balanced = false;           // median splits only (fallback when the SAH tree needs too much stack)
    Bvh4Node* nodes = nullptr;       // output
    uint32_t n_nodes = 0;
    std::vector<uint32_t> leaf_first;   // per leaf block: first position in order[], count in the reference
    uint32_t leaf_block(uint32_t ref2) {   // binary-tree leaf reference (first position) -> block reference
        leaf_first.push_back(ref2 & RT_LEAF_START_MASK);
        return (ref2 & ~RT_LEAF_START_MASK) | uint32_t(leaf_first.size() - 1);
    }

    static constexpr int NBINS = 16;

    Box range_box(uint32_t b, uint32_t e) const {
        Box bx; bx.reset();
        for (uint32_t i = b; i < e; ++i) bx.grow(tri_box[order[i]]);
        return bx;
    }

    static void pad(const Box& in, float* lo, float* hi) {
        for (int a = 0; a < 3; ++a) {
            const float m = std::max(std::fabs(in.lo[a]), std::fabs(in.hi[a]));
            const float d = m * 2e-5f + 1e-7f;
            lo[a] = in.lo[a] - d;
            hi[a] = in.hi[a] + d;
        }
    }

    // Returns the child reference for faces order[b..e) and their (unpadded) box in `bx`.
    uint32_t build(uint32_t b, uint32_t e, uint32_t depth, Box& bx) {
        const uint32_t n = e - b;
        bx = range_box(b, e);
        if (n <= RT_MAX_LEAF_TRIS) return RT_LEAF_BIT | ((n - 1) << 27) | b;
        // depth guard: once a balanced split is the only way to stay within RT_BVH_MAX_DEPTH, use it
        uint32_t need = 0;
        while ((uint64_t(RT_MAX_LEAF_TRIS) << need) < n) ++need;
        uint32_t mid = 0;
        // (RT_BVH_FULL, experiments: the exact sweep with full leaves for ranges up to 64 / 256 / 1024 faces instead of 16
        // changes frames by under 1 % - bunny_1080p 0.274 / 0.273 / 0.272 / 0.272 ms, 10 M-triangle field 4.59 / 4.56 / 4.55 / 4.54)
        static const uint32_t full_max = std::getenv("RT_BVH_FULL") ? uint32_t(std::atoi(std::getenv("RT_BVH_FULL"))) : 4 * RT_MAX_LEAF_TRIS;
        if (!balanced && n <= full_max) mid = full_leaf_split(b, e);
        else if (!balanced && depth + need + 1 < RT_BVH2_MAX_DEPTH) mid = sah_split(b, e, bx);
        if (mid <= b || mid >= e) mid = median_split(b, e, bx);

        const uint32_t idx = n_nodes2.fetch_add(1);
        Box lb, rb;
        uint32_t lc, rc;
        if (n > 100000) {
#pragma omp task shared(lc, lb) default(shared)
            lc = build(b, mid, depth + 1, lb);
#pragma omp task shared(rc, rb) default(shared)
            rc = build(mid, e, depth + 1, rb);
#pragma omp taskwait
        } else {
            lc = build(b, mid, depth + 1, lb);
            rc = build(mid, e, depth + 1, rb);
        }
        Node2& nd = nodes2[idx];
        nd.box[0] = lb; nd.box[1] = rb;
        nd.child[0] = lc; nd.child[1] = rc;
        return idx;
    }

    // Collapse the binary subtree under interior node `n2` into 4-wide nodes. Children are gathered by opening,
    // while fewer than 4 are held, the interior child with the largest surface (by_area) or simply both children of
    // the node (level pairs: exactly two binary levels per 4-wide level, used with the balanced fallback tree).
    // Returns the new node's index; depth = interior nodes on the longest path below and including it; stack = the
    // most entries a traversal below it can have pending (k-1 siblings at every level it descends through).
    uint32_t collapse(uint32_t n2, bool by_area, uint32_t& depth, uint32_t& stack) {
        Box cb[4];
        uint32_t cr[4];
        int k = 2;
        cb[0] = nodes2[n2].box[0]; cb[1] = nodes2[n2].box[1];
        cr[0] = nodes2[n2].child[0]; cr[1] = nodes2[n2].child[1];
        if (by_area) {
            while (k < 4) {
                int pick = -1;
                float area = -1;
                for (int i = 0; i < k; ++i)
                    if (!(cr[i] & RT_LEAF_BIT) && cb[i].half_area() > area) { area = cb[i].half_area(); pick = i; }
                if (pick < 0) break;
                const Node2& c = nodes2[cr[pick]];
                cb[k] = c.box[1]; cr[k] = c.child[1];
                cb[pick] = c.box[0]; cr[pick] = c.child[0];
                ++k;
            }
        } else {
            const Box sb[2] = {nodes2[n2].box[0], nodes2[n2].box[1]};
            const uint32_t sr[2] = {nodes2[n2].child[0], nodes2[n2].child[1]};
            k = 0;
            for (int side = 0; side < 2; ++side) {
                if (sr[side] & RT_LEAF_BIT) { cb[k] = sb[side]; cr[k] = sr[side]; ++k; continue; }
                const Node2& g = nodes2[sr[side]];
                cb[k] = g.box[0]; cr[k] = g.child[0]; ++k;
                cb[k] = g.box[1]; cr[k] = g.child[1]; ++k;
            }
        }
        const uint32_t idx = n_nodes++;
        uint32_t dmax = 0, smax = 0;
        Bvh4Node out;
        for (int i = 0; i < 4; ++i) {
            out.b[i].pad = 0;
            if (i >= k) { for (int a = 0; a < 3; ++a) { out.a[i].lo[a] = INFINITY; out.b[i].hi[a] = -INFINITY; } out.a[i].ref = RT_NO_CHILD; continue; }
            pad(cb[i], out.a[i].lo, out.b[i].hi);
            if (cr[i] & RT_LEAF_BIT) { out.a[i].ref = leaf_block(cr[i]); continue; }
            uint32_t d = 0, st = 0;
            out.a[i].ref = collapse(cr[i], by_area, d, st);
            dmax = std::max(dmax, d); smax = std::max(smax, st);
        }
        nodes[idx] = out;
        depth = dmax + 1;
        stack = smax + uint32_t(k - 1);
        return idx;
    }

    // Small ranges (<= 4 leaves' worth): a leaf costs the quad one step whether it holds 1 or 4 triangles, so only
    // splits that leave a multiple of RT_MAX_LEAF_TRIS on the left are considered (full leaves, fewer nodes); among
    // them the surface-area cost in leaf steps decides.
    uint32_t full_leaf_split(uint32_t b, uint32_t e) {
        const uint32_t n = e - b;
        float best_cost = INFINITY;
        int best_axis = 0;
        uint32_t best_c = RT_MAX_LEAF_TRIS;
        std::vector<uint32_t> tmp(order.begin() + b, order.begin() + e);
        for (int axis = 0; axis < 3; ++axis) {
            std::sort(tmp.begin(), tmp.end(), [&](uint32_t x, uint32_t y) {
                const float cx = centroid[3 * size_t(x) + axis], cy = centroid[3 * size_t(y) + axis];
                return cx < cy || (cx == cy && x < y);
            });
            std::vector<float> right_area(n + 1, 0.0f);      // surface of faces [c, n)
            Box acc; acc.reset();
            for (uint32_t i = n; i-- > 0;) { acc.grow(tri_box[tmp[i]]); right_area[i] = acc.half_area(); }
            acc.reset();
            for (uint32_t c = 1; c < n; ++c) {
                acc.grow(tri_box[tmp[c - 1]]);
                if (c % RT_MAX_LEAF_TRIS) continue;
                const float cost = acc.half_area() * float((c + RT_MAX_LEAF_TRIS - 1) / RT_MAX_LEAF_TRIS) +
                                   right_area[c] * float((n - c + RT_MAX_LEAF_TRIS - 1) / RT_MAX_LEAF_TRIS);
                if (cost < best_cost) { best_cost = cost; best_axis = axis; best_c = c; }
            }
        }
        std::sort(order.begin() + b, order.begin() + e, [&](uint32_t x, uint32_t y) {
            const float cx = centroid[3 * size_t(x) + best_axis], cy = centroid[3 * size_t(y) + best_axis];
            return cx < cy || (cx == cy && x < y);
        });
        return b + best_c;
    }

    uint32_t median_split(uint32_t b, uint32_t e, const Box& bx) {
        int axis = 0;
        float ext = -1;
        for (int a = 0; a < 3; ++a) if (bx.hi[a] - bx.lo[a] > ext) { ext = bx.hi[a] - bx.lo[a]; axis = a; }
        uint32_t half = ((e - b) / 2 + RT_MAX_LEAF_TRIS - 1) / RT_MAX_LEAF_TRIS * RT_MAX_LEAF_TRIS;   // keeps leaves full
        if (half >= e - b) half = (e - b) / 2;
        const uint32_t mid = b + half;
        std::nth_element(order.begin() + b, order.begin() + mid, order.begin() + e, [&](uint32_t x, uint32_t y) {
            const float cx = centroid[3 * size_t(x) + axis], cy = centroid[3 * size_t(y) + axis];
            return cx < cy || (cx == cy && x < y);
        });
        return mid;
    }

    uint32_t sah_split(uint32_t b, uint32_t e, const Box& /*bx*/) {
        Box cb; cb.reset();
        for (uint32_t i = b; i < e; ++i) cb.grow(&centroid[3 * size_t(order[i])]);
        float best_cost = INFINITY;
        int best_axis = -1, best_bin = -1;
        for (int axis = 0; axis < 3; ++axis) {
            const float lo = cb.lo[axis], ext = cb.hi[axis] - lo;
            if (!(ext > 0)) continue;
            const float scale = NBINS / ext;
            Box bins[NBINS];
            uint32_t cnt[NBINS] = {0};
            for (auto& x : bins) x.reset();
            for (uint32_t i = b; i < e; ++i) {
                const uint32_t f = order[i];
                int k = int((centroid[3 * size_t(f) + axis] - lo) * scale);
                k = k < 0 ? 0 : (k >= NBINS ? NBINS - 1 : k);
                bins[k].grow(tri_box[f]);
                ++cnt[k];
            }
            float right_area[NBINS];
            uint32_t right_cnt[NBINS];
            Box acc; acc.reset();
            uint32_t c = 0;
            for (int k = NBINS - 1; k > 0; --k) {
                if (cnt[k]) acc.grow(bins[k]);
                c += cnt[k];
                right_area[k] = c ? acc.half_area() : 0;
                right_cnt[k] = c;
            }
            acc.reset();
            c = 0;
            for (int k = 0; k < NBINS - 1; ++k) {
                if (cnt[k]) acc.grow(bins[k]);
                c += cnt[k];
                if (c == 0 || right_cnt[k + 1] == 0) continue;
                const float cost = acc.half_area() * float(c) + right_area[k + 1] * float(right_cnt[k + 1]);
                if (cost < best_cost) { best_cost = cost; best_axis = axis; best_bin = k; }
            }
        }
        if (best_axis < 0) return b;   // all centroids equal: caller falls back to the median split
        const float lo = cb.lo[best_axis], scale = NBINS / (cb.hi[best_axis] - lo);
        auto it = std::partition(order.begin() + b, order.begin() + e, [&](uint32_t f) {
            int k = int((centroid[3 * size_t(f) + best_axis] - lo) * scale);
            k = k < 0 ? 0 : (k >= NBINS ? NBINS - 1 : k);
            return k <= best_bin;
        });
        return uint32_t(it - order.begin());
    }
};

}  // namespace

int bvh_build(const rt_flat_scene* sc, BvhBuildResult* out) {
    std::memset(out, 0, sizeof *out);
    const uint32_t nf = sc->n_faces;
    Builder B;
    B.sc = sc;
    B.tri_box.resize(nf);
    B.centroid.resize(size_t(nf) * 3);
    B.order.resize(nf);
#pragma omp parallel for schedule(static)
    for (uint32_t f = 0; f < nf; ++f) {
        Box bx; bx.reset();
        bx.grow(sc->v0 + 3 * size_t(f));
        bx.grow(sc->v1 + 3 * size_t(f));
        bx.grow(sc->v2 + 3 * size_t(f));
        B.tri_box[f] = bx;
        for (int a = 0; a < 3; ++a) B.centroid[3 * size_t(f) + a] = 0.5f * (bx.lo[a] + bx.hi[a]);
        B.order[f] = f;
    }
    // order[] must be a permutation within each model's own face range
    for (uint32_t m = 0; m < sc->n_models; ++m) {
        const rt_model& md = sc->models[m];
        if (uint64_t(md.first_face) + md.num_faces > nf) return -1;
    }
    out->nodes = static_cast<Bvh4Node*>(std::malloc(sizeof(Bvh4Node) * std::max<size_t>(1, nf)));
    out->model_root = static_cast<uint32_t*>(std::malloc(sizeof(uint32_t) * std::max<uint32_t>(1, sc->n_models)));
    if (!out->nodes || !out->model_root) { bvh_free(out); return -2; }
    B.nodes = out->nodes;
    B.nodes2.resize(std::max<size_t>(1, nf));
    uint32_t max_depth = 0, max_stack = 0;
    for (uint32_t m = 0; m < sc->n_models; ++m) {
        const rt_model& md = sc->models[m];
        if (md.num_faces == 0) { out->model_root[m] = RT_NO_CHILD; continue; }
        // SAH tree first; if its 4-wide collapse could overflow the traversal stack (pathological geometry), rebuild
        // this model balanced (median splits, two binary levels per 4-wide level), which cannot.
        for (int attempt = 0; attempt < 2; ++attempt) {
            B.balanced = attempt == 1;
            B.n_nodes2 = 0;
            const uint32_t nodes_before = B.n_nodes;
            const size_t leaves_before = B.leaf_first.size();
            Box bx;
            uint32_t root2 = 0;
#pragma omp parallel
#pragma omp single
            root2 = B.build(md.first_face, md.first_face + md.num_faces, 0, bx);
            uint32_t depth = 0, stack = 0, root = root2;
            if (!(root2 & RT_LEAF_BIT)) root = B.collapse(root2, !B.balanced, depth, stack);
            else root = B.leaf_block(root2);
            if (stack + 1 > RT_QSTACK && attempt == 0) { B.n_nodes = nodes_before; B.leaf_first.resize(leaves_before); continue; }
            if (stack + 1 > RT_QSTACK) { bvh_free(out); return -3; }
            out->model_root[m] = root;
            max_depth = std::max(max_depth, depth);
            max_stack = std::max(max_stack, stack + 1);
            break;
        }
    }
    out->n_nodes = B.n_nodes;
    out->max_depth = max_depth;
    out->max_stack = max_stack;
    // leaf blocks; a block's triangle count is in every reference to it, so the block itself only needs the data
    // (unused slots are zero: never read, the quad tests lanes j < count only)
    const size_t nl = B.leaf_first.size();
    out->leaves = static_cast<LeafBlock*>(std::calloc(std::max<size_t>(1, nl), sizeof(LeafBlock)));
    if (!out->leaves) { bvh_free(out); return -2; }
    out->n_leaves = uint32_t(nl);
    std::vector<uint32_t> leaf_cnt(nl, 0);
    {   // counts: from the references (nodes and roots)
        auto note = [&](uint32_t ref) { if (ref != RT_NO_CHILD && (ref & RT_LEAF_BIT)) leaf_cnt[ref & RT_LEAF_START_MASK] = ((ref >> 27) & 15u) + 1u; };
        for (uint32_t n = 0; n < out->n_nodes; ++n) for (int i = 0; i < 4; ++i) note(out->nodes[n].a[i].ref);
        for (uint32_t m = 0; m < sc->n_models; ++m) note(out->model_root[m]);
    }
#pragma omp parallel for schedule(static)
    for (size_t l = 0; l < nl; ++l) {
        LeafBlock& lb = out->leaves[l];
        for (uint32_t j = 0; j < leaf_cnt[l]; ++j) {
            const uint32_t f = B.order[B.leaf_first[l] + j];
            const float* a = sc->v0 + 3 * size_t(f);
            const float* b = sc->v1 + 3 * size_t(f);
            const float* c = sc->v2 + 3 * size_t(f);
            for (int k = 0; k < 3; ++k) {
                lb.a[j].v0[k] = a[k];
                lb.b[j].e1[k] = b[k] - a[k];   // v0v1, reference src/primitives.cpp:52
                lb.c[j].e2[k] = c[k] - a[k];   // v0v2, reference src/primitives.cpp:53
            }
            lb.a[j].face = f;
        }
    }
    return 0;
}

void bvh_free(BvhBuildResult* r) {
    std::free(r->nodes);
    std::free(r->leaves);
    std::free(r->model_root);
    std::memset(r, 0, sizeof *r);
}

// ---- rt_bvh_check: build on the host (no GPU), then walk the tree and verify what the traversal relies on ----
namespace {
struct Checker {
    const rt_flat_scene* sc;
    const BvhBuildResult* r;
    std::vector<uint8_t> seen;     // per face
    uint64_t errors = 0;
    uint32_t depth_seen = 0;
    // returns the exact box of the vertices below `ref`
    Box walk(uint32_t ref, const rt_model& md, uint32_t depth) {
        Box bx; bx.reset();
        if (ref & RT_LEAF_BIT) {
            const uint32_t blk = ref & RT_LEAF_START_MASK, cnt = ((ref >> 27) & 15u) + 1u;
            if (blk >= r->n_leaves || cnt > RT_MAX_LEAF_TRIS) { ++errors; return bx; }
            const LeafBlock& lb = r->leaves[blk];
            for (uint32_t j = 0; j < cnt; ++j) {
                const uint32_t f = lb.a[j].face;
                if (f < md.first_face || f >= md.first_face + md.num_faces) { ++errors; continue; }
                if (seen[f]++) ++errors;                       // a face in two leaves
                const float* a = sc->v0 + 3 * size_t(f);
                const float* b = sc->v1 + 3 * size_t(f);
                const float* c = sc->v2 + 3 * size_t(f);
                for (int k = 0; k < 3; ++k)                    // the arithmetic Triangle::intersect would redo
                    if (lb.a[j].v0[k] != a[k] || lb.b[j].e1[k] != b[k] - a[k] || lb.c[j].e2[k] != c[k] - a[k]) { ++errors; break; }
                bx.grow(a); bx.grow(b); bx.grow(c);
            }
            return bx;
        }
        if (ref >= r->n_nodes) { ++errors; return bx; }
        depth_seen = std::max(depth_seen, depth + 1);
        const Bvh4Node& n = r->nodes[ref];
        bool gap = false;
        for (int i = 0; i < 4; ++i) {
            if (n.a[i].ref == RT_NO_CHILD) { gap = true; continue; }
            if (gap) ++errors;                                 // children are packed to the front
            const Box cb = walk(n.a[i].ref, md, depth + 1);
            for (int k = 0; k < 3; ++k)
                if (!(n.a[i].lo[k] <= cb.lo[k] && n.b[i].hi[k] >= cb.hi[k])) { ++errors; break; }
            bx.grow(cb);
        }
        return bx;
    }
};
}  // namespace

// Walks a finished tree (host arrays) and fills `out` (everything but build_seconds).
void bvh_walk_check(const rt_flat_scene* sc, const BvhBuildResult& r, rt_bvh_check_result* out) {
    Checker C{sc, &r, std::vector<uint8_t>(sc->n_faces, 0)};
    for (uint32_t m = 0; m < sc->n_models; ++m) {
        const rt_model& md = sc->models[m];
        if (md.num_faces == 0) { if (r.model_root[m] != RT_NO_CHILD) ++C.errors; continue; }
        C.walk(r.model_root[m], md, 0);
        for (uint32_t f = md.first_face; f < md.first_face + md.num_faces; ++f) if (C.seen[f] != 1) ++C.errors;
    }
    if (C.depth_seen > r.max_depth) ++C.errors;      // (the device builder reports a bound, the host builder the exact depth)
    if (r.max_stack > RT_QSTACK) ++C.errors;
    out->n_nodes = r.n_nodes; out->n_leaves = r.n_leaves; out->n_faces = sc->n_faces;
    out->max_depth = r.max_depth; out->max_stack = r.max_stack;
    out->errors = C.errors;
    out->host_bytes = uint64_t(r.n_nodes) * sizeof(Bvh4Node) + uint64_t(r.n_leaves) * sizeof(LeafBlock);
}

extern "C" int rt_bvh_check(const rt_flat_scene* sc, rt_bvh_check_result* out) {
    if (!sc || !out) return RT_ERR_INVALID;
    std::memset(out, 0, sizeof *out);
    BvhBuildResult r;
    const auto t0 = std::chrono::steady_clock::now();
    const int rc = bvh_build(sc, &r);
    out->build_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (rc != 0) return RT_ERR_INVALID;
    bvh_walk_check(sc, r, out);
    bvh_free(&r);
    return RT_OK;
}
