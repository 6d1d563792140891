// rt_device.h — data layout shared by the host API (rt_api.cpp), the BVH builder (bvh_build.cpp) and the
// HIP kernels (rt_kernels.hip). Plain structs only.
#pragma once
#include <stdint.h>

#include "raytracer_amd.h"

// ---- BVH node, 128 B = one gfx950 cache line: 4 children of 32 B each ----
// A ray is traversed by a QUAD of lanes: lane j of the quad fetches child j (two 16-B loads, the quad's 8 loads cover
// exactly this line), tests its box, and the quad orders the hit children with cross-lane moves. Leaves hold up to 4
// triangles, one per lane of the quad. Child reference: bit 31 clear -> index of another Bvh4Node; bit 31 set -> leaf:
//   bits 0..26 = leaf block index, bits 27..30 = triangle count-1.
// Unused child slots have ref = RT_NO_CHILD.
// Layout inside the node: the four {lo, ref} quarters first, then the four {hi, pad} quarters, so that each of the
// quad's two 16-B loads reads 64 CONTIGUOUS bytes (lane j at +16j and +64+16j): one fully coalesced request per quad.
struct Bvh4Node {
    struct { float lo[3]; uint32_t ref; } a[4];
    struct { float hi[3]; uint32_t pad; } b[4];
};
static_assert(sizeof(Bvh4Node) == 128, "Bvh4Node must be 128 bytes");

#define RT_LEAF_BIT 0x80000000u
#define RT_LEAF_START_MASK 0x07FFFFFFu
#define RT_MAX_LEAF_TRIS 4
#define RT_BVH2_MAX_DEPTH 30     // depth limit of the binary tree the 4-wide tree is collapsed from
#define RT_QSTACK 64             // traversal stack entries per ray (LDS); the builder guarantees the tree never needs more
#define RT_NO_CHILD 0xFFFFFFFFu

// ---- leaf block, 192 B: the (up to) 4 triangles of one leaf, laid out like a node: four {v0, face} quarters, four
// e1 quarters, four e2 quarters (e1 = v1-v0, e2 = v2-v0). Lane j of the quad tests triangle j with three 16-B loads,
// each contiguous over the quad. A leaf reference holds the BLOCK index. ----
// e1/e2 are the float differences Triangle::intersect forms on every call (reference src/primitives.cpp:52-53);
// forming them once on the host gives the same bits.
struct LeafBlock {
    struct { float v0[3]; uint32_t face; } a[4];   // face = index into the ORIGINAL face arrays (tie-break + normals)
    struct { float e1[3]; float pad; } b[4];
    struct { float e2[3]; float pad; } c[4];
};
static_assert(sizeof(LeafBlock) == 192, "LeafBlock must be 192 bytes");

// ---- per-face vertex normals in ORIGINAL face order, 48 B, fetched once per closest hit ----
struct TriNormals {
    float n0[3]; float pad0;
    float n1[3]; float pad1;
    float n2[3]; float pad2;
};

struct DModel {
    uint32_t root;        // child reference of the model's BVH root (may itself be a leaf)
    uint32_t first_face;
    uint32_t num_faces;
    float bb_center[3];
    float bb_radius;      // < 0: no gate
    uint32_t pad;
};
static_assert(sizeof(DModel) == 32, "DModel must be 32 bytes");

// Host-side result of the BVH build for all models of a scene.
struct BvhBuildResult {
    Bvh4Node* nodes;       uint32_t n_nodes;
    LeafBlock* leaves;     uint32_t n_leaves;
    uint32_t* model_root;                       // per model
    uint32_t max_depth;                         // interior nodes on the longest root-to-leaf path
    uint32_t max_stack;                         // worst-case traversal stack entries (incl. the sentinel)
};

// Builds one BVH per model (binned SAH binary tree, <= RT_MAX_LEAF_TRIS per leaf, collapsed to 4 children per node;
// worst-case traversal stack <= RT_QSTACK) into one shared node array. Returns 0 on success. Caller frees with bvh_free.
int bvh_build(const rt_flat_scene* scene, BvhBuildResult* out);
void bvh_free(BvhBuildResult* r);
struct rt_bvh_check_result;
// Walks a finished tree (host arrays) and fills `out` (everything but build_seconds); see rt_bvh_check.
void bvh_walk_check(const rt_flat_scene* scene, const BvhBuildResult& r, rt_bvh_check_result* out);
