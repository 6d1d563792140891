// bvh_device.h — GPU BVH builder (bvh_device.hip). Output arrays are DEVICE memory (hipMalloc), owned by the caller.
#pragma once
#include <stdint.h>
#include <vector>

#include "raytracer_amd.h"
#include "rt_device.h"

struct DeviceBvh {
    Bvh4Node* nodes = nullptr;   uint32_t n_nodes = 0;
    LeafBlock* leaves = nullptr; uint32_t n_leaves = 0;
    std::vector<uint32_t> model_root;      // per model: child reference of its root (RT_NO_CHILD for an empty model)
    uint32_t max_depth = 0;                // four-wide nodes on the longest path
    uint32_t max_stack = 1;                // bound on the traversal stack (3 per level + sentinel)
    double seconds = 0;                    // wall time of the build, vertex upload included
};

// Builds on the current HIP device. 0 = built; 1 = not applicable (a model's tree would be too deep for the traversal
// stack: use the host builder); < 0 = error (rt_last_error).
int bvh_build_device(const rt_flat_scene* scene, DeviceBvh* out);
