// Host BVH builder and flattener.  The tree is the reference's
// (src/bounding_volume_hierarchy.cpp:65-180): split axis cycles x,y,z with depth,
// triangles ordered by descending centroid, split at the median
// (std::nth_element) or at the minimum of the reference's 1-D sweep cost
// (std::sort + scan) when enableBvhSahBinning is set, leaves at
// ceil(0.8*log2 n) levels or below two triangles.  It is written straight into
// the 64-byte records of include/rtx.h that the device traverses.
#pragma once
#include <vector>

#include "rt_types.h"
#include "rtx.h"

namespace rt {

struct FlatScene {
    std::vector<rtx_node> nodes;
    std::vector<rtx_tri> tris;          // leaf order
    std::vector<rtx_tri_attr> attrs;    // same order
    std::vector<rtx_material> materials;
    std::vector<rtx_texture> textures;
    std::vector<float> texels;          // RGB triplets
    std::vector<rtx_light> lights;
    std::vector<rtx_sphere> spheres;
    rtx_envmap envmap { -1, RTX_ENV_SPHERICAL, 1.f, { 0, 0, 0 } };
    std::vector<float> envBins;         // radiance bins of the environment image, 4 floats each
    int32_t rootRef = INT32_MIN, rootCnt = 0;
    float rootLo[3] = { 0, 0, 0 }, rootHi[3] = { 0, 0, 0 };
    int numLevels = 0, numLeaves = 0;
    // meshTrianglePairs order (mesh-major): which mesh / local triangle a global id is
    std::vector<int32_t> idMesh, idLocal;

    rtx_scene_desc desc() const;
};

// Build the tree over scene.meshes and gather materials, textures (with their
// mip chains, src/main.cpp:745-771), lights, spheres and the environment map.
// With buildTree == false no tree is made: tris / attrs come in meshTrianglePairs (id) order
// and rootRef stays empty, which is what rtx_scene_build_device takes to build its own tree
// on the GPU.
FlatScene flattenScene(const Scene& scene, const Features& features, bool buildTree = true);

} // namespace rt
