// scene_pack.h -- turns a parsed scene_t into the flat arrays of ctgpu_scene_desc.
// This is the work RayThread does once, on its first call (raythread.cpp:647-654:
// GetSceneTriangles -> InitializeBVHState -> BuildBVH), plus the repacking the
// C ABI asks for.
#ifndef CT_SCENE_PACK_H
#define CT_SCENE_PACK_H

#include "cobbletrace_host.h"
#include "ctgpu.h"

#include <vector>

namespace ct {

struct PackedScene {
    std::vector<triangle_t>     triangles;   // the copy the BVH is built over (raythread.cpp:621-639)
    bvh_state_t                 bvh{};
    std::vector<double>         vertices;    // [n][9]
    std::vector<ctgpu_material> materials;   // [n]
    std::vector<ctgpu_light>    lights;
    // the scene's OT_SPHERE objects in stack order, and a tree over their bounding cubes (CTGPU_F_SPHERES)
    std::vector<ctgpu_sphere>   spheres;
    std::vector<triangle_t>     sphere_proxies;
    bvh_state_t                 sphere_bvh{};
    bool                        built = false;

    ~PackedScene();
    PackedScene() = default;
    PackedScene(const PackedScene &) = delete;
    PackedScene &operator=(const PackedScene &) = delete;

    // Gathers the OT_TRIANGLE objects in stack order, builds the BVH, fills the arrays.  Where the tree comes from:
    //   kHostTree        BuildBVH on the host (bvh_build.cpp)
    //   kDeviceTreeCopy  BuildBVH on that GPU, arrays copied back (ctgpu_build_bvh)
    //   kNoTree          none here: desc() carries vertices / materials / lights only and ctgpu_upload_scene builds the
    //                    tree in HBM, where it stays (nothing of it is ever in host memory)
    enum TreeSource { kHostTree, kDeviceTreeCopy, kNoTree };
    void build_from(const scene_t *scene, TreeSource tree = kHostTree, int device = 0);
    void build_tree_on_host();         // after kNoTree: the host tree after all (the device refused the mesh)
    // Same from bare vertices (n x 9 doubles), default material; used by the BvhTest KAT.
    void build_from_vertices(const double *verts, uint32_t n);
    ctgpu_scene_desc desc() const;
    uint32_t tree_depth() const;
};

}  // namespace ct
#endif
