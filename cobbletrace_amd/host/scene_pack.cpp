#include "scene_pack.h"
#include "host_internal.h"

#include <cstddef>
#include <cstdlib>
#include <cstring>

// The C ABI node is the reference node, byte for byte (bvh.h:5-11).
static_assert(sizeof(bvh_node_t) == sizeof(ctgpu_node), "node size");
static_assert(offsetof(bvh_node_t, aabbMax) == offsetof(ctgpu_node, bmax), "node layout");
static_assert(offsetof(bvh_node_t, leftNode) == offsetof(ctgpu_node, left), "node layout");
static_assert(offsetof(bvh_node_t, firstTriangleIndex) == offsetof(ctgpu_node, first), "node layout");
static_assert(offsetof(bvh_node_t, triangleCount) == offsetof(ctgpu_node, count), "node layout");
static_assert(sizeof(light_t) == sizeof(ctgpu_light), "light size");
static_assert(sizeof(material_t) == sizeof(ctgpu_material), "material size");

namespace ct {

PackedScene::~PackedScene() {
    std::free(bvh.bvhNodes);
    std::free(bvh.triangles.indexes);
    std::free(sphere_bvh.bvhNodes);
    std::free(sphere_bvh.triangles.indexes);
}

// The sphere tree is the triangle builder's work on stand-ins: per sphere a "triangle" whose corners are the two
// opposite corners of its bounding cube and its centre, so that BuildBVH's own bounds, centroids (of the three
// corners, x 0.3333f as always) and midpoint splits apply unchanged (bvh.cpp:16-121).
static void build_sphere_tree(PackedScene &p) {
    std::free(p.sphere_bvh.bvhNodes);
    std::free(p.sphere_bvh.triangles.indexes);
    p.sphere_bvh = bvh_state_t{};
    p.sphere_proxies.clear();
    if (p.spheres.empty()) return;
    for (const ctgpu_sphere &s : p.spheres) {
        const double r = (double)s.radius;
        triangle_t t{};
        t.p1 = v3_t{s.center[0] - r, s.center[1] - r, s.center[2] - r};
        t.p2 = v3_t{s.center[0] + r, s.center[1] + r, s.center[2] + r};
        t.p3 = v3_t{s.center[0], s.center[1], s.center[2]};
        p.sphere_proxies.push_back(t);
    }
    p.sphere_bvh = InitializeBVHState(p.sphere_proxies.data(), (int)p.sphere_proxies.size());
    BuildBVH(&p.sphere_bvh);
}

static void flatten(PackedScene &p, PackedScene::TreeSource tree = PackedScene::kHostTree, int device = 0) {
    const size_t n = p.triangles.size();
    p.vertices.resize(n * 9);
    for (size_t i = 0; i < n; i++) {
        const triangle_t &t = p.triangles[i];
        const double v[9] = {t.p1.x, t.p1.y, t.p1.z, t.p2.x, t.p2.y, t.p2.z, t.p3.x, t.p3.y, t.p3.z};
        std::memcpy(&p.vertices[i * 9], v, sizeof v);
    }
    std::free(p.bvh.bvhNodes);
    std::free(p.bvh.triangles.indexes);
    p.bvh = bvh_state_t{};
    p.built = false;
    if (tree == PackedScene::kNoTree) return;          // (not even InitializeBVHState's 2N-node calloc)
    p.bvh = InitializeBVHState(p.triangles.data(), (int)n);
    if (tree == PackedScene::kDeviceTreeCopy && n > 0) {
        // (BuildBVH also leaves each triangle's centroid in the caller's array, bvh.cpp:110-112: nothing reads it afterwards)
        p.bvh.nodesUsed = build_bvh_on_device(device, p.vertices.data(), (uint32_t)n, p.bvh.bvhNodes, p.bvh.triangles.indexes);
    } else {
        BuildBVH(&p.bvh);
    }
    p.built = true;
}

void PackedScene::build_tree_on_host() {
    if (built) return;
    bvh = InitializeBVHState(triangles.data(), (int)triangles.size());
    BuildBVH(&bvh);
    built = true;
}

void PackedScene::build_from(const scene_t *scene, TreeSource tree, int device) {
    triangles.clear();
    materials.clear();
    lights.clear();
    spheres.clear();
    for (int i = 0; i < scene->objectStack.index; i++) {
        const scene_object_t &o = scene->objectStack.objects[i];
        if (o.type == OT_SPHERE) {
            ctgpu_sphere s;
            std::memset(&s, 0, sizeof s);
            s.center[0] = o.sphere.center.x; s.center[1] = o.sphere.center.y; s.center[2] = o.sphere.center.z;
            s.radius = o.sphere.radius;
            s.material = ctgpu_material{o.material.color, o.material.specular, o.material.reflection};
            spheres.push_back(s);
        }
        if (o.type != OT_TRIANGLE) continue;
        triangles.push_back(o.triangle);
        materials.push_back(ctgpu_material{o.material.color, o.material.specular, o.material.reflection});
    }
    for (int i = 0; i < scene->lightStack.index; i++) {
        const light_t &l = scene->lightStack.lights[i];
        ctgpu_light g;
        std::memset(&g, 0, sizeof g);
        g.type = (int32_t)l.type;
        g.intensity = l.intensity;
        g.position[0] = l.position.x; g.position[1] = l.position.y; g.position[2] = l.position.z;
        g.direction[0] = l.direction.x; g.direction[1] = l.direction.y; g.direction[2] = l.direction.z;
        lights.push_back(g);
    }
    build_sphere_tree(*this);
    flatten(*this, tree, device);
}

void PackedScene::build_from_vertices(const double *verts, uint32_t n) {
    triangles.assign(n, triangle_t{});
    materials.assign(n, ctgpu_material{0x00ffffffu, -1, 0.0f});
    lights.clear();
    spheres.clear();
    build_sphere_tree(*this);
    for (uint32_t i = 0; i < n; i++) {
        const double *v = verts + (size_t)i * 9;
        triangles[i].p1 = v3_t{v[0], v[1], v[2]};
        triangles[i].p2 = v3_t{v[3], v[4], v[5]};
        triangles[i].p3 = v3_t{v[6], v[7], v[8]};
    }
    flatten(*this);
}

ctgpu_scene_desc PackedScene::desc() const {
    ctgpu_scene_desc d;
    std::memset(&d, 0, sizeof d);
    d.n_triangles = (uint32_t)triangles.size();
    d.n_nodes = built ? bvh.nodesUsed : 0u;            // no tree here: ctgpu_upload_scene builds it on the device
    d.n_lights = (uint32_t)lights.size();
    d.vertices = vertices.data();
    d.materials = materials.data();
    d.nodes = built ? reinterpret_cast<const ctgpu_node *>(bvh.bvhNodes) : nullptr;
    d.indexes = built ? bvh.triangles.indexes : nullptr;
    d.lights = lights.data();
    if (!spheres.empty()) {
        d.n_spheres = (uint32_t)spheres.size();
        d.n_sphere_nodes = sphere_bvh.nodesUsed;
        d.spheres = spheres.data();
        d.sphere_nodes = reinterpret_cast<const ctgpu_node *>(sphere_bvh.bvhNodes);
        d.sphere_indexes = sphere_bvh.triangles.indexes;
    }
    return d;
}

uint32_t PackedScene::tree_depth() const {
    // explicit stack: (node, depth)
    if (!built || triangles.empty()) return 0;
    std::vector<std::pair<uint32_t, uint32_t>> todo{{0u, 1u}};
    uint32_t deepest = 0;
    while (!todo.empty()) {
        auto [id, depth] = todo.back();
        todo.pop_back();
        if (depth > deepest) deepest = depth;
        const bvh_node_t &n = bvh.bvhNodes[id];
        if (n.triangleCount == 0) {
            todo.push_back({n.leftNode, depth + 1});
            todo.push_back({n.leftNode + 1, depth + 1});
        }
    }
    return deepest;
}

}  // namespace ct
