// host_internal.h -- helpers shared by the host translation units (not installed).
#ifndef CT_HOST_INTERNAL_H
#define CT_HOST_INTERNAL_H

#include <cstdint>
#include <cstddef>
#include <string>

namespace ct {

// 0x00BBGGRR packing of three 8-bit channels (color.h:77-80).
inline uint32_t pack_rgb(uint8_t r, uint8_t g, uint8_t b) {
    return ((uint32_t)b << 16) | ((uint32_t)g << 8) | (uint32_t)r;
}

// FNV-1a-64 used by the known-answer tests.
struct Fnv64 {
    uint64_t h = 0xcbf29ce484222325ull;
    void bytes(const void *p, size_t n) {
        const unsigned char *b = (const unsigned char *)p;
        for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 0x100000001b3ull; }
    }
};

extern std::string g_scene_dir;

// BuildBVH through libctgpu.so's ctgpu_build_bvh (ray_thread.cpp): fills `nodes` (room for 2n - 1) and `indexes`
// (n) with the arrays the host builder would produce and returns the number of nodes; exits like every other
// fatal GPU condition when the library or a device is missing.
uint32_t build_bvh_on_device(int device, const double *vertices, uint32_t n, void *nodes, uint32_t *indexes);

}  // namespace ct
#endif
