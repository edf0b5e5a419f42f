// layout_check.cpp -- compile-time proof that include/cobbletrace_host.h has the
// reference's struct layouts (sizes measured from the reference declarations:
// scenefile.h:19-108, environment.h:7-20, eventQueue.h:26-39, bvh.h:5-25 on LP64).
#include "cobbletrace_host.h"
#include <cstddef>

static_assert(sizeof(v3_t) == 24 && sizeof(m3x3_t) == 72, "math carriers");
static_assert(sizeof(material_t) == 12, "material_t");
static_assert(sizeof(triangle_t) == 96, "triangle_t");
static_assert(sizeof(import_t) == 96, "import_t");
static_assert(sizeof(scene_object_t) == 120, "scene_object_t (SURVEY.md 3.1: 1,000,000 x 120 B)");
static_assert(offsetof(scene_object_t, triangle) == 8 && offsetof(scene_object_t, material) == 104, "scene_object_t");
static_assert(sizeof(light_t) == 56, "light_t");
static_assert(sizeof(camera_t) == 96, "camera_t");
static_assert(sizeof(scene_stack_t) == 24, "scene_stack_t");
static_assert(sizeof(settings_t) == 8, "settings_t");
static_assert(sizeof(scene_t) == 24 + 24 + 16 + 96 + 12 + 8 + 4, "scene_t");
static_assert(sizeof(ray_t) == 56, "ray_t");
static_assert(sizeof(event_t) == 32 && sizeof(eventManager_t) == 24, "event queue");
static_assert(sizeof(bitmapSettings_t) == 24 && sizeof(environment_t) == 48, "environment");
static_assert(sizeof(bvh_node_t) == 64, "bvh_node_t");
static_assert(sizeof(bvh_state_t) == 40, "bvh_state_t");
