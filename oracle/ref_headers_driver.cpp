// ref_headers_driver.cpp -- build-owned shim that compiles the reference's two
// self-contained headers (mymath.h, color.h) IN PLACE from /root/reference and
// exports their functions with C linkage, so tests can compare the oracle's
// restatement with the real thing call by call.  TEST INFRASTRUCTURE ONLY.
//
// Only these headers build without SDL2 (absent from the image); the renderer
// proper does not (see ct_oracle.h).  Output goes to oracle/_ref/ (git-ignored).
#include <cstdint>
#include <cmath>
#include <vector>     // (before the reference's headers: mymath.h defines min / max as macros)
#include "color.h"      // from -I/root/reference; pulls in mymath.h
#include "eventQueue.h" // the key-event ring RayThread drains; eventQueue.cpp is compiled in place beside this file

extern "C" {

void ref_color_to_hsv(uint32_t color, float out[3]) {
    hsv_t h = ColorToHsv(color);
    out[0] = h.h; out[1] = h.s; out[2] = h.v;
}

uint32_t ref_hsv_to_color(const float in[3]) {
    hsv_t h; h.h = in[0]; h.s = in[1]; h.v = in[2];
    return HsvToColor(h);
}

uint32_t ref_shade_value(uint32_t color, float value) {
    hsv_t h = ColorToHsv(color);
    h.v = value;
    return HsvToColor(h);
}

float ref_dot(const double a[3], const double b[3]) {
    return DotProduct(v3_t{a[0], a[1], a[2]}, v3_t{b[0], b[1], b[2]});
}

float ref_magnitude(const double a[3]) { return Magnitude(v3_t{a[0], a[1], a[2]}); }

void ref_cross(const double a[3], const double b[3], double out[3]) {
    v3_t c = CrossProduct(v3_t{a[0], a[1], a[2]}, v3_t{b[0], b[1], b[2]});
    out[0] = c.x; out[1] = c.y; out[2] = c.z;
}

void ref_vec_times_mat(const double v[3], const double m[9], double out[3]) {
    m3x3_t mm;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) mm.data[i][j] = m[i * 3 + j];
    v3_t r = v3_t{v[0], v[1], v[2]} * mm;
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}

uint32_t ref_mix_reflection(uint32_t local, uint32_t reflected, float reflection) {
    v3_t a = ColorToRgbV3(local), b = ColorToRgbV3(reflected);
    v3_t c = a * (1 - reflection) + b * reflection;
    return RgbToColor(c.x, c.y, c.z);
}

uint32_t ref_blend_sample(uint32_t color, uint32_t tempColor) {
    float denom = 8;
    rgb_t rgb = ColorToRgb(color);
    rgb_t tempRgb = ColorToRgb(tempColor);
    rgb.r -= (rgb.r/denom); rgb.r += (tempRgb.r/denom);
    rgb.g -= (rgb.g/denom); rgb.g += (tempRgb.g/denom);
    rgb.b -= (rgb.b/denom); rgb.b += (tempRgb.b/denom);
    return RgbToColor(rgb);
}


// A script of operations on the reference's event ring (eventQueue.cpp:5-24): op >= 0 -> AddEvent of a key-down
// with that value; op == -1 -> ReadEvent, whose value (0xffffffff for NULL) is appended to out.  Returns the
// number of values written.
uint32_t ref_event_queue_script(uint32_t capacity, const int32_t *ops, uint32_t n, uint32_t *out) {
    std::vector<event_t> ring(capacity);
    eventManager_t em{};
    em.queue = ring.data();
    em.capacity = capacity;
    uint32_t written = 0;
    for (uint32_t i = 0; i < n; i++) {
        if (ops[i] >= 0) {
            AddEvent(&em, event_t{ET_KEY_DOWN, EM_NONE, v2_t{0, 0}, (uint32_t)ops[i]});
        } else {
            event_t *e = ReadEvent(&em);
            out[written++] = e ? e->value : 0xffffffffu;
        }
    }
    return written;
}
}
