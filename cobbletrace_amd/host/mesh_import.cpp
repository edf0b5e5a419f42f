// mesh_import.cpp -- OBJ / ASCII-PLY import behind ImportObject (objectLoader.h:4).
//
// What must hold for parity (objectLoader.cpp:16-288): every face becomes one
// OT_TRIANGLE object carrying the import record's material, with vertices
// transformed in double by row-vector x 4x4 products applied in the order
// rotate-Y, rotate-X, rotate-Z, scale, translate, each product summed left to
// right over all four terms; vertex and face numbers are parsed in binary32
// (TextCursor::number) and widened.
#include "cobbletrace_host.h"
#include "host_internal.h"
#include "text_cursor.h"

#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace {

using ct::TextCursor;

struct Affine { double m[4][4]; };

// (x y z 1) * M, w dropped (objectLoader.cpp:16-25 with mymath.h:58-66)
inline v3_t transform(const Affine &a, const v3_t &p) {
    const double w = 1.0;
    v3_t r;
    r.x = p.x * a.m[0][0] + p.y * a.m[1][0] + p.z * a.m[2][0] + w * a.m[3][0];
    r.y = p.x * a.m[0][1] + p.y * a.m[1][1] + p.z * a.m[2][1] + w * a.m[3][1];
    r.z = p.x * a.m[0][2] + p.y * a.m[1][2] + p.z * a.m[2][2] + w * a.m[3][2];
    return r;
}

Affine identity() {
    Affine a;
    std::memset(&a, 0, sizeof a);
    for (int i = 0; i < 4; i++) a.m[i][i] = 1.0;
    return a;
}

// The five placement matrices of objectLoader.cpp:29-119.
struct Placement {
    Affine rot_y, rot_x, rot_z, scale, translate;

    Placement(const v3_t &pos, const v3_t &rot, const v3_t &scl) {
        rot_x = identity();
        rot_x.m[1][1] = std::cos(rot.x);  rot_x.m[1][2] = -std::sin(rot.x);
        rot_x.m[2][1] = std::sin(rot.x);  rot_x.m[2][2] = std::cos(rot.x);

        rot_y = identity();
        rot_y.m[0][0] = std::cos(rot.y);  rot_y.m[0][2] = std::sin(rot.y);
        rot_y.m[2][0] = -std::sin(rot.y); rot_y.m[2][2] = std::cos(rot.y);

        rot_z = identity();
        rot_z.m[0][0] = std::cos(rot.z);  rot_z.m[0][1] = -std::sin(rot.z);
        rot_z.m[1][0] = std::sin(rot.z);  rot_z.m[1][1] = std::cos(rot.z);

        scale = identity();
        scale.m[0][0] = scl.x; scale.m[1][1] = scl.y; scale.m[2][2] = scl.z;

        translate = identity();
        translate.m[3][0] = pos.x; translate.m[3][1] = pos.y; translate.m[3][2] = pos.z;
        translate.m[3][3] = 0.0;   // objectLoader.cpp:119; w is discarded anyway
    }

    v3_t apply(v3_t p) const {
        p = transform(rot_y, p);
        p = transform(rot_x, p);
        p = transform(rot_z, p);
        p = transform(scale, p);
        p = transform(translate, p);
        return p;
    }
};

void emit_triangle(scene_t *scene, const scene_object_t &record, const Placement &place,
                   const v3_t &a, const v3_t &b, const v3_t &c, TextCursor &tc) {
    scene_object_t obj = record;      // keeps the material; the union is overwritten below
    obj.type = OT_TRIANGLE;
    obj.triangle.p1 = place.apply(a);
    obj.triangle.p2 = place.apply(b);
    obj.triangle.p3 = place.apply(c);
    if (scene->objectStack.index >= scene->objectStack.size) tc.fatal("object stack full");
    std::memcpy(&scene->objectStack.objects[scene->objectStack.index++], &obj, sizeof obj);
    scene->triangleLookup.triangleCount++;
}

std::string resolve(const scene_object_t &record) {
    std::string name(record.import.filename.data ? record.import.filename.data : "",
                     (size_t)record.import.filename.size);
    if (FILE *f = std::fopen(name.c_str(), "rb")) { std::fclose(f); return name; }
    // not under the working directory: look beside the scene file
    std::string beside = ct::g_scene_dir + "/" + name;
    if (FILE *f = std::fopen(beside.c_str(), "rb")) { std::fclose(f); return beside; }
    return name;
}

// ---- the body of a large PLY file on several threads -------------------------------------------
// The reference reads the body record by record through one cursor: three numbers, rest of the line
// skipped (objectLoader.cpp:160-191).  When every record's numbers stand on a line of their own -- any
// file a mesh tool writes -- record i is simply the i-th non-blank line, and the lines can be parsed
// independently with the same TextCursor::number().  This function finds the line starts (threads over
// byte ranges), parses vertices and then faces (threads over line ranges), and CHECKS the assumption on
// every record: a number that ran past its line's end, a record count that does not match, a face the
// serial code would reject -- and it gives up (returns false, nothing written) so that the serial code
// below does whatever the reference does with such a file.
unsigned loader_threads() {
    if (const char *t = std::getenv("CT_LOAD_THREADS")) return (unsigned)std::max(1, std::atoi(t));
    const unsigned hw = std::thread::hardware_concurrency();
    return std::min(hw ? hw : 1u, 16u);
}

template <typename Fn>
void run_threads(unsigned n, Fn fn) {
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < n; t++) pool.emplace_back(fn, t);
    fn(0u);
    for (std::thread &th : pool) th.join();
}

bool import_ply_body_parallel(const TextCursor &tc, const scene_object_t &record, const Placement &place, scene_t *scene,
                              uint32_t n_vertices, uint32_t n_faces) {
    const size_t n_records = (size_t)n_vertices + n_faces;
    const unsigned n_threads = loader_threads();
    if (n_threads < 2 || n_records < 20000) return false;
    if ((size_t)scene->objectStack.index + n_faces > (size_t)scene->objectStack.size) return false;   // serial code reports it
    const char *bytes = tc.bytes();
    const size_t body = tc.position(), size = tc.size();
    if (body >= size) return false;
    const bool timing = std::getenv("CT_LOAD_TIMING") != nullptr;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
        return std::chrono::duration<double, std::milli>(b - a).count();
    };
    const auto t0 = now();

    // 1. record starts: a non-space byte whose preceding white space holds a line break (the first one is `body`)
    std::vector<std::vector<size_t>> found(n_threads);
    run_threads(n_threads, [&](unsigned t) {
        const size_t lo = body + (size - body) * t / n_threads, hi = body + (size - body) * (t + 1) / n_threads;
        std::vector<size_t> &out = found[t];
        out.reserve((hi - lo) / 16 + 16);
        for (size_t p = lo; p < hi; p++) {
            if (TextCursor::is_space(bytes[p])) continue;
            if (p == body) { out.push_back(p); continue; }
            if (!TextCursor::is_space(bytes[p - 1])) continue;
            for (size_t q = p; q > body && TextCursor::is_space(bytes[q - 1]); q--)
                if (bytes[q - 1] == '\n' || bytes[q - 1] == '\r') { out.push_back(p); break; }
        }
    });
    std::vector<size_t> starts;
    starts.reserve(std::min(n_records, size - body) + 1);     // (a record is at least one byte: the header's counts do not size memory)
    for (const std::vector<size_t> &part : found) starts.insert(starts.end(), part.begin(), part.end());
    if (starts.size() < n_records) return false;          // the file ends early: the serial code's business
    auto line_end = [&](size_t from) {
        while (from < size && bytes[from] != '\n' && bytes[from] != '\r') from++;
        return from;
    };

    const auto t1 = now();
    // 2. vertices, then 3. faces straight into their object slots
    std::atomic<bool> clean{true};
    std::vector<v3_t> vertices(n_vertices);
    run_threads(n_threads, [&](unsigned t) {
        const size_t lo = (size_t)n_vertices * t / n_threads, hi = (size_t)n_vertices * (t + 1) / n_threads;
        TextCursor line(tc, 0);                 // one view per worker (sharing the bytes), moved from record to record
        for (size_t i = lo; i < hi && clean.load(std::memory_order_relaxed); i++) {
            line.seek(starts[i]);
            double v[3];
            line.vec3_bare(v);
            if (line.position() > line_end(starts[i])) { clean = false; break; }
            vertices[i] = v3_t{v[0], v[1], v[2]};
        }
    });
    if (!clean) return false;
    const auto t2 = now();
    scene_object_t *slots = scene->objectStack.objects + scene->objectStack.index;
    run_threads(n_threads, [&](unsigned t) {
        const size_t lo = (size_t)n_faces * t / n_threads, hi = (size_t)n_faces * (t + 1) / n_threads;
        TextCursor line(tc, 0);
        for (size_t i = lo; i < hi && clean.load(std::memory_order_relaxed); i++) {
            line.seek(starts[n_vertices + i]);
            const int corners = TextCursor::to_int(line.number());
            double f[3];
            line.vec3_bare(f);
            const int ia = TextCursor::to_int(f[0]), ib = TextCursor::to_int(f[1]), ic = TextCursor::to_int(f[2]);
            if (line.position() > line_end(starts[n_vertices + i]) || corners != 3 || ia < 0 || ib < 0 || ic < 0 ||
                (uint32_t)ia >= n_vertices || (uint32_t)ib >= n_vertices || (uint32_t)ic >= n_vertices) {
                clean = false;
                break;
            }
            scene_object_t obj = record;      // as emit_triangle: the record's material, a triangle in the union
            obj.type = OT_TRIANGLE;
            obj.triangle.p1 = place.apply(vertices[ia]);
            obj.triangle.p2 = place.apply(vertices[ib]);
            obj.triangle.p3 = place.apply(vertices[ic]);
            std::memcpy(&slots[i], &obj, sizeof obj);
        }
    });
    if (!clean) return false;                              // (slots beyond objectStack.index are not part of the scene)
    scene->objectStack.index += (int)n_faces;
    scene->triangleLookup.triangleCount += (int)n_faces;
    if (timing)
        std::fprintf(stderr, "ply body on %u threads: line index %.1f ms, %u vertices %.1f ms, %u faces %.1f ms\n", n_threads,
                     ms(t0, t1), n_vertices, ms(t1, t2), n_faces, ms(t2, now()));
    return true;
}

// objectLoader.cpp:142-202 -- ASCII PLY: "element vertex N", "element face M",
// first three numbers of each vertex line, "3 i j k" face lines.
void import_ply(const scene_object_t &record, scene_t *scene) {
    TextCursor tc(resolve(record));
    if (!tc.ok()) tc.fatal("cannot open mesh");
    if (tc.bare_word() != "ply") tc.fatal("not a PLY file");

    uint32_t n_vertices = 0, n_faces = 0;
    while (!tc.at_end()) {
        std::string word = tc.bare_word();
        if (word == "element") {
            word = tc.bare_word();
            if (word == "vertex") n_vertices = TextCursor::to_u32(tc.number());
            else if (word == "face") n_faces = TextCursor::to_u32(tc.number());
        } else if (word == "end_header") {
            tc.skip_line();
            break;
        }
    }

    const Placement place(record.import.position, record.import.rotation, record.import.scale);
    if (import_ply_body_parallel(tc, record, place, scene, n_vertices, n_faces)) return;

    // (the reference mallocs totalVertices entries and fills what the file holds; here the array is sized by what the file
    //  CAN hold -- a record is at least one byte -- so that a header's count does not size memory, and a face that names a
    //  vertex the file never gave reads zeros, where the reference reads its uninitialised block)
    const size_t bytes_left = tc.size() > tc.position() ? tc.size() - tc.position() : 0;
    std::vector<v3_t> vertices(std::min<size_t>(n_vertices, bytes_left + 1));
    const auto vertex_at = [&](int i) { return (size_t)i < vertices.size() ? vertices[i] : v3_t{0.0, 0.0, 0.0}; };
    for (uint32_t i = 0; i < n_vertices && !tc.at_end(); i++) {
        double v[3];
        tc.vec3_bare(v);
        if (i < vertices.size()) vertices[i] = v3_t{v[0], v[1], v[2]};
        tc.skip_line();
    }

    for (uint32_t i = 0; i < n_faces && !tc.at_end(); i++) {
        const int corners = TextCursor::to_int(tc.number());
        if (corners != 3) tc.fatal("only triangular faces are supported");
        double f[3];
        tc.vec3_bare(f);
        tc.skip_line();
        const int ia = TextCursor::to_int(f[0]), ib = TextCursor::to_int(f[1]), ic = TextCursor::to_int(f[2]);
        if (ia < 0 || ib < 0 || ic < 0 || (uint32_t)ia >= n_vertices || (uint32_t)ib >= n_vertices ||
            (uint32_t)ic >= n_vertices)
            tc.fatal("face index out of range");
        emit_triangle(scene, record, place, vertex_at(ia), vertex_at(ib), vertex_at(ic), tc);
    }
}

// objectLoader.cpp:206-279 -- Wavefront OBJ subset: lines starting with a single
// 'v' or 'f' token carry three numbers; '#' and 'o' lines are skipped; any other
// byte is consumed one at a time; faces are 1-based; at most 1023 of each.
void import_obj(const scene_object_t &record, scene_t *scene) {
    TextCursor tc(resolve(record));
    if (!tc.ok()) tc.fatal("cannot open mesh");

    const size_t kMax = 1024;
    std::vector<v3_t> vertices, faces;
    while (!tc.at_end()) {
        const char c = tc.token();
        if (c == '#' || c == 'o') {
            tc.skip_line();
        } else if (c == 'v' || c == 'f') {
            double v[3];
            tc.vec3_bare(v);
            std::vector<v3_t> &dst = (c == 'v') ? vertices : faces;
            dst.push_back(v3_t{v[0], v[1], v[2]});
            if (dst.size() >= kMax) tc.fatal("OBJ import holds at most 1023 vertices / faces");
        }
    }
    if (faces.empty()) return;

    const Placement place(record.import.position, record.import.rotation, record.import.scale);
    // (the range is checked on the number as read: converting a value outside int's range is undefined, and a NaN fails too)
    const auto index_of = [&](double one_based) -> size_t {
        if (!(one_based >= 1.0 && one_based < (double)vertices.size() + 1.0)) tc.fatal("face index out of range");
        return (size_t)one_based - 1;
    };
    for (const v3_t &f : faces) {
        const size_t ia = index_of(f.x), ib = index_of(f.y), ic = index_of(f.z);
        emit_triangle(scene, record, place, vertices[ia], vertices[ib], vertices[ic], tc);
    }
}

}  // namespace

void ImportObject(scene_object_t importObject, scene_t *scene) {
    if (importObject.import.format == IT_BLENDER) import_obj(importObject, scene);
    else if (importObject.import.format == IT_PLY) import_ply(importObject, scene);
}
