// scene_parse.cpp -- scene-file reader behind the reference's input API.
//
// InitSceneData / ParseSceneFile keep the reference's signatures and observable
// results (scenefile.cpp:11-29, 256-299): same defaults, same object/light order
// on the stacks, same material packing, same triangle-lookup table.  The JSON
// subset is the one the reference accepts: a top-level object whose keys are
// "objects", "lights", "camera", "settings"; values are read positionally, keys
// are matched by exact string compare, numbers go through TextCursor::number().
#include "cobbletrace_host.h"
#include "host_internal.h"
#include "text_cursor.h"

#include <cstdlib>
#include <cstring>
#include <string>

namespace {

using ct::TextCursor;

v3_t read_v3(TextCursor &tc) {
    double a[3];
    tc.vec3_bracketed(a);
    return v3_t{a[0], a[1], a[2]};
}

// key of the next "key": pair (scenefile.cpp:45-48 and its three siblings)
std::string read_key(TextCursor &tc) {
    tc.expect('"');
    tc.unget();
    std::string k = tc.quoted();
    tc.expect(':');
    return k;
}

// after a value: true when the enclosing {...} / [...] closes, else eats the comma
bool closes(TextCursor &tc, char closer) {
    char c = tc.token();
    if (c == closer) return true;
    tc.unget();
    tc.expect(',');
    return false;
}

void push_object(scene_t *scene, const scene_object_t &obj, TextCursor &tc) {
    if (scene->objectStack.index >= scene->objectStack.size) tc.fatal("object stack full");
    std::memcpy(&scene->objectStack.objects[scene->objectStack.index++], &obj, sizeof obj);
}

// scenefile.cpp:31-131
void read_objects(TextCursor &tc, scene_t *scene) {
    tc.expect('[');
    int inline_triangles = 0;   // running count, re-added after every inline object (scenefile.cpp:119)
    std::string filename;       // storage the import record's string view points into

    while (!tc.at_end()) {
        tc.expect('{');
        scene_object_t obj;
        std::memset(&obj, 0, sizeof obj);   // type 0 = sphere, format 0 = OBJ when omitted

        while (!tc.at_end()) {
            const std::string key = read_key(tc);
            if (key == "type") {
                const std::string kind = tc.quoted();
                if (kind == "sphere") obj.type = OT_SPHERE;
                else if (kind == "triangle") { obj.type = OT_TRIANGLE; inline_triangles++; }
                else if (kind == "import") obj.type = OT_IMPORT;
                else tc.fatal("unknown object type " + kind);
            } else if (key == "center") {
                obj.sphere.center = read_v3(tc);
            } else if (key == "radius") {
                obj.sphere.radius = tc.number();
            } else if (key == "color") {
                const v3_t c = read_v3(tc);
                // double -> uint8_t at RgbToColor's parameters (scenefile.cpp:74, color.h:78): the low byte of the int
                // conversion, which is what the reference's binary makes of scene_file_1.json's [0, 100, 2000] (208) and
                // is defined for every value here
                const auto channel = [](double v) { return (uint8_t)(TextCursor::to_int(v) & 0xff); };
                obj.material.color = ct::pack_rgb(channel(c.x), channel(c.y), channel(c.z));
            } else if (key == "specular") {
                obj.material.specular = TextCursor::to_int(tc.number());
            } else if (key == "reflection") {
                obj.material.reflection = tc.number();
            } else if (key == "p1") {
                obj.triangle.p1 = read_v3(tc);
            } else if (key == "p2") {
                obj.triangle.p2 = read_v3(tc);
            } else if (key == "p3") {
                obj.triangle.p3 = read_v3(tc);
            } else if (key == "filename") {
                filename = tc.quoted();
                obj.import.filename.size = (int)filename.size();
                obj.import.filename.data = filename.empty() ? nullptr : &filename[0];
            } else if (key == "position") {
                obj.import.position = read_v3(tc);
            } else if (key == "rotation") {
                obj.import.rotation = read_v3(tc);
            } else if (key == "scale") {
                obj.import.scale = read_v3(tc);
            } else if (key == "format") {
                const std::string fmt = tc.quoted();
                if (fmt == "blender") obj.import.format = IT_BLENDER;
                else if (fmt == "ply") obj.import.format = IT_PLY;
            }
            if (closes(tc, '}')) break;
        }

        if (obj.type == OT_IMPORT) {
            ImportObject(obj, scene);
        } else {
            push_object(scene, obj, tc);
            scene->triangleLookup.triangleCount += inline_triangles;
        }
        if (closes(tc, ']')) break;
    }
}

// scenefile.cpp:133-191
void read_lights(TextCursor &tc, scene_t *scene) {
    tc.expect('[');
    while (!tc.at_end()) {
        tc.expect('{');
        light_t light;
        std::memset(&light, 0, sizeof light);
        while (!tc.at_end()) {
            const std::string key = read_key(tc);
            if (key == "type") {
                const std::string kind = tc.quoted();
                if (kind == "ambient") light.type = LT_AMBIENT;
                else if (kind == "point") light.type = LT_POINT;
                else if (kind == "directional") light.type = LT_DIRECTIONAL;
                else tc.fatal("unknown light type " + kind);
            } else if (key == "intensity") {
                light.intensity = tc.number();
            } else if (key == "position") {
                light.position = read_v3(tc);
            } else if (key == "direction") {
                light.direction = read_v3(tc);
            } else {
                tc.fatal("unknown light key " + key);
            }
            if (closes(tc, '}')) break;
        }
        if (scene->lightStack.index >= scene->lightStack.size) tc.fatal("light stack full");
        std::memcpy(&scene->lightStack.lights[scene->lightStack.index++], &light, sizeof light);
        if (closes(tc, ']')) break;
    }
}

// scenefile.cpp:193-210
void read_camera(TextCursor &tc, scene_t *scene) {
    tc.expect('{');
    const std::string key = read_key(tc);
    if (key != "position") tc.fatal("camera expects \"position\"");
    scene->camera.position = read_v3(tc);
    tc.expect('}');
}

// scenefile.cpp:212-244
void read_settings(TextCursor &tc, scene_t *scene) {
    tc.expect('{');
    while (!tc.at_end()) {
        const std::string key = read_key(tc);
        if (key == "numberOfThreads") scene->settings.numberOfThreads = TextCursor::to_int(tc.number());
        else if (key == "subsampling") scene->settings.subsampling = tc.boolean();
        else if (key == "wireframe") scene->settings.wireframe = tc.boolean();
        else if (key == "supersampling") scene->settings.supersampling = tc.boolean();
        else tc.fatal("unknown settings key " + key);
        if (closes(tc, '}')) break;
    }
}

}  // namespace

namespace ct {
std::string g_scene_dir;   // directory of the scene file being parsed (import fallback)
}

void InitSceneData(scene_t *scene) {
    scene->lightStack.size = MAX_LIGHTS;
    scene->lightStack.index = 0;
    scene->lightStack.type = ST_LIGHT;
    scene->lightStack.lights = (light_t *)std::calloc(MAX_LIGHTS, sizeof(light_t));

    scene->objectStack.size = MAX_OBJECTS;
    scene->objectStack.index = 0;
    scene->objectStack.type = ST_OBJECT;
    scene->objectStack.objects = (scene_object_t *)std::calloc(MAX_OBJECTS, sizeof(scene_object_t));

    scene->settings.numberOfThreads = 8;
    scene->settings.subsampling = true;
    scene->settings.wireframe = false;
    scene->settings.supersampling = false;

    scene->camera.position = v3_t{0, 0, -8};
}

void ParseSceneFile(char *filename, scene_t *scene) {
    TextCursor tc(filename);
    if (!tc.ok()) tc.fatal("cannot open scene file");

    const std::string path(filename);
    const size_t slash = path.find_last_of('/');
    ct::g_scene_dir = (slash == std::string::npos) ? std::string(".") : path.substr(0, slash);

    tc.expect('{');
    scene->triangleLookup.triangleCount = 0;

    while (!tc.at_end()) {
        tc.expect('"');
        tc.unget();
        const std::string section = tc.quoted();
        if (section == "objects") { tc.expect(':'); read_objects(tc, scene); }
        else if (section == "lights") { tc.expect(':'); read_lights(tc, scene); }
        else if (section == "camera") { tc.expect(':'); read_camera(tc, scene); }
        else if (section == "settings") { tc.expect(':'); read_settings(tc, scene); }
        if (closes(tc, '}')) break;
    }

    // triangle ordinal -> object-stack slot (scenefile.cpp:246-254)
    scene->triangleLookup.indexes =
        (uint32_t *)std::calloc(scene->triangleLookup.triangleCount ? scene->triangleLookup.triangleCount : 1,
                                sizeof(uint32_t));
    uint32_t n = 0;
    for (int i = 0; i < scene->objectStack.index; i++)
        if (scene->objectStack.objects[i].type == OT_TRIANGLE) scene->triangleLookup.indexes[n++] = (uint32_t)i;
}
