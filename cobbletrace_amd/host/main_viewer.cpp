// main_viewer.cpp -- a window around this build's RayThread and Blit.  What the reference's application shell does
// (cobbletrace.cpp:26-125) is kept as BEHAVIOUR: a 640 x 640 bitmap in a resizable window titled "Cobble Trace", a client
// area of twice that size, every key-down handed to RayThread through the event queue, one RayThread call and one Blit
// per ~33 ms turn, usage and exit 0 without a scene file.  How it is written is this repository's own: the window and
// the bitmap are one object that owns both, the command line is read before anything is opened, and RayThread runs
// asynchronously (ct_raythread_options.async) -- the loop never waits for the GPU, a frame is shown by the first Blit after
// it has finished, on an MI355X the next turn: what the reference's progressive display comes to when a frame takes
// milliseconds instead of seconds.
// Built only where SDL2 is installed (host/Makefile: sdl2-config) or against tests/sdl_double; cobbletrace_headless is the
// same loop without a window.
#ifndef CT_HAVE_SDL
#error "main_viewer.cpp needs SDL2 (make -C cobbletrace_amd/host viewer)"
#endif
#include "cobbletrace_host.h"
#include "ctgpu.h"

#include <SDL.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace {

constexpr int kBitmapSide = 640;            // cobbletrace.cpp:14-15 (WIDTH, HEIGHT)
constexpr int kPendingEvents = 1000;        // cobbletrace.cpp:16
constexpr unsigned kTurnMs = 33;            // cobbletrace.cpp:121

struct CommandLine {
    char *scene_file = nullptr;              // (ParseSceneFile keeps the reference's signature: char *)
    uint32_t frame_flags = 0;
    bool ok = true;
};

CommandLine read_command_line(int argc, char **argv) {
    CommandLine c;
    for (int i = 1; i < argc; i++) {
        char *a = argv[i];
        if (!std::strcmp(a, "--spheres")) c.frame_flags |= CTGPU_F_SPHERES;
        else if (!std::strcmp(a, "--reflect-fix")) c.frame_flags |= CTGPU_F_REFLECT_FIX;
        else if (a[0] == '-' && a[1] == '-') { std::fprintf(stderr, "cobbletrace_viewer: no such option: %s\n", a); c.ok = false; }
        else if (!c.scene_file) c.scene_file = a;
    }
    return c;
}

// the window, its renderer, the bitmap RayThread draws into and the queue its keys arrive through
class Viewer {
public:
    Viewer() : pixels_((size_t)kBitmapSide * kBitmapSide, 0u), pending_((size_t)kPendingEvents) {}
    ~Viewer() {
        if (window_) SDL_DestroyWindow(window_);
        if (sdl_up_) SDL_Quit();
    }
    bool open() {
        sdl_up_ = SDL_Init(SDL_INIT_EVERYTHING) == 0;
        if (!sdl_up_) return complain("SDL_Init");
        window_ = SDL_CreateWindow("Cobble Trace", SDL_WINDOWPOS_UNDEFINED, SDL_WINDOWPOS_UNDEFINED, kBitmapSide, kBitmapSide, SDL_WINDOW_RESIZABLE);
        if (!window_) return complain("SDL_CreateWindow");
        SDL_Renderer *renderer = SDL_CreateRenderer(window_, -1, 0);
        if (!renderer) return complain("SDL_CreateRenderer");
        bitmap_ = bitmapSettings_t{pixels_.data(), kBitmapSide, kBitmapSide, nullptr};
        env_ = environment_t{};
        env_.bitmap = &bitmap_;
        env_.renderer = renderer;
        env_.clientWidth = 2 * kBitmapSide;               // cobbletrace.cpp:72-73: Blit stretches the bitmap over the client area
        env_.clientHeight = 2 * kBitmapSide;
        env_.events.queue = pending_.data();
        env_.events.capacity = (uint32_t)pending_.size();
        return true;
    }
    // one turn: what the window system has for us, a RayThread call, the bitmap to the screen.  False once the window was closed.
    bool turn(scene_t *scene) {
        bool open = true;
        for (SDL_Event e; SDL_PollEvent(&e);) {
            if (e.type == SDL_QUIT) open = false;
            if (e.type == SDL_KEYDOWN) AddEvent(&env_.events, event_t{ET_KEY_DOWN, EM_NONE, v2_t{0, 0}, (uint32_t)e.key.keysym.sym});
        }
        if (open) {
            RayThread(&env_, scene);
            Blit(&env_);
            SDL_Delay(kTurnMs);
        }
        return open;
    }

private:
    static bool complain(const char *what) {
        std::fprintf(stderr, "cobbletrace_viewer: %s failed: %s\n", what, SDL_GetError());
        return false;
    }
    std::vector<uint32_t> pixels_;
    std::vector<event_t> pending_;
    bitmapSettings_t bitmap_{};
    environment_t env_{};
    SDL_Window *window_ = nullptr;
    bool sdl_up_ = false;
};

}  // namespace

int main(int argc, char *argv[]) {
    const CommandLine cmd = read_command_line(argc, argv);
    if (!cmd.ok) return 2;
    if (!cmd.scene_file) {                    // (the reference prints its help and leaves with 0: cobbletrace.cpp:20-24)
        std::fprintf(stderr, "usage: cobbletrace_viewer scene.json [--spheres] [--reflect-fix]\n"
                             "  keys: w s a d i o r y p c m move and turn the camera, as in the reference (raythread.cpp:385-432)\n");
        return 0;
    }
    scene_t scene;
    std::memset(&scene, 0, sizeof scene);
    InitSceneData(&scene);
    ParseSceneFile(cmd.scene_file, &scene);
    if (scene.settings.wireframe) {           // cobbletrace.cpp:108-114 hands such scenes to the rasteriser: out of scope here
        std::fprintf(stderr, "cobbletrace_viewer: %s asks for wireframe drawing, which is the reference's rasteriser and not part of this build\n", cmd.scene_file);
        return 1;
    }
    ct_raythread_options options{CTGPU_RNG_HASH, 0, 0, 1, 1, cmd.frame_flags, 1};
    RayThreadSetOptions(&options);

    int rc = 0;
    {
        Viewer viewer;
        if (viewer.open()) while (viewer.turn(&scene)) {}
        else rc = 1;
        RayThreadReset();                     // (frames in flight finish before the bitmap they draw into goes away)
    }
    return rc;
}
