// main_viewer.cpp -- the reference's application shell (cobbletrace.cpp:26-125) over this build's
// RayThread and Blit: an SDL2 window, key-downs forwarded through the event queue, one RayThread call and
// one Blit per ~33 ms iteration.  RayThread runs asynchronously here (ct_raythread_options.async): the loop
// never waits for the GPU, a frame appears on the first Blit after it has finished -- on an MI355X that is
// the next iteration -- which is what the reference's progressive display degenerates to when a frame takes
// milliseconds instead of seconds.
// Built only where SDL2 is installed (host/Makefile: sdl2-config); cobbletrace_headless is the same loop
// without a window.
#ifndef CT_HAVE_SDL
#error "main_viewer.cpp needs SDL2 (make -C cobbletrace_amd/host viewer)"
#endif
#include "cobbletrace_host.h"
#include "ctgpu.h"

#include <SDL.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

static const int kWidth = 640, kHeight = 640;          // cobbletrace.cpp:14-15
static const int kEventQueueSize = 1000;               // cobbletrace.cpp:16

int main(int argc, char *argv[]) {
    if (argc < 2) {
        SDL_Log("cobbletrace scenefile\n\n");
        SDL_Log("To run cobbletrace provide a scene file.  See scene_file_1.json for an example.");
        return 0;
    }
    scene_t scene;
    std::memset(&scene, 0, sizeof scene);
    InitSceneData(&scene);
    ParseSceneFile(argv[1], &scene);
    if (scene.settings.wireframe) {
        SDL_Log("wireframe scenes use the rasteriser, which this build does not provide");
        return 1;
    }

    SDL_Init(SDL_INIT_EVERYTHING);
    SDL_Window *window = SDL_CreateWindow("CobbleTrace", SDL_WINDOWPOS_UNDEFINED, SDL_WINDOWPOS_UNDEFINED, kWidth, kHeight,
                                          SDL_WINDOW_RESIZABLE);
    if (!window) {
        SDL_Log("Could not create window\n");
        return 1;
    }
    SDL_SetWindowTitle(window, "Cobble Trace");
    SDL_Renderer *renderer = SDL_CreateRenderer(window, -1, 0);
    uint32_t *data = (uint32_t *)std::calloc((size_t)kWidth * kHeight, sizeof(uint32_t));
    if (!data) {
        SDL_Log("Failed to get memory for the buffer");
        return 1;
    }

    environment_t env{};
    bitmapSettings_t bitmap{data, kWidth, kHeight, nullptr};
    env.bitmap = &bitmap;
    env.clientHeight = kHeight * 2;
    env.clientWidth = kWidth * 2;
    env.renderer = renderer;
    env.events.capacity = kEventQueueSize;
    env.events.queue = (event_t *)std::calloc(env.events.capacity, sizeof(event_t));

    ct_raythread_options options{CTGPU_RNG_HASH, 0, 0, 1, 1, 0u, 1};
    for (int i = 2; i < argc; i++) {
        if (!std::strcmp(argv[i], "--spheres")) options.frame_flags |= CTGPU_F_SPHERES;
        else if (!std::strcmp(argv[i], "--reflect-fix")) options.frame_flags |= CTGPU_F_REFLECT_FIX;
    }
    RayThreadSetOptions(&options);

    bool running = true;
    while (running) {
        SDL_Event e;
        while (SDL_PollEvent(&e)) {
            if (e.type == SDL_QUIT) running = false;
            else if (e.type == SDL_KEYDOWN) AddEvent(&env.events, event_t{ET_KEY_DOWN, EM_NONE, v2_t{0, 0}, (uint32_t)e.key.keysym.sym});
        }
        RayThread(&env, &scene);
        Blit(&env);
        SDL_Delay(33);
    }
    RayThreadReset();
    SDL_DestroyWindow(window);
    SDL_Quit();
    return 0;
}
