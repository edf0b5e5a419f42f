// main_headless.cpp -- the reference's application shell (cobbletrace.cpp:26-125) without
// SDL: parse the scene, allocate the bitmap, call RayThread once per "frame", write the
// result as a binary PPM instead of blitting it.  The bitmap size, fixed at 640x640 in the
// reference (cobbletrace.cpp:14-15), is a command-line option here.
#include "cobbletrace_host.h"
#include "ctgpu.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

static void usage() {
    std::fprintf(stderr,
                 "cobbletrace_headless scenefile [--width W] [--height H] [--out image.ppm]\n"
                 "                     [--rng hash|libc] [--seed N] [--device N] [--gpus N] [--keys wasd...] [--async] [--spheres] [--reflect-fix]\n\n"
                 "Renders the scene file's view (and one more frame per key of --keys) on the GPU and writes the last frame.\n"
                 "Scene files: assets/*.json (the reference's format).  Like the reference's Help(), this leaves with status 0.\n");
    std::exit(0);
}

int main(int argc, char **argv) {
    if (argc < 2) usage();
    int width = 640, height = 640;
    std::string out = "cobbletrace.ppm", keys;
    ct_raythread_options opt{CTGPU_RNG_HASH, 0, 0, 1, 0, 0u, 1};      // (the bitmap below lives as long as the process)
    for (int i = 2; i < argc; i++) {
        const std::string a = argv[i];
        auto need = [&](const char *name) -> const char * {
            if (i + 1 >= argc) { std::fprintf(stderr, "%s needs a value\n", name); std::exit(1); }
            return argv[++i];
        };
        if (a == "--width") width = std::atoi(need("--width"));
        else if (a == "--height") height = std::atoi(need("--height"));
        else if (a == "--out") out = need("--out");
        else if (a == "--rng") opt.rng_mode = std::strcmp(need("--rng"), "libc") == 0 ? CTGPU_RNG_LIBC : CTGPU_RNG_HASH;
        else if (a == "--seed") opt.rng_seed = std::strtoull(need("--seed"), nullptr, 10);
        else if (a == "--device") opt.device = std::atoi(need("--device"));
        else if (a == "--gpus") opt.n_devices = std::atoi(need("--gpus"));
        else if (a == "--keys") keys = need("--keys");
        else if (a == "--async") opt.async = 1;
        else if (a == "--spheres") opt.frame_flags |= CTGPU_F_SPHERES;
        else if (a == "--reflect-fix") opt.frame_flags |= CTGPU_F_REFLECT_FIX;
        else usage();
    }

    scene_t scene;
    std::memset(&scene, 0, sizeof scene);
    InitSceneData(&scene);
    ParseSceneFile(argv[1], &scene);
    if (scene.settings.wireframe) {
        std::fprintf(stderr, "wireframe scenes use the rasteriser, which this build does not provide\n");
        return 1;
    }

    std::vector<uint32_t> pixels((size_t)width * height, 0u);   // calloc, cobbletrace.cpp:57
    bitmapSettings_t bitmap{pixels.data(), width, height, nullptr};
    std::vector<event_t> ring(1000);                            // EVENT_QUEUE_SIZE
    environment_t env{};
    env.bitmap = &bitmap;
    env.clientWidth = width * 2;
    env.clientHeight = height * 2;
    env.events.queue = ring.data();
    env.events.capacity = (uint32_t)ring.size();

    RayThreadSetOptions(&opt);
    // the main loop of cobbletrace.cpp:90-120: key-downs in, RayThread, Blit.  With --async RayThread only queues
    // a frame and the loop keeps turning until it has arrived (no 33 ms sleep here: nobody is watching)
    size_t next_key = 0;
    unsigned long iterations = 0;
    for (;;) {
        if (iterations > 0 && !RayThreadBusy() && next_key < keys.size())     // one frame per key, like one SDL_KEYDOWN each
            AddEvent(&env.events, event_t{ET_KEY_DOWN, EM_NONE, v2_t{0, 0}, (uint32_t)keys[next_key++]});
        RayThread(&env, &scene);
        Blit(&env);
        iterations++;
        if (!RayThreadBusy() && next_key >= keys.size()) break;
    }
    std::fprintf(stderr, "%lu loop iterations, %llu frames\n", iterations, (unsigned long long)RayThreadFramesFinished());

    FILE *f = std::fopen(out.c_str(), "wb");
    if (!f) { std::perror(out.c_str()); return 1; }
    std::fprintf(f, "P6\n%d %d\n255\n", width, height);
    for (uint32_t p : pixels) {
        const unsigned char rgb[3] = {(unsigned char)(p & 0xff), (unsigned char)((p >> 8) & 0xff),
                                      (unsigned char)((p >> 16) & 0xff)};
        std::fwrite(rgb, 1, 3, f);
    }
    std::fclose(f);
    std::fprintf(stderr, "wrote %s (%dx%d)\n", out.c_str(), width, height);
    return 0;
}
