// blit.cpp -- Blit(environment_t*), the consumer of the framebuffer RayThread fills (draw2d.h:22-64).
//
// With SDL2 (CT_HAVE_SDL, set by the Makefile when sdl2-config is found) and a renderer in the
// environment this is the reference's hand-off: the bitmap is wrapped, not copied, into an RGB surface
// (32 bits per pixel, pitch 4*width, masks R=0x000000ff G=0x0000ff00 B=0x00ff0000, no alpha -- the
// 0x00BBGGRR words the ray tracer stores), turned into a texture, copied to {0, 0, W, H} and presented;
// surface and texture live for one frame.
// Without SDL2, or without a renderer, Blit is a frame sink: it counts the frames it was shown and, if
// CT_BLIT_PPM names a file, leaves the latest one there as a binary PPM (the same bytes a display would get).
#include "cobbletrace_host.h"

#include <cstdio>
#include <cstdlib>

#ifdef CT_HAVE_SDL
#include <SDL.h>
#endif

namespace {
uint64_t g_frames_shown = 0;

void write_ppm(const char *path, const bitmapSettings_t *bitmap) {
    FILE *f = std::fopen(path, "wb");
    if (!f) { std::perror(path); return; }
    std::fprintf(f, "P6\n%d %d\n255\n", bitmap->width, bitmap->height);
    const uint32_t *pixels = (const uint32_t *)bitmap->memory;
    const size_t n = (size_t)bitmap->width * (size_t)bitmap->height;
    for (size_t i = 0; i < n; i++) {
        const uint32_t p = pixels[i];
        const unsigned char rgb[3] = {(unsigned char)(p & 0xff), (unsigned char)((p >> 8) & 0xff), (unsigned char)((p >> 16) & 0xff)};
        std::fwrite(rgb, 1, 3, f);
    }
    std::fclose(f);
}
}  // namespace

uint64_t BlitFrameCount(void) { return g_frames_shown; }

void Blit(environment_t *environment) {
    if (!environment || !environment->bitmap || !environment->bitmap->memory) return;
    g_frames_shown++;
#ifdef CT_HAVE_SDL
    if (environment->renderer) {
        const bitmapSettings_t *bitmap = environment->bitmap;
        SDL_Surface *surface = SDL_CreateRGBSurfaceFrom(bitmap->memory, bitmap->width, bitmap->height, 32, 4 * bitmap->width,
                                                        0x000000ffu, 0x0000ff00u, 0x00ff0000u, 0);
        if (!surface) {
            SDL_Log("Creating surface failed: %s", SDL_GetError());
            std::exit(1);                      // draw2d.h:45-48
        }
        SDL_Rect whole{0, 0, bitmap->width, bitmap->height};
        SDL_Texture *texture = SDL_CreateTextureFromSurface(environment->renderer, surface);
        SDL_RenderCopy(environment->renderer, texture, nullptr, &whole);
        SDL_RenderPresent(environment->renderer);
        SDL_FreeSurface(surface);
        SDL_DestroyTexture(texture);
        return;
    }
#endif
    if (const char *path = std::getenv("CT_BLIT_PPM")) write_ppm(path, environment->bitmap);
}
