// tests/sdl_double/sdl_double.cpp -- TEST DOUBLE, not SDL2 (see SDL.h here).
//
// A scripted stand-in for the calls of host/main_viewer.cpp and host/blit.cpp:
//   CT_SDL_DOUBLE_SCRIPT  space-separated: a letter = one key-down of that key, a number N = N loop iterations without
//                         an event, "quit" = SDL_QUIT (also delivered when the script runs out)
//   CT_SDL_DOUBLE_OUT     where SDL_Quit leaves the last PRESENTED frame (binary PPM from the texture's R, G, B masks)
// A frame counts as presented when SDL_RenderPresent follows an SDL_RenderCopy of a texture made from a surface; the
// texture copies the surface's pixels when it is created (SDL2 uploads them then), so freeing the surface first, or
// the caller's bitmap changing later, cannot alter what was shown.  Misuse aborts: a null renderer, a copy of a
// destroyed texture, a rectangle that is not the whole surface, masks other than the three the hand-off names.
#include "SDL.h"

#include <unistd.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <set>
#include <sstream>
#include <string>
#include <vector>

struct SDL_Window { std::string title; int w, h; };
struct SDL_Renderer { SDL_Window *window; std::vector<uint32_t> target; int w = 0, h = 0; bool copied = false; };
struct SDL_Texture { std::vector<uint32_t> pixels; int w, h; Uint32 masks[4]; };

namespace {
std::vector<std::string> g_script;
size_t g_next = 0;
long g_idle = 0;
bool g_quit_sent = false, g_initialised = false;
std::vector<uint32_t> g_presented;
int g_pw = 0, g_ph = 0;
Uint32 g_masks[4] = {0, 0, 0, 0};
unsigned long g_presents = 0, g_distinct = 0, g_keys = 0, g_polls = 0;
std::set<SDL_Texture *> g_live_textures;
std::set<SDL_Surface *> g_live_surfaces;

[[noreturn]] void misuse(const char *what) {
    std::fprintf(stderr, "sdl double: MISUSE: %s\n", what);
    std::abort();
}
int shift_of(Uint32 mask) { int s = 0; while (mask && !(mask & 1u)) { mask >>= 1; s++; } return s; }
}  // namespace

extern "C" {

int SDL_Init(Uint32 flags) {
    if (!(flags & SDL_INIT_VIDEO)) misuse("SDL_Init without video");
    g_initialised = true;
    const char *s = std::getenv("CT_SDL_DOUBLE_SCRIPT");
    std::istringstream in(s ? s : "quit");
    for (std::string tok; in >> tok;) g_script.push_back(tok);
    return 0;
}

void SDL_Quit(void) {
    if (!g_live_textures.empty() || !g_live_surfaces.empty()) misuse("textures or surfaces alive at SDL_Quit");
    std::fprintf(stderr, "sdl double: %lu polls, %lu key events, %lu presents, %lu distinct frames\n", g_polls, g_keys, g_presents, g_distinct);
    const char *out = std::getenv("CT_SDL_DOUBLE_OUT");
    if (out && !g_presented.empty()) {
        FILE *f = std::fopen(out, "wb");
        if (!f) { std::perror(out); return; }
        std::fprintf(f, "P6\n%d %d\n255\n", g_pw, g_ph);
        const int rs = shift_of(g_masks[0]), gs = shift_of(g_masks[1]), bs = shift_of(g_masks[2]);
        for (uint32_t p : g_presented) {
            const unsigned char rgb[3] = {(unsigned char)((p & g_masks[0]) >> rs), (unsigned char)((p & g_masks[1]) >> gs),
                                          (unsigned char)((p & g_masks[2]) >> bs)};
            std::fwrite(rgb, 1, 3, f);
        }
        std::fclose(f);
    }
}

void SDL_Log(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    std::vfprintf(stderr, fmt, ap);
    va_end(ap);
    std::fputc('\n', stderr);
}

const char *SDL_GetError(void) { return "sdl double: no error"; }

void SDL_Delay(Uint32 ms) { usleep(1000u * (ms < 4u ? ms : 4u)); }       // (a scripted session does not need real time)

int SDL_PollEvent(SDL_Event *event) {
    if (!g_initialised) misuse("SDL_PollEvent before SDL_Init");
    g_polls++;
    if (g_idle > 0) { g_idle--; return 0; }                     // an iteration without events
    if (g_next >= g_script.size()) {
        if (g_quit_sent) return 0;
        g_quit_sent = true;
        std::memset(event, 0, sizeof *event);
        event->type = SDL_QUIT;
        return 1;
    }
    const std::string &tok = g_script[g_next++];
    std::memset(event, 0, sizeof *event);
    if (tok == "quit") { g_quit_sent = true; event->type = SDL_QUIT; return 1; }
    if (tok.size() > 1 || (tok[0] >= '0' && tok[0] <= '9')) {   // N quiet iterations: this poll ends the first of them
        g_idle = std::atol(tok.c_str()) - 1;
        return 0;
    }
    event->key.type = SDL_KEYDOWN;
    event->key.state = 1;
    event->key.keysym.sym = (SDL_Keycode)tok[0];
    g_keys++;
    return 1;
}

SDL_Window *SDL_CreateWindow(const char *title, int, int, int w, int h, Uint32) {
    if (!g_initialised) misuse("SDL_CreateWindow before SDL_Init");
    return new SDL_Window{title ? title : "", w, h};
}
void SDL_SetWindowTitle(SDL_Window *window, const char *title) { if (!window) misuse("null window"); window->title = title; }
void SDL_DestroyWindow(SDL_Window *window) { delete window; }
SDL_Renderer *SDL_CreateRenderer(SDL_Window *window, int index, Uint32) {
    if (!window || index != -1) misuse("SDL_CreateRenderer: null window or an explicit driver index");
    SDL_Renderer *r = new SDL_Renderer();
    r->window = window;
    return r;
}

SDL_Surface *SDL_CreateRGBSurfaceFrom(void *pixels, int width, int height, int depth, int pitch, Uint32 Rmask, Uint32 Gmask,
                                      Uint32 Bmask, Uint32 Amask) {
    if (!pixels || width <= 0 || height <= 0 || depth != 32 || pitch < 4 * width) return nullptr;
    SDL_Surface *s = new SDL_Surface();
    s->w = width; s->h = height; s->pitch = pitch; s->pixels = pixels;
    Uint32 *masks = new Uint32[4]{Rmask, Gmask, Bmask, Amask};
    s->userdata = masks;
    g_live_surfaces.insert(s);
    return s;
}
void SDL_FreeSurface(SDL_Surface *surface) {
    if (!surface) return;
    if (!g_live_surfaces.erase(surface)) misuse("SDL_FreeSurface of something that is not a live surface");
    delete[] (Uint32 *)surface->userdata;
    delete surface;
}
SDL_Texture *SDL_CreateTextureFromSurface(SDL_Renderer *renderer, SDL_Surface *surface) {
    if (!renderer) misuse("SDL_CreateTextureFromSurface: null renderer");
    if (!surface || !g_live_surfaces.count(surface)) misuse("SDL_CreateTextureFromSurface: not a live surface");
    SDL_Texture *t = new SDL_Texture();
    t->w = surface->w; t->h = surface->h;
    std::memcpy(t->masks, surface->userdata, sizeof t->masks);
    t->pixels.resize((size_t)t->w * t->h);
    for (int y = 0; y < t->h; y++)
        std::memcpy(&t->pixels[(size_t)y * t->w], (const char *)surface->pixels + (size_t)y * surface->pitch, (size_t)t->w * 4);
    g_live_textures.insert(t);
    return t;
}
void SDL_DestroyTexture(SDL_Texture *texture) {
    if (!texture) return;
    if (!g_live_textures.erase(texture)) misuse("SDL_DestroyTexture of something that is not a live texture");
    delete texture;
}
int SDL_RenderCopy(SDL_Renderer *renderer, SDL_Texture *texture, const SDL_Rect *srcrect, const SDL_Rect *dstrect) {
    if (!renderer) misuse("SDL_RenderCopy: null renderer");
    if (!texture || !g_live_textures.count(texture)) misuse("SDL_RenderCopy: not a live texture");
    if (srcrect) misuse("SDL_RenderCopy: the hand-off copies the whole texture");
    if (dstrect && (dstrect->x != 0 || dstrect->y != 0 || dstrect->w != texture->w || dstrect->h != texture->h))
        misuse("SDL_RenderCopy: destination is not {0, 0, W, H}");
    if (texture->masks[0] != 0x000000ffu || texture->masks[1] != 0x0000ff00u || texture->masks[2] != 0x00ff0000u || texture->masks[3] != 0)
        misuse("SDL_RenderCopy: masks are not R=0xff G=0xff00 B=0xff0000 A=0");
    renderer->target = texture->pixels;
    renderer->w = texture->w; renderer->h = texture->h;
    std::memcpy(g_masks, texture->masks, sizeof g_masks);
    renderer->copied = true;
    return 0;
}
void SDL_RenderPresent(SDL_Renderer *renderer) {
    if (!renderer) misuse("SDL_RenderPresent: null renderer");
    if (!renderer->copied) misuse("SDL_RenderPresent with nothing copied since the last present");
    renderer->copied = false;
    g_presents++;
    if (renderer->target != g_presented) g_distinct++;
    g_presented = renderer->target;
    g_pw = renderer->w; g_ph = renderer->h;
}

}  // extern "C"
