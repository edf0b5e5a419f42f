/* tests/sdl_double/SDL.h -- TEST DOUBLE, not SDL2.
 *
 * The image has no SDL2, so the windowed half of the host layer (host/main_viewer.cpp, the CT_HAVE_SDL branch of
 * host/blit.cpp) could never be compiled.  This header declares the part of SDL2's C API those two files use --
 * names, signatures, constants and event layout as SDL2 2.0's documentation gives them -- and sdl_double.cpp implements
 * it as a scripted stand-in: no window, key events read from a script, "presented" frames kept in memory and written
 * out at SDL_Quit.  `make -C cobbletrace_amd/host cobbletrace_viewer_double` builds the viewer against it and
 * tests/test_gpu_raythread.py runs it.  What this proves: the SDL branch compiles, links and behaves against an API of
 * this shape.  What it cannot prove: anything about the real library.
 */
#ifndef CT_TEST_SDL_DOUBLE_H
#define CT_TEST_SDL_DOUBLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef uint8_t Uint8;
typedef uint16_t Uint16;
typedef uint32_t Uint32;
typedef int32_t Sint32;
typedef Sint32 SDL_Keycode;

#define SDL_INIT_TIMER 0x00000001u
#define SDL_INIT_AUDIO 0x00000010u
#define SDL_INIT_VIDEO 0x00000020u
#define SDL_INIT_JOYSTICK 0x00000200u
#define SDL_INIT_HAPTIC 0x00001000u
#define SDL_INIT_GAMECONTROLLER 0x00002000u
#define SDL_INIT_EVENTS 0x00004000u
#define SDL_INIT_SENSOR 0x00008000u
#define SDL_INIT_EVERYTHING (SDL_INIT_TIMER | SDL_INIT_AUDIO | SDL_INIT_VIDEO | SDL_INIT_EVENTS | SDL_INIT_JOYSTICK | \
                             SDL_INIT_HAPTIC | SDL_INIT_GAMECONTROLLER | SDL_INIT_SENSOR)

#define SDL_WINDOWPOS_UNDEFINED 0x1FFF0000
#define SDL_WINDOW_RESIZABLE 0x00000020u

typedef struct SDL_Window SDL_Window;
typedef struct SDL_Renderer SDL_Renderer;
typedef struct SDL_Texture SDL_Texture;

typedef struct SDL_Rect { int x, y, w, h; } SDL_Rect;

typedef struct SDL_Surface {
    Uint32 flags;
    void *format;
    int w, h;
    int pitch;
    void *pixels;
    void *userdata;
} SDL_Surface;

typedef enum { SDL_FIRSTEVENT = 0, SDL_QUIT = 0x100, SDL_KEYDOWN = 0x300, SDL_KEYUP = 0x301 } SDL_EventType;

typedef struct SDL_Keysym {
    int scancode;
    SDL_Keycode sym;        /* letters are their ASCII codes (SDLK_w == 'w') */
    Uint16 mod;
    Uint32 unused;
} SDL_Keysym;

typedef struct SDL_KeyboardEvent {
    Uint32 type;
    Uint32 timestamp;
    Uint32 windowID;
    Uint8 state;
    Uint8 repeat;
    Uint8 padding2;
    Uint8 padding3;
    SDL_Keysym keysym;
} SDL_KeyboardEvent;

typedef union SDL_Event {
    Uint32 type;
    SDL_KeyboardEvent key;
    Uint8 padding[56];
} SDL_Event;

int SDL_Init(Uint32 flags);
void SDL_Quit(void);
void SDL_Log(const char *fmt, ...);
const char *SDL_GetError(void);
void SDL_Delay(Uint32 ms);
int SDL_PollEvent(SDL_Event *event);

SDL_Window *SDL_CreateWindow(const char *title, int x, int y, int w, int h, Uint32 flags);
void SDL_SetWindowTitle(SDL_Window *window, const char *title);
void SDL_DestroyWindow(SDL_Window *window);
SDL_Renderer *SDL_CreateRenderer(SDL_Window *window, int index, Uint32 flags);

SDL_Surface *SDL_CreateRGBSurfaceFrom(void *pixels, int width, int height, int depth, int pitch, Uint32 Rmask, Uint32 Gmask,
                                      Uint32 Bmask, Uint32 Amask);
void SDL_FreeSurface(SDL_Surface *surface);
SDL_Texture *SDL_CreateTextureFromSurface(SDL_Renderer *renderer, SDL_Surface *surface);
void SDL_DestroyTexture(SDL_Texture *texture);
int SDL_RenderCopy(SDL_Renderer *renderer, SDL_Texture *texture, const SDL_Rect *srcrect, const SDL_Rect *dstrect);
void SDL_RenderPresent(SDL_Renderer *renderer);

#ifdef __cplusplus
}
#endif
#endif
