// render_do_cxx.cpp -- the reference declares its launch wrapper WITHOUT extern "C":
//     extern void render_do(uint32_t coreDim, void *l2ctrl, void *stream,
//                           uint8_t *rays, uint8_t *spheres, uint8_t *colors);        src/main.cpp:9-10
// and defines it as a plain C++ function (src/render.cpp:264-266), so an unmodified main.o references the
// Itanium-mangled symbol _Z9render_dojPvS_PhS0_S0_.  This translation unit exports exactly that symbol next to
// the C one of the same name (render_kernels.hip); it must not see include/render_mi355x.h, whose extern "C"
// declaration of render_do would clash with the C++-linkage definition below.
#include <stdint.h>

extern "C" void apt_render_do(uint32_t blockDim, void *l2ctrl, void *stream, uint8_t *rays, uint8_t *spheres,
                              uint8_t *colors);

__attribute__((visibility("default")))
void render_do(uint32_t blockDim, void *l2ctrl, void *stream, uint8_t *rays, uint8_t *spheres, uint8_t *colors) {
    apt_render_do(blockDim, l2ctrl, stream, rays, spheres, colors);
}
