// render_main.cpp -- host harness of the GPU path, the counterpart of the reference's
// src/main.cpp:46-91 (NPU variant): read ./input/rays.bin and ./input/spheres.bin, copy to
// the device, call the render_do boundary on a stream, synchronise, copy back, write
// ./output/color.bin.  Same fixed cwd-relative paths (main.cpp:32-33,40); unlike the
// reference, every I/O and runtime error is fatal (main ignores ReadFile's result and
// CHECK_ACL only prints: data_utils.h:41-47) and spheres are copied to the sphere buffer
// (main.cpp:72 copies them over the rays -- the "all black" bug of problem.md:41).
//
//   render_gpu [--width W] [--height H] [--samples S] [--depth D] [--spheres Ns]
//              [--mode k|o] [--retire] [--frame] [--seed X] [--gpus N] [--stripes K]
// With no options it behaves like the reference binary: 16x16, S=1, depth 5, 8 spheres.
// --frame runs the fused device path instead (rays generated on the device, samples
// accumulated on the device) and writes ./output/color.ppm directly.  With --gpus N (> 1, implies --frame) the
// frame is sharded over N devices in this ONE process (apt_multi_*: band d on device d, the reference's
// contiguous split of src/render.cpp:9-10,24-27; --stripes K interleaves K stripes per device) and assembled on
// device 0 by peer copies; per-band kernel times are printed.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>

#include <string>
#include <vector>

#include "../../include/render_mi355x.h"

#define CHECK_HIP(x)                                                                          \
    do {                                                                                      \
        hipError_t e_ = (x);                                                                  \
        if (e_ != hipSuccess) {                                                               \
            fprintf(stderr, "[ERROR]  %s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            exit(2);                                                                          \
        }                                                                                     \
    } while (0)

static bool read_file(const char *path, void *buf, size_t want) { // data_utils.h:55-96 ReadFile
    struct stat sb;
    if (stat(path, &sb) != 0 || !S_ISREG(sb.st_mode)) { fprintf(stderr, "[ERROR]  failed to get file %s\n", path); return false; }
    if ((size_t)sb.st_size == 0) { fprintf(stderr, "[ERROR]  file size is 0: %s\n", path); return false; }
    if ((size_t)sb.st_size > want) { fprintf(stderr, "[ERROR]  file size is larger than buffer size: %s\n", path); return false; }
    FILE *f = fopen(path, "rb");
    if (!f) { fprintf(stderr, "[ERROR]  Open file failed. path = %s\n", path); return false; }
    const size_t got = fread(buf, 1, (size_t)sb.st_size, f);
    fclose(f);
    if (got != (size_t)sb.st_size || got != want) {
        fprintf(stderr, "[ERROR]  %s: expected %zu bytes, got %zu\n", path, want, got);
        return false;
    }
    return true;
}

static bool write_file(const char *path, const void *buf, size_t size) { // data_utils.h:105-122 WriteFile
    FILE *f = fopen(path, "wb");
    if (!f) { fprintf(stderr, "[ERROR]  Open file failed. path = %s\n", path); return false; }
    const size_t put = fwrite(buf, 1, size, f);
    if (fclose(f) != 0 || put != size) { fprintf(stderr, "[ERROR]  Write file Failed.\n"); return false; }
    return true;
}

int main(int argc, char **argv) {
    apt_render_params prm;
    apt_default_params(&prm);
    bool frame = false;
    int gpus = 1, stripes = 1;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto val = [&]() -> const char * { if (i + 1 >= argc) { fprintf(stderr, "missing value for %s\n", a.c_str()); exit(1); } return argv[++i]; };
        if (a == "--width") prm.width = (uint32_t)atoi(val());
        else if (a == "--height") prm.height = (uint32_t)atoi(val());
        else if (a == "--samples") prm.samples = (uint32_t)atoi(val());
        else if (a == "--depth") prm.depth = (uint32_t)atoi(val());
        else if (a == "--spheres") { prm.num_spheres = (uint32_t)atoi(val()); prm.light_index = (int32_t)prm.num_spheres - 1; }
        else if (a == "--mode") prm.mode = (val()[0] == 'o') ? APT_MODE_ORACLE : APT_MODE_KERNEL;
        else if (a == "--seed") prm.seed = strtoull(val(), nullptr, 0);
        else if (a == "--retire") prm.flags |= APT_FLAG_RETIRE;
        else if (a == "--frame") frame = true;
        else if (a == "--gpus") { gpus = atoi(val()); frame = true; }
        else if (a == "--stripes") stripes = atoi(val());
        else { fprintf(stderr, "unknown option %s\n", a.c_str()); return 1; }
    }
    if (apt_device_count() < 1) { fprintf(stderr, "[ERROR]  no HIP device: the GPU path cannot run here\n"); return 2; }

    const uint32_t blockDim = 8;                                              // main.cpp:18
    const size_t n = (size_t)prm.width * prm.height * 4 * prm.samples;        // main.cpp:19
    const size_t rayBytes = n * sizeof(float) * 6;                            // main.cpp:47
    const size_t sphFloats = ((size_t)prm.num_spheres * 10 + 127) / 128 * 128;
    const size_t sphBytes = sphFloats * sizeof(float);                        // 512 for Ns = 8: main.cpp:48
    const size_t colBytes = n * sizeof(float) * 3;                            // main.cpp:49

    CHECK_HIP(hipSetDevice(0));                                               // main.cpp:52-53
    hipStream_t stream;
    CHECK_HIP(hipStreamCreate(&stream));                                      // main.cpp:54-55

    std::vector<float> sphHost(sphFloats);
    if (!read_file("./input/spheres.bin", sphHost.data(), sphBytes)) return 3; // main.cpp:71
    float *sphDev = nullptr;
    CHECK_HIP(hipMalloc(&sphDev, sphBytes));
    CHECK_HIP(hipMemcpyAsync(sphDev, sphHost.data(), sphBytes, hipMemcpyHostToDevice, stream));

    if (frame && gpus > 1) {
        if (gpus > apt_device_count() || stripes < 1) { fprintf(stderr, "[ERROR]  --gpus %d: only %d device(s) visible\n", gpus, apt_device_count()); return 2; }
        const size_t npix = (size_t)prm.width * prm.height;
        std::vector<int> ids(gpus);
        for (int d = 0; d < gpus; ++d) ids[d] = d;
        apt_multi *mg = nullptr;
        if (apt_multi_create(ids.data(), (uint32_t)gpus, (uint32_t)stripes, &prm, sphHost.data(), &mg) != APT_OK) { fprintf(stderr, "[ERROR]  %s\n", apt_last_error()); return 4; }
        float *fbDev = nullptr; uint8_t *u8Dev = nullptr;
        CHECK_HIP(hipMalloc(&fbDev, npix * 3 * sizeof(float)));
        CHECK_HIP(hipMalloc(&u8Dev, npix * 3));
        std::vector<float> band_ms(gpus);
        if (apt_multi_render(mg, fbDev, u8Dev, band_ms.data()) != APT_OK) { fprintf(stderr, "[ERROR]  %s\n", apt_last_error()); return 4; }
        for (int d = 0; d < gpus; ++d) printf("[INFO]  device %d band kernel %.3f ms\n", d, band_ms[d]);
        std::vector<uint8_t> u8(npix * 3);
        CHECK_HIP(hipMemcpy(u8.data(), u8Dev, u8.size(), hipMemcpyDeviceToHost));
        if (apt_write_ppm("./output/color.ppm", prm.width, prm.height, u8.data()) != APT_OK) return 5;
        apt_multi_destroy(mg);
        CHECK_HIP(hipFree(fbDev)); CHECK_HIP(hipFree(u8Dev));
    } else if (frame) {
        const size_t npix = (size_t)prm.width * prm.height;
        float *fbDev = nullptr; uint8_t *u8Dev = nullptr;
        CHECK_HIP(hipMalloc(&fbDev, npix * 3 * sizeof(float)));
        CHECK_HIP(hipMalloc(&u8Dev, npix * 3));
        if (render_frame(&prm, stream, sphDev, 0, npix, fbDev, u8Dev) != APT_OK) { fprintf(stderr, "[ERROR]  %s\n", apt_last_error()); return 4; }
        std::vector<uint8_t> u8(npix * 3);
        CHECK_HIP(hipMemcpyAsync(u8.data(), u8Dev, u8.size(), hipMemcpyDeviceToHost, stream));
        CHECK_HIP(hipStreamSynchronize(stream));
        if (apt_check(stream) != APT_OK) { fprintf(stderr, "[ERROR]  %s\n", apt_last_error()); return 4; }   // the kernel-side ASSERT (render.cpp:68-73)
        if (apt_write_ppm("./output/color.ppm", prm.width, prm.height, u8.data()) != APT_OK) return 5;
        CHECK_HIP(hipFree(fbDev)); CHECK_HIP(hipFree(u8Dev));
    } else {
        std::vector<float> rayHost(n * 6), colHost(n * 3);
        if (!read_file("./input/rays.bin", rayHost.data(), rayBytes)) return 3; // main.cpp:68
        float *rayDev = nullptr, *colDev = nullptr;
        CHECK_HIP(hipMalloc(&rayDev, rayBytes));                              // main.cpp:64-66
        CHECK_HIP(hipMalloc(&colDev, colBytes));
        CHECK_HIP(hipMemcpyAsync(rayDev, rayHost.data(), rayBytes, hipMemcpyHostToDevice, stream)); // main.cpp:69
        if (apt_set_default_params(&prm) != APT_OK) { fprintf(stderr, "[ERROR]  %s\n", apt_last_error()); return 4; }
        render_do(blockDim, nullptr, stream, (uint8_t *)rayDev, (uint8_t *)sphDev, (uint8_t *)colDev); // main.cpp:74
        if (apt_last_status() != APT_OK) { fprintf(stderr, "[ERROR]  %s\n", apt_last_error()); return 4; }
        CHECK_HIP(hipStreamSynchronize(stream));                              // main.cpp:75
        if (apt_check(stream) != APT_OK) { fprintf(stderr, "[ERROR]  %s\n", apt_last_error()); return 4; }   // the kernel-side ASSERT (render.cpp:68-73)
        CHECK_HIP(hipMemcpy(colHost.data(), colDev, colBytes, hipMemcpyDeviceToHost)); // main.cpp:77
        if (!write_file("./output/color.bin", colHost.data(), colBytes)) return 5;     // main.cpp:79
        CHECK_HIP(hipFree(rayDev)); CHECK_HIP(hipFree(colDev));
    }
    CHECK_HIP(hipFree(sphDev));
    CHECK_HIP(hipStreamDestroy(stream));                                      // main.cpp:89
    return 0;
}
