// fray_render -- a C++ host over the C ABI, shaped like the reference's main() (src/main.cpp:494-530)
// minus the SDL window: parse a .fray scene, render it on the GPU, write a BMP, print the timing line.
//
//   fray_render scene.fray [out.bmp] [width height] [spp]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "frayhip.h"

static int fail(const char* what)
{
    fprintf(stderr, "%s: %s\n", what, frayhip_last_error());
    return 1;
}

int main(int argc, char** argv)
{
    const char* sceneFile = argc > 1 ? argv[1] : "data/forest.fray";     // the reference's default (main.cpp:54)
    const char* out = argc > 2 ? argv[2] : "fray_0000.bmp";
    frayhip_host_scene* hs = nullptr;
    if (frayhip_scene_parse(sceneFile, &hs) != FRAYHIP_OK) return fail("Could not parse the scene");
    frayhip_scene_desc* d = frayhip_host_scene_desc(hs);
    if (argc > 4) { d->settings.frameWidth = atoi(argv[3]); d->settings.frameHeight = atoi(argv[4]); }
    if (argc > 5) {
        if (d->settings.gi) d->settings.numPaths = atoi(argv[5]);
        else if (d->camera.dof) d->camera.numDOFSamples = atoi(argv[5]);
    }
    d->settings.interactive = 0;
    if (frayhip_init(0) != FRAYHIP_OK) return fail("Cannot set up the GPU");
    frayhip_scene* scene = nullptr;
    if (frayhip_scene_create(d, &scene) != FRAYHIP_OK) return fail("Cannot upload the scene");
    const int W = d->settings.frameWidth, H = d->settings.frameHeight;
    std::vector<float> vfb((size_t)W * H * 3);
    frayhip_frame f = {FRAYHIP_MODE_RENDER, 42u, 0, 1, 0, 0};
    frayhip_stats st;
    auto t0 = std::chrono::steady_clock::now();
    if (frayhip_render(scene, &f, vfb.data(), nullptr, nullptr, &st) != FRAYHIP_OK) return fail("Render failed");
    double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("Render took %.2fs\n", sec);                                   // main.cpp:520
    if (frayhip_save_bmp(out, vfb.data(), W, H) != FRAYHIP_OK) return fail("Cannot write the image");
    frayhip_scene_destroy(scene);
    frayhip_host_scene_free(hs);
    printf("Exited cleanly\n");
    return 0;
}
