// Test harness (CPU only): the host loaders under AddressSanitizer + UndefinedBehaviorSanitizer.
// usage: san_loaders FILE...   -- loads each file by extension; a loader may REJECT a file, it may not crash,
// read out of bounds or overflow.  Built and driven by tests/test_host_scene.py.
#include <cstdio>
#include <cstring>
#include <string>

#include "host_scene.h"

int main(int argc, char** argv)
{
    for (int i = 1; i < argc; i++) {
        const std::string path = argv[i];
        const char* ext = strrchr(argv[i], '.');
        std::string err;
        bool ok = false;
        if (ext && !strcmp(ext, ".exr")) { frayhost::Image img; ok = frayhost::load_exr(argv[i], img, err); }
        else if (ext && !strcmp(ext, ".bmp")) { frayhost::Image img; ok = frayhost::load_bmp(argv[i], img, err); }
        else if (ext && !strcmp(ext, ".obj")) { frayhost::MeshData m; ok = frayhost::load_obj(argv[i], m); if (ok) frayhost::build_kd(m); }
        else if (ext && !strcmp(ext, ".fray")) { frayhost::HostScene* hs = frayhost::parse_scene_file(argv[i], err); ok = hs != nullptr; delete hs; }
        printf("%s %s %s\n", ok ? "loaded  " : "rejected", argv[i], err.c_str());
    }
    return 0;
}
