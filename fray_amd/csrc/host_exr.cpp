// OpenEXR reader for the cubemap faces (reference: Bitmap::loadEXR, src/bitmap.cpp:238-264, which
// goes through the OpenEXR library).  Placeholder until the PIZ decoder lands: reports failure,
// which leaves the environment "declared but not loaded" (misses shade black).
#include "host_scene.h"

namespace frayhost {

bool load_exr(const char* path, Image& img, std::string& err)
{
    (void)path; (void)img;
    err = "EXR decoding not available";
    return false;
}

}  // namespace frayhost
