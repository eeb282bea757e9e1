// C ABI, host-only entry points (no HIP calls): scene parsing and small utilities.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>

#include "capi_common.h"
#include "host_scene.h"

namespace frayhip_detail {
thread_local std::string g_last_error;
void set_error(const std::string& s) { g_last_error = s; }
}  // namespace frayhip_detail

struct frayhip_host_scene {
    frayhost::HostScene* hs;
};

extern "C" {

int frayhip_scene_parse(const char* fray_path, frayhip_host_scene** out)
{
    if (!fray_path || !out) { frayhip_detail::set_error("frayhip_scene_parse: null argument"); return FRAYHIP_E_ARG; }
    std::string err;
    frayhost::HostScene* hs = frayhost::parse_scene_file(fray_path, err);
    if (!hs) { frayhip_detail::set_error(err); return FRAYHIP_E_PARSE; }
    *out = new frayhip_host_scene{hs};
    return FRAYHIP_OK;
}

frayhip_scene_desc* frayhip_host_scene_desc(frayhip_host_scene* hs) { return hs ? &hs->hs->desc : nullptr; }

void frayhip_host_scene_free(frayhip_host_scene* hs)
{
    if (!hs) return;
    delete hs->hs;
    delete hs;
}

int frayhip_bucket_count(int width, int height, int bucket_first, int bucket_stride)
{
    if (width <= 0 || height <= 0 || bucket_stride <= 0 || bucket_first < 0 || bucket_first >= bucket_stride) return FRAYHIP_E_ARG;
    int total = ((width - 1) / 48 + 1) * ((height - 1) / 48 + 1);
    return (total - bucket_first + bucket_stride - 1) / bucket_stride;
}

int frayhip_bucket_xy(int width, int height, int bucket, int* bx, int* by)
{
    if (width <= 0 || height <= 0 || !bx || !by) return FRAYHIP_E_ARG;
    const int BW = (width - 1) / 48 + 1, BH = (height - 1) / 48 + 1;
    if (bucket < 0 || bucket >= BW * BH) return FRAYHIP_E_ARG;
    *by = bucket / BW;
    *bx = (bucket % BW + FRAYHIP_BUCKET_SKEW * *by) % BW;
    return FRAYHIP_OK;
}

// convertTo8bit / Color::toRGB32 (color.h:29-34, 59-65): clamp, floor(x*255 + 0.5), 0x00RRGGBB.
int frayhip_to_rgb32(const float* rgb, uint32_t* out, int n_pixels)
{
    if (!rgb || !out || n_pixels < 0) return FRAYHIP_E_ARG;
    for (int i = 0; i < n_pixels; i++) {
        uint32_t c[3];
        for (int k = 0; k < 3; k++) {
            float x = rgb[i * 3 + k];
            if (x < 0) x = 0;
            if (x > 1) x = 1;
            c[k] = (uint32_t)(int)floor(x * 255.0f + 0.5f);
        }
        out[i] = (c[2]) | (c[1] << 8) | (c[0] << 16);
    }
    return FRAYHIP_OK;
}

int frayhip_save_bmp(const char* path, const float* rgb, int width, int height)
{
    if (!path || !rgb || width <= 0 || height <= 0) { frayhip_detail::set_error("frayhip_save_bmp: bad argument"); return FRAYHIP_E_ARG; }
    FILE* fp = fopen(path, "wb");
    if (!fp) { frayhip_detail::set_error(std::string("frayhip_save_bmp: cannot open ") + path); return FRAYHIP_E_ARG; }
    int rowsz = width * 3;
    if (rowsz % 4) rowsz += 4 - (rowsz % 4);
    unsigned char hdr[54] = {0};
    auto put32 = [&](int o, uint32_t v) { hdr[o] = v & 255; hdr[o + 1] = (v >> 8) & 255; hdr[o + 2] = (v >> 16) & 255; hdr[o + 3] = (v >> 24) & 255; };
    hdr[0] = 'B'; hdr[1] = 'M';
    put32(2, (uint32_t)(rowsz * height + 54));
    put32(10, 54);
    put32(14, 40);
    put32(18, (uint32_t)width);
    put32(22, (uint32_t)height);
    hdr[26] = 1; hdr[28] = 24;
    fwrite(hdr, 1, 54, fp);
    std::string row((size_t)rowsz, '\0');
    for (int y = height - 1; y >= 0; y--) {
        for (int x = 0; x < width; x++) {
            uint32_t t;
            frayhip_to_rgb32(rgb + ((size_t)y * width + x) * 3, &t, 1);
            row[(size_t)x * 3] = (char)(t & 0xff);
            row[(size_t)x * 3 + 1] = (char)((t >> 8) & 0xff);
            row[(size_t)x * 3 + 2] = (char)((t >> 16) & 0xff);
        }
        fwrite(row.data(), 1, (size_t)rowsz, fp);
    }
    fclose(fp);
    return FRAYHIP_OK;
}

const char* frayhip_last_error(void) { return frayhip_detail::g_last_error.c_str(); }
int frayhip_abi_version(void) { return FRAYHIP_ABI_VERSION; }

int frayhip_sizeof(const char* n)
{
    if (!n) return -1;
#define FRAYHIP_SZ(T) if (!strcmp(n, #T)) return (int)sizeof(T);
    FRAYHIP_SZ(frayhip_transform) FRAYHIP_SZ(frayhip_geom_ref) FRAYHIP_SZ(frayhip_node) FRAYHIP_SZ(frayhip_plane)
    FRAYHIP_SZ(frayhip_sphere) FRAYHIP_SZ(frayhip_cube) FRAYHIP_SZ(frayhip_csg) FRAYHIP_SZ(frayhip_triangle)
    FRAYHIP_SZ(frayhip_kdnode) FRAYHIP_SZ(frayhip_mesh) FRAYHIP_SZ(frayhip_texture) FRAYHIP_SZ(frayhip_shader)
    FRAYHIP_SZ(frayhip_layer) FRAYHIP_SZ(frayhip_light) FRAYHIP_SZ(frayhip_camera) FRAYHIP_SZ(frayhip_settings)
    FRAYHIP_SZ(frayhip_environment) FRAYHIP_SZ(frayhip_scene_desc) FRAYHIP_SZ(frayhip_frame) FRAYHIP_SZ(frayhip_stats)
#undef FRAYHIP_SZ
    return -1;
}

}  // extern "C"
