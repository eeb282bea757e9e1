// Bucket-major <-> row-major copies for the multi-GPU exchange (SURVEY 8e, K9 "untile"): the packed layout
// is bucket-major, rows of 48 pixels inside a bucket (include/frayhip.h).
#pragma once
#include "dev_math.hpp"
#include "dev_scene.hpp"

// the packed (gather) layout: bucket-major, rows of 48 pixels inside a bucket (include/frayhip.h)
FD bool packed_pixel(const DFrame& F, int item, int& x, int& y)
{
    int k = item / 2304, local = item - k * 2304;
    int b = F.bucketFirst + k * F.bucketStride;
    int by = b / F.BW, bx = (b - by * F.BW + FRAYHIP_BUCKET_SKEW * by) % F.BW;      // frayhip_bucket_xy (include/frayhip.h)
    x = bx * 48 + local % 48;
    y = by * 48 + local / 48;
    return x < F.W && y < F.H;
}

// ---- multi-GPU bucket exchange ----------------------------------------------------------------------
static __global__ __launch_bounds__(256) void k_pack(DFrame F, int nItems, int channels, float* __restrict__ frame, float* __restrict__ packed, int unpack)
{
    for (int item = blockIdx.x * blockDim.x + threadIdx.x; item < nItems; item += gridDim.x * blockDim.x) {
        int x, y;
        bool ok = packed_pixel(F, item, x, y);
        for (int ch = 0; ch < channels; ch++) {
            size_t a = ((size_t)y * F.W + x) * channels + ch, b = (size_t)item * channels + ch;
            if (unpack) { if (ok) frame[a] = packed[b]; }
            else packed[b] = ok ? frame[a] : 0.0f;
        }
    }
}
