// One instantiation of render_impl<ST> per translation unit (the Makefile compiles this file eight
// times, -DFRAY_ST=0..5, 8, 9), so the kernel variants build in parallel.
#include "render_impl.hpp"

#ifndef FRAY_ST
#error "compile with -DFRAY_ST=0..5, 8 or 9"
#endif

namespace frayhip_detail {
template int render_impl<FRAY_ST>(frayhip_scene*, const frayhip_frame*, float*, int32_t*, double*, hipStream_t, frayhip_stats*);
}
