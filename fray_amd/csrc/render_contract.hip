// The path tracer's two ray kernels once more, compiled with -ffp-contract=fast (Makefile: variantC<ST>.o, -DFRAY_ARITH=1): multiply-add pairs of the
// FP64 geometry fuse into v_fma_f64.  The reference's build has no FMA (x86-64 baseline), so these kernels do NOT reproduce its bits; they are used,
// when a scene's option "fp_contract" is 1, only for what north_star bounds by colour (1e-4 RMS per channel): every bounce AFTER a camera sample's
// first closest hit, and every next-event visibility query.  Primary hit records (MODE_PRIMARY_ID) and a sample's first bounce never run here.
#ifndef FRAY_ARITH
#error "compile with -DFRAY_ARITH=1 -ffp-contract=fast"
#endif
#include "render_state.hpp"
#include "kernels.hpp"

#ifndef FRAY_ST
#error "compile with -DFRAY_ST=0..5, 8 or 9"
#endif

namespace frayhip_detail {
template <int ST> void launch_bounce_contracted(int grid, hipStream_t stream, const BounceArgs& A)
{
    hipLaunchKernelGGL((k_pt_bounce<ST, false, false, FRAY_ARITH>), dim3(grid), dim3(256), 0, stream, A);
}
template <int ST> void launch_shadow_contracted(int grid, hipStream_t stream, const ShadowArgs& A)
{
    hipLaunchKernelGGL((k_pt_shadow<ST, FRAY_ARITH>), dim3(grid), dim3(256), 0, stream, A);
}
template void launch_bounce_contracted<FRAY_ST>(int, hipStream_t, const BounceArgs&);
template void launch_shadow_contracted<FRAY_ST>(int, hipStream_t, const ShadowArgs&);
}
