// gemm_bt256w_kernel (gemm_wide.h) with the M4D_EPI_STORE_F32 epilogue; gemm.hip dispatches here (GemmArgs lives in an anonymous
// namespace: handed over as an opaque pointer)
#include "gemm_wide.h"

extern "C" __attribute__((visibility("hidden"))) int m4d_launch_gemm_wide_f32(const void* args, unsigned nwg, hipStream_t st) {
    return launch_gemm_wide<M4D_EPI_STORE_F32>(*(const GemmArgs*)args, nwg, st);
}
// batched form (gridDim.y = K-slices; m4d_gemm_bt_taps)
extern "C" __attribute__((visibility("hidden"))) int m4d_launch_gemm_wide_f32_batched(const void* args, unsigned nwg, unsigned nby, hipStream_t st) {
    return launch_gemm_wide<M4D_EPI_STORE_F32>(*(const GemmArgs*)args, nwg, st, nby);
}
