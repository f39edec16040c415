// gemm_bt256w_kernel (gemm_wide.h) with the M4D_EPI_STORE epilogue; gemm.hip dispatches here (GemmArgs lives in an anonymous
// namespace: handed over as an opaque pointer)
#include "gemm_wide.h"

extern "C" __attribute__((visibility("hidden"))) int m4d_launch_gemm_wide_store(const void* args, unsigned nwg, hipStream_t st) {
    return launch_gemm_wide<M4D_EPI_STORE>(*(const GemmArgs*)args, nwg, st);
}

extern "C" __attribute__((visibility("hidden"))) int m4d_launch_gemm_wide_store_persistent(const void* args, unsigned nwg, unsigned ncu, hipStream_t st) {
    return launch_gemm_wide_persistent<M4D_EPI_STORE>(*(const GemmArgs*)args, nwg, ncu, st);
}
