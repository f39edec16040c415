// Geometry either side of the stage-1 sampler (scripts/inference/infer.py): depth back-projection + the depth control image
// in front of the VAE encodes, and the recovery of 3-D point trajectories from the decoded displacement video behind the
// decoder prompt.  All of it is HBM-bound elementwise / reduction work on [3, F, H, W]-sized float tensors (49x480x832:
// 235 MB in, 235 MB out for the recovery): coalesced grid-stride kernels, one workgroup per reduction group.
#include "common.h"
#include "more4d_hip.h"

namespace {

// min / max of each contiguous group (one 1024-thread workgroup per group; groups here are single frames, <= 0.4 M floats)
__global__ __launch_bounds__(1024) void minmax_kernel(const float* x, int64_t group_len, float* out) {
    __shared__ float smin[16], smax[16];
    const float* g = x + (int64_t)blockIdx.x * group_len;
    float lo = INFINITY, hi = -INFINITY;
    for (int64_t i = threadIdx.x; i < group_len; i += 1024) { const float v = g[i]; lo = fminf(lo, v); hi = fmaxf(hi, v); }
    lo = -wave_max(-lo); hi = wave_max(hi);
    if ((threadIdx.x & 63) == 0) { smin[threadIdx.x >> 6] = lo; smax[threadIdx.x >> 6] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < 16; ++i) { lo = fminf(lo, smin[i]); hi = fmaxf(hi, smax[i]); }
        out[2 * blockIdx.x] = fminf(lo, smin[0]); out[2 * blockIdx.x + 1] = fmaxf(hi, smax[0]);
    }
}

// points = K^-1 (u, v, 1) * depth on the linspace(0,1) pixel grid (infer.py:179-195) + the cleaned z of :823-825
__global__ __launch_bounds__(256) void backproject_kernel(const float* depth, int H, int W, float inv_fx, float inv_fy, float* coords,
                                                          float* zclean) {
    const int64_t hw = (int64_t)H * W;
    const float du = W > 1 ? 1.f / (W - 1) : 0.f, dv = H > 1 ? 1.f / (H - 1) : 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < hw; i += (int64_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / W), c = (int)(i % W);
        const float d = depth[i];
        const float u = c == W - 1 ? 1.f : c * du, v = r == H - 1 ? 1.f : r * dv;     // torch.linspace end point is exact
        coords[i] = (u * inv_fx + (-0.5f * inv_fx)) * d;
        coords[hw + i] = (v * inv_fy + (-0.5f * inv_fy)) * d;
        coords[2 * hw + i] = d;
        float z = fminf(fmaxf(d, 0.f), 10000.f);       // clamp first (so +inf -> 1e4), then nan / tiny -> 1
        if (!(z == z) || z < 1e-5f) z = 1.f;
        zclean[i] = z;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void depth_control_kernel(const float* zclean, const float* minmax, T* out, int64_t hw) {
    const float lo = minmax[0], hi = minmax[1];
    const float den = hi - lo + 1e-8f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < hw; i += (int64_t)gridDim.x * blockDim.x) {
        const T v = (T)(2.f * (zclean[i] - lo) / den - 1.f);
        out[i] = v; out[hw + i] = v; out[2 * hw + i] = v;
    }
}

// out[b, c, f] = (rel + frame0 / diff_b) * diff_b   (mode 0 / 2, infer.py:198-219)   or   rel + frame0   (mode 1 / 3, --normalize_track_z
// :857-861); modes 0 / 1 replace frame 0 by frame0[b, c] itself (the stored point cloud of :870), modes 2 / 3 keep the recovered frame 0
// (what the reference's inverse_flow_norm_transform_no_diff returns)
template <typename T>
__global__ __launch_bounds__(256) void flow_recover_kernel(const T* rel, const float* frame0, const float* minmax, float* out, int B,
                                                           int F, int64_t hw, int mode) {
    const int64_t total = (int64_t)B * 3 * F * hw;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = i % hw;
        int64_t r = i / hw;
        const int f = (int)(r % F); r /= F;
        const int c = (int)(r % 3);
        const int b = (int)(r / 3);
        const float f0 = frame0[((int64_t)b * 3 + c) * hw + p];
        float v;
        if (f == 0 && !(mode & 2)) v = f0;
        else if (mode & 1) v = (float)rel[i] + f0;
        else {
            const float* mm = minmax + (int64_t)b * 6;
            float diff = fmaxf(fmaxf(mm[1] - mm[0], mm[3] - mm[2]), mm[5] - mm[4]);
            if (diff == 0.f) diff = 1.f;
            v = ((float)rel[i] + f0 / diff) * diff;
        }
        out[i] = v;
    }
}

inline unsigned grid_for(int64_t n) {
    int64_t g = (n + 255) / 256;
    if (g > 8192) g = 8192;
    return (unsigned)(g < 1 ? 1 : g);
}

}  // namespace

extern "C" int m4d_minmax(const float* x, int64_t n_groups, int64_t group_len, float* out, m4d_stream stream) {
    M4D_CHECK_ARG(x && out && n_groups > 0 && group_len > 0 && n_groups < (1ll << 31), "minmax: null/empty");
    hipLaunchKernelGGL(minmax_kernel, dim3((unsigned)n_groups), dim3(1024), 0, (hipStream_t)stream, x, group_len, out);
    M4D_CHECK_LAUNCH("minmax");
    return 0;
}

extern "C" int m4d_backproject(const float* depth, int H, int W, float inv_fx, float inv_fy, float* coords, float* zclean,
                               m4d_stream stream) {
    M4D_CHECK_ARG(depth && coords && zclean && H > 0 && W > 0, "backproject: null/empty");
    hipLaunchKernelGGL(backproject_kernel, dim3(grid_for((int64_t)H * W)), dim3(256), 0, (hipStream_t)stream, depth, H, W, inv_fx,
                       inv_fy, coords, zclean);
    M4D_CHECK_LAUNCH("backproject");
    return 0;
}

extern "C" int m4d_depth_control(m4d_dtype out_dt, const float* zclean, const float* minmax, void* out, int64_t hw, m4d_stream stream) {
    M4D_CHECK_ARG(zclean && minmax && out && hw > 0, "depth_control: null/empty");
    if (out_dt == M4D_BF16) hipLaunchKernelGGL(depth_control_kernel<bf16_t>, dim3(grid_for(hw)), dim3(256), 0, (hipStream_t)stream, zclean, minmax, (bf16_t*)out, hw);
    else if (out_dt == M4D_F32) hipLaunchKernelGGL(depth_control_kernel<float>, dim3(grid_for(hw)), dim3(256), 0, (hipStream_t)stream, zclean, minmax, (float*)out, hw);
    else { m4d_set_error("depth_control: bad dtype"); return -1; }
    M4D_CHECK_LAUNCH("depth_control");
    return 0;
}

extern "C" int m4d_flow_recover(m4d_dtype in_dt, const void* rel, const float* frame0, const float* minmax, float* out, int B, int F,
                                int64_t hw, int mode, m4d_stream stream) {
    M4D_CHECK_ARG(rel && frame0 && out && B > 0 && F > 0 && hw > 0, "flow_recover: null/empty");
    M4D_CHECK_ARG(mode >= 0 && mode <= 3 && ((mode & 1) || minmax), "flow_recover: modes 0 / 2 need the first frame's per-channel min/max");
    const dim3 grid(grid_for((int64_t)B * 3 * F * hw));
    if (in_dt == M4D_BF16) hipLaunchKernelGGL(flow_recover_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)rel, frame0, minmax, out, B, F, hw, mode);
    else if (in_dt == M4D_F32) hipLaunchKernelGGL(flow_recover_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)rel, frame0, minmax, out, B, F, hw, mode);
    else { m4d_set_error("flow_recover: bad dtype"); return -1; }
    M4D_CHECK_LAUNCH("flow_recover");
    return 0;
}
