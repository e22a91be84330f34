// Highway layers of the reference's Embedding (layers/encoding.py:9-59), SURVEY 8(f) row N2: per layer
//   g = sigmoid(W_g x + b_g),  t = relu(W_t x + b_t),  y = g*t + (1-g)*x
// The two linear maps run as ONE library GEMM against the stacked [W_g ; W_t] (rows x 2H pre-activations); the kernels
// here fuse everything element-wise around it (the stock module spends 8 launches per layer on it, ~25 with autograd):
//   forward   GT (rows, 2H) pre-activations  ->  [g | t] in place,  y
//   backward  d_y, x, [g | t]  ->  D = [d pre_g | d pre_t] in place over [g | t],  d_x_direct = d_y * (1 - g)
// (the caller then adds D . [W_g ; W_t] to d_x_direct and forms the weight gradients D^T . x with library GEMMs).
#include "common.h"

namespace mmb {

__global__ __launch_bounds__(256) void highway_gate_fwd_kernel(const float* __restrict__ x, float* __restrict__ gt,
                                                               float* __restrict__ y, long rows, int H) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;   // over rows * H/4
    if (i >= rows * (H / 4)) return;
    const long row = i / (H / 4);
    const int d = (i % (H / 4)) * 4;
    float* pg = gt + row * 2 * H + d;
    const f4 a = *reinterpret_cast<const f4*>(pg), b = *reinterpret_cast<const f4*>(pg + H);
    const f4 xv = *reinterpret_cast<const f4*>(x + row * H + d);
    f4 g, t, o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        g[e] = 1.0f / (1.0f + expf(-a[e]));
        t[e] = fmaxf(b[e], 0.0f);
        o[e] = g[e] * t[e] + (1.0f - g[e]) * xv[e];
    }
    *reinterpret_cast<f4*>(pg) = g;
    *reinterpret_cast<f4*>(pg + H) = t;
    *reinterpret_cast<f4*>(y + row * H + d) = o;
}

__global__ __launch_bounds__(256) void highway_gate_bwd_kernel(const float* __restrict__ d_y, const float* __restrict__ x,
                                                               float* __restrict__ gt, float* __restrict__ d_x, long rows, int H) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * (H / 4)) return;
    const long row = i / (H / 4);
    const int d = (i % (H / 4)) * 4;
    float* pg = gt + row * 2 * H + d;
    const f4 g = *reinterpret_cast<const f4*>(pg), t = *reinterpret_cast<const f4*>(pg + H);
    const f4 xv = *reinterpret_cast<const f4*>(x + row * H + d), dy = *reinterpret_cast<const f4*>(d_y + row * H + d);
    f4 dg, dt, dx;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        dg[e] = dy[e] * (t[e] - xv[e]) * g[e] * (1.0f - g[e]);
        dt[e] = t[e] > 0.0f ? dy[e] * g[e] : 0.0f;     // relu'(0) = 0, as torch
        dx[e] = dy[e] * (1.0f - g[e]);
    }
    *reinterpret_cast<f4*>(pg) = dg;
    *reinterpret_cast<f4*>(pg + H) = dt;
    *reinterpret_cast<f4*>(d_x + row * H + d) = dx;
}

}  // namespace mmb

using namespace mmb;

extern "C" int mmb_highway_gate_fwd(const float* x, float* gt, float* y, long rows, int H, int device, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    MMB_REQUIRE(x && gt && y && rows >= 1 && H >= 4 && H % 4 == 0, "mmb_highway_gate_fwd: bad argument (H must be a multiple of 4)");
    MMB_HIP(hipSetDevice(device));
    const long n = rows * (H / 4);
    hipLaunchKernelGGL(highway_gate_fwd_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, x, gt, y, rows, H);
    MMB_HIP(hipGetLastError());
    return MMB_OK;
}

extern "C" int mmb_highway_gate_bwd(const float* d_y, const float* x, float* gt, float* d_x, long rows, int H, int device, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    MMB_REQUIRE(d_y && x && gt && d_x && rows >= 1 && H >= 4 && H % 4 == 0, "mmb_highway_gate_bwd: bad argument (H must be a multiple of 4)");
    MMB_HIP(hipSetDevice(device));
    const long n = rows * (H / 4);
    hipLaunchKernelGGL(highway_gate_bwd_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, d_y, x, gt, d_x, rows, H);
    MMB_HIP(hipGetLastError());
    return MMB_OK;
}
