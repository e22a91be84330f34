// Grouped element-wise products with dropout masks: dst_k = a_k * m_k  or  dst_k += a_k * m_k  for up to MMB_MASK_MAX tensors in ONE
// launch.  The reference applies dropout as F.dropout(x) (layers/encoding.py:81,104; layers/attention.py:66-67): the host draws the
// masks with torch's generator (one flat draw per step, region_fn.py) and the products used to be torch._foreach_mul / addcmul
// launches -- nine multi-tensor launches of 23-29 us each per training-mode step (290 us of a 2.6-ms step,
// profiles/r04_drop02_timeline_graph.md), at 2-3 TB/s.  Here: 16-B accesses, 8 of them per thread requested before the first use.
#include "common.h"

namespace mmb {

constexpr int MK_PER_BLOCK = 256 * 4 * 8;      // elements one workgroup covers
// a mask operand is either the mask itself (keep < 0) or the UNIFORM draw u in [0, 1) it is decided by: m = u < keep ? scale : 0
// (keep = 1 - p, scale = 1 / (1 - p): F.dropout's Bernoulli(1 - p) from torch.rand of the same generator -- one 14-us kernel per step
// where F.dropout over a vector of ones took 40)
__device__ __forceinline__ f4 mask_of(f4 u, float keep, float scale) {
    if (keep < 0.f) return u;
    return f4{u.x < keep ? scale : 0.f, u.y < keep ? scale : 0.f, u.z < keep ? scale : 0.f, u.w < keep ? scale : 0.f};
}
__device__ __forceinline__ float mask_of(float u, float keep, float scale) { return keep < 0.f ? u : (u < keep ? scale : 0.f); }
struct MaskMulArgs {
    const float* a[MMB_MASK_MAX];
    const float* m[MMB_MASK_MAX];
    float* dst[MMB_MASK_MAX];
    long n[MMB_MASK_MAX];
    int blk_begin[MMB_MASK_MAX + 1];
    int k;
    float keep, scale;
};

template <bool ACC>
__global__ __launch_bounds__(256) void masked_mul_kernel(const MaskMulArgs p) {
    int k = 0;
    for (int i = 1; i < p.k; ++i)
        if ((int)blockIdx.x >= p.blk_begin[i]) k = i;
    const float* a = p.a[k];          // (dst may alias a: no __restrict__)
    const float* m = p.m[k];
    float* dst = p.dst[k];
    const long n = p.n[k];
    const long base = (long)(blockIdx.x - p.blk_begin[k]) * MK_PER_BLOCK;
    if (base + MK_PER_BLOCK <= n) {
        f4 av[8], mv[8], dv[8];
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const long i = base + ((long)it * 256 + threadIdx.x) * 4;
            av[it] = *reinterpret_cast<const f4*>(a + i);
            mv[it] = mask_of(*reinterpret_cast<const f4*>(m + i), p.keep, p.scale);
            if (ACC) dv[it] = *reinterpret_cast<const f4*>(dst + i);
        }
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const long i = base + ((long)it * 256 + threadIdx.x) * 4;
            *reinterpret_cast<f4*>(dst + i) = ACC ? dv[it] + av[it] * mv[it] : av[it] * mv[it];
        }
    } else {
        for (int it = 0; it < 8; ++it) {
            const long i = base + ((long)it * 256 + threadIdx.x) * 4;
            if (i + 3 < n) {
                const f4 v = *reinterpret_cast<const f4*>(a + i) * mask_of(*reinterpret_cast<const f4*>(m + i), p.keep, p.scale);
                *reinterpret_cast<f4*>(dst + i) = ACC ? *reinterpret_cast<const f4*>(dst + i) + v : v;
            } else {
                for (long j = i; j < n; ++j) dst[j] = ACC ? dst[j] + a[j] * mask_of(m[j], p.keep, p.scale) : a[j] * mask_of(m[j], p.keep, p.scale);
            }
        }
    }
}

// dst = (sum_i x_i * m_i) * mo with up to MMB_MASK_TERMS terms; m_i / mo null = 1.  The backward pass's cotangent of the input
// encoders' outputs: d_text = (d_text_aa + d_text_ai + d_text_d_aa * m_aa + d_text_d_ai * m_ai) * m_out was an add_, two addcmul
// and a mul launch (four passes over d_text); one pass here.  dst may alias any x_i.
struct MaskSumArgs {
    float* dst[MMB_MASK_MAX];
    const float* x[MMB_MASK_MAX][MMB_MASK_TERMS];
    const float* m[MMB_MASK_MAX][MMB_MASK_TERMS];
    const float* mo[MMB_MASK_MAX];
    long n[MMB_MASK_MAX];
    int nterms[MMB_MASK_MAX];
    int blk_begin[MMB_MASK_MAX + 1];
    int k;
    float keep, scale;
};
constexpr int MS_PER_BLOCK = 256 * 4 * 4;
__global__ __launch_bounds__(256) void masked_sum_kernel(const MaskSumArgs p) {
    int k = 0;
    for (int i = 1; i < p.k; ++i)
        if ((int)blockIdx.x >= p.blk_begin[i]) k = i;
    float* dst = p.dst[k];             // (may alias a term: no __restrict__)
    const long n = p.n[k];
    const int nt = p.nterms[k];
    const float* mo = p.mo[k];
    const long base = (long)(blockIdx.x - p.blk_begin[k]) * MS_PER_BLOCK;
    const bool whole = base + MS_PER_BLOCK <= n;
    f4 acc[4];
    f4 ov[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        acc[it] = f4{0.f, 0.f, 0.f, 0.f};
        ov[it] = f4{1.f, 1.f, 1.f, 1.f};
    }
    if (whole) {
#pragma unroll
        for (int t = 0; t < MMB_MASK_TERMS; ++t) {
            if (t < nt) {           // (uniform)
                const float* x = p.x[k][t];
                const float* m = p.m[k][t];
                f4 xv[4], mv[4];
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const long i = base + ((long)it * 256 + threadIdx.x) * 4;
                    xv[it] = *reinterpret_cast<const f4*>(x + i);
                    mv[it] = m ? mask_of(*reinterpret_cast<const f4*>(m + i), p.keep, p.scale) : f4{1.f, 1.f, 1.f, 1.f};
                }
#pragma unroll
                for (int it = 0; it < 4; ++it) acc[it] += xv[it] * mv[it];
            }
        }
        if (mo) {
#pragma unroll
            for (int it = 0; it < 4; ++it) ov[it] = mask_of(*reinterpret_cast<const f4*>(mo + base + ((long)it * 256 + threadIdx.x) * 4), p.keep, p.scale);
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) *reinterpret_cast<f4*>(dst + base + ((long)it * 256 + threadIdx.x) * 4) = acc[it] * ov[it];
    } else {
        for (int it = 0; it < 4; ++it) {
            const long i0 = base + ((long)it * 256 + threadIdx.x) * 4;
            for (long j = i0; j < i0 + 4 && j < n; ++j) {
                float a = 0.f;
                for (int t = 0; t < nt; ++t) a += p.x[k][t][j] * (p.m[k][t] ? mask_of(p.m[k][t][j], p.keep, p.scale) : 1.f);
                dst[j] = a * (mo ? mask_of(mo[j], p.keep, p.scale) : 1.f);
            }
        }
    }
}

}  // namespace mmb

using namespace mmb;

extern "C" int mmb_masked_sum(const mmb_masked_sum_desc* d, int k, float keep, float scale, int device, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    MMB_REQUIRE(d && k >= 1 && k <= MMB_MASK_MAX, "mmb_masked_sum: 1..%d tensors", MMB_MASK_MAX);
    MaskSumArgs p{};
    p.k = k; p.keep = keep; p.scale = scale;
    int blk = 0;
    for (int i = 0; i < k; ++i) {
        MMB_REQUIRE(d[i].dst && d[i].n >= 0 && d[i].nterms >= 1 && d[i].nterms <= MMB_MASK_TERMS, "mmb_masked_sum: bad descriptor %d", i);
        uintptr_t al = (uintptr_t)d[i].dst | (uintptr_t)d[i].mo;
        for (int t = 0; t < d[i].nterms; ++t) {
            MMB_REQUIRE(d[i].x[t], "mmb_masked_sum: null term %d of tensor %d", t, i);
            al |= (uintptr_t)d[i].x[t] | (uintptr_t)d[i].m[t];
            p.x[i][t] = d[i].x[t];
            p.m[i][t] = d[i].m[t];
        }
        MMB_REQUIRE(al % 16 == 0, "mmb_masked_sum: tensor %d is not 16-byte aligned", i);
        p.dst[i] = d[i].dst; p.mo[i] = d[i].mo; p.n[i] = d[i].n; p.nterms[i] = d[i].nterms;
        p.blk_begin[i] = blk;
        blk += (int)((d[i].n + MS_PER_BLOCK - 1) / MS_PER_BLOCK);
    }
    p.blk_begin[k] = blk;
    MMB_HIP(hipSetDevice(device));
    if (blk == 0) return MMB_OK;
    hipLaunchKernelGGL(masked_sum_kernel, dim3(blk), dim3(256), 0, stream, p);
    MMB_HIP(hipGetLastError());
    return MMB_OK;
}

extern "C" int mmb_masked_mul(const float* const* a, const float* const* m, float* const* dst, const long* n, int k, int accumulate,
                              float keep, float scale, int device, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    MMB_REQUIRE(a && m && dst && n && k >= 1 && k <= MMB_MASK_MAX, "mmb_masked_mul: 1..%d tensors", MMB_MASK_MAX);
    MaskMulArgs p{};
    p.k = k; p.keep = keep; p.scale = scale;
    int blk = 0;
    for (int i = 0; i < k; ++i) {
        MMB_REQUIRE(a[i] && m[i] && dst[i] && n[i] >= 0, "mmb_masked_mul: null tensor %d", i);
        MMB_REQUIRE(((uintptr_t)a[i] | (uintptr_t)m[i] | (uintptr_t)dst[i]) % 16 == 0, "mmb_masked_mul: tensor %d is not 16-byte aligned", i);
        p.a[i] = a[i]; p.m[i] = m[i]; p.dst[i] = dst[i]; p.n[i] = n[i];
        p.blk_begin[i] = blk;
        blk += (int)((n[i] + MK_PER_BLOCK - 1) / MK_PER_BLOCK);
    }
    p.blk_begin[k] = blk;
    MMB_HIP(hipSetDevice(device));
    if (blk == 0) return MMB_OK;
    if (accumulate) hipLaunchKernelGGL(masked_mul_kernel<true>, dim3(blk), dim3(256), 0, stream, p);
    else hipLaunchKernelGGL(masked_mul_kernel<false>, dim3(blk), dim3(256), 0, stream, p);
    MMB_HIP(hipGetLastError());
    return MMB_OK;
}
