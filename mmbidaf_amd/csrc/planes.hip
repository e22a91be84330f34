// "Planes" GEMM: fp32-accurate products on the bf16 matrix cores with operands split ONCE.
//
//   split pass   fp32 matrix -> three bf16 planes x = x0 + x1 + x2 (exact 8+8+8-bit split, see gemm_bf16.hip), written
//                K-contiguous with K padded to 32.  The pass is bandwidth-bound and does every layout change the
//                GEMMs need, so the GEMM kernel itself is a single "NT" form (both operands K-contiguous):
//                  split_rows       rows as they are; optional row stacking (forward / reverse weights) and the
//                                   LSTM gate interleave as a row permutation
//                  split_transpose  columns become plane rows; the source may be a virtual concatenation of up to 3
//                                   column blocks, each with its own row shift inside periods of T rows
//                                   ([x | y_fwd(t-1) | y_rev(t+1)] for the weight gradients)
//   gemm_planes  C (M,N) fp32 = sum over the 6 cross terms of order <= 2 of A_i . B_j^T, v_mfma_f32_16x16x32_bf16,
//                workgroup 4 waves x (MT*16) x (NT*16), K tile 32, LDS-DMA (global_load_lds_dwordx4) into a
//                double-buffered, XOR-swizzled [plane][row][64 B] image (swizzle applied on the per-lane SOURCE
//                address, LDS destination linear), next tile's DMA in flight under the current tile's 6*MT*NT MFMAs.
#include <stdlib.h>

#include "common.h"

namespace mmb {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void split3(const float* x, bf16x8& h0, bf16x8& h1, bf16x8& h2) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const __bf16 a = (__bf16)x[j];
        float r = x[j] - (float)a;
        const __bf16 b = (__bf16)r;
        r -= (float)b;
        h0[j] = a;
        h1[j] = b;
        h2[j] = (__bf16)r;
    }
}

// ------------------------------------------------------------------------------------------ split passes
// planes[s][dst_row][Cp]; rows r < R1 come from src1, the rest from src2 (stacked); gate_H > 0 permutes each 4H block
// of rows so that plane row u*4+g holds source row g*H+u.  Optional bias_out[dst_row] = b1[src_row] + b2[src_row].
__global__ __launch_bounds__(256) void split_rows_kernel(const SplitRowsArgs a) {
    const int oct_per_row = a.Cp / 8;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)a.R * oct_per_row) return;
    const int r = idx / oct_per_row, oc = idx % oct_per_row;
    const bool second = r >= a.R1;
    const float* src = (second ? a.src2 + (size_t)(r - a.R1) * a.ld : a.src1 + (size_t)r * a.ld) + 8 * oc;
    float x[8];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        f4 t = f4{0.f, 0.f, 0.f, 0.f};
        if (8 * oc + 4 * h < a.C) t = *reinterpret_cast<const f4*>(src + 4 * h);
        x[4 * h] = t.x; x[4 * h + 1] = t.y; x[4 * h + 2] = t.z; x[4 * h + 3] = t.w;
    }
    int dr = r;
    if (a.gate_H > 0) {  // source row g*H+u of a 4H block -> plane row u*4+g
        const int H = a.gate_H, blk = r / (4 * H), rem = r % (4 * H);
        dr = blk * 4 * H + (rem % H) * 4 + rem / H;
    }
    bf16x8 h0, h1, h2;
    split3(x, h0, h1, h2);
    __bf16* dst = a.planes + (size_t)dr * a.Cp + 8 * oc;
    *reinterpret_cast<bf16x8*>(dst) = h0;
    *reinterpret_cast<bf16x8*>(dst + a.plane_stride) = h1;
    *reinterpret_cast<bf16x8*>(dst + 2 * a.plane_stride) = h2;
    if (a.bias_out && oc == 0) {
        const int lr = second ? r - a.R1 : r;
        a.bias_out[dr] = second ? a.b1b[lr] + a.b2b[lr] : a.b1a[lr] + a.b2a[lr];
    }
}

// planes[s][col (global, over the concatenated segments)][Rp]: the transpose of a virtual (R x sum cols) matrix.
__global__ __launch_bounds__(256) void split_transpose_kernel(const SplitTArgs a) {
    __shared__ float tile[32][65];
    const int k0 = blockIdx.x * 32, c0 = blockIdx.y * 64;
    const int t = threadIdx.x;
    // load 32 source rows x 64 columns (two float4 per thread), per-chunk segment lookup, shifted rows, zero outside
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int kk = (t >> 4) + 16 * h, col = c0 + (t & 15) * 4;
        f4 v = f4{0.f, 0.f, 0.f, 0.f};
        if (col < a.Ctot && k0 + kk < a.R) {
            int c = col, sg = 0;
            while (sg < a.nseg - 1 && c >= a.seg_cols[sg]) c -= a.seg_cols[sg++];
            const int sh = a.seg_shift[sg];
            const int gk = k0 + kk;
            const int tt = gk % a.period + sh;
            if (tt >= 0 && tt < a.period) {
                const int sr = gk + sh;
                const float* base = (a.stack_ptr && sr >= a.stack_R1) ? a.stack_ptr + (size_t)(sr - a.stack_R1) * a.seg_ld[sg]
                                                                      : a.seg_ptr[sg] + (size_t)sr * a.seg_ld[sg];
                v = *reinterpret_cast<const f4*>(base + c);
            }
        }
        float* d = &tile[kk][(t & 15) * 4];
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
    __syncthreads();
    // each thread: one output row (source column), one k-octet
    const int oc = t & 3, cl = t >> 2;
    const int col = c0 + cl;
    if (col >= a.Ctot || k0 + 8 * oc >= a.Rp) return;
    float x[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = tile[8 * oc + j][cl];
    bf16x8 h0, h1, h2;
    split3(x, h0, h1, h2);
    __bf16* dst = a.planes + (size_t)col * a.Rp + k0 + 8 * oc;
    *reinterpret_cast<bf16x8*>(dst) = h0;
    *reinterpret_cast<bf16x8*>(dst + a.plane_stride) = h1;
    *reinterpret_cast<bf16x8*>(dst + 2 * a.plane_stride) = h2;
}

// ------------------------------------------------------------------------------------------ GEMM on planes
__device__ __forceinline__ int pl_swz(int row) { return ((row >> 3) & 1) << 1; }

// STAGES = 2: next tile's DMA in flight under this tile's MFMAs (1 workgroup per CU at the large tiles);
// STAGES = 1: DMA, wait, multiply -- latency is hidden across the 2-3 workgroups that then fit on a CU.
template <int MT, int NT, int STAGES>
__global__ __launch_bounds__(256) void gemm_planes_kernel(const PlanesGemmArgs g, const int kchunk) {
    constexpr int BM = 4 * MT * 16, BN = NT * 16, RT = BM + BN;  // rows per plane image
    constexpr int STAGE = 3 * RT * 64;                           // bytes per stage
    constexpr int NDMA = 3 * RT / 16;                            // 1-KiB wave-instructions per stage
    constexpr int PER_WAVE = (NDMA + 3) / 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, kg = lane >> 4;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int kb = blockIdx.z * kchunk;
    const int ke = min(g.K, kb + kchunk);

    // per-lane source pointers of this wave's DMA instructions (loop-invariant except for the K offset)
    const __bf16* src[PER_WAVE];
#pragma unroll
    for (int k = 0; k < PER_WAVE; ++k) {
        const int q = min(wave + 4 * k, NDMA - 1);
        const int rr = 16 * q + (lane >> 2);        // row of the [3][RT] image
        const int plane = rr / RT, rl = rr % RT;
        const int slot = lane & 3;
        if (rl < BM) {
            const int oct = slot ^ pl_swz(rl);
            src[k] = g.A + plane * g.a_plane + (size_t)min(m0 + rl, g.M - 1) * g.lda + kb + 8 * oct;
        } else {
            const int tr = rl - BM;
            const int oct = slot ^ pl_swz(tr);
            src[k] = g.B + plane * g.b_plane + (size_t)min(n0 + tr, g.N - 1) * g.ldb + kb + 8 * oct;
        }
    }
    // one LDS-DMA piece (1 KiB) of this wave; pieces are issued one per MFMA group inside the tile loop, not as a burst
    auto dma_piece = [&](int stage, int k) {
        const int q = wave + 4 * k;
        if (NDMA % 4 == 0 || q < NDMA)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src[k],
                                             (__attribute__((address_space(3))) void*)(smem + stage * STAGE + q * 1024), 16, 0, 0);
        src[k] += 32;
    };

    f4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};

    int offA[MT], offB[NT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int tr = (wave * MT + i) * 16 + r;
        offA[i] = tr * 64 + ((kg ^ pl_swz(tr)) << 4);
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int tr = j * 16 + r;
        offB[j] = (BM + tr) * 64 + ((kg ^ pl_swz(tr)) << 4);
    }

    auto multiply = [&](const char* img, int next_stage, bool more) {
        bf16x8 a[MT][3];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int s = 0; s < 3; ++s) a[i][s] = *reinterpret_cast<const bf16x8*>(img + s * (RT * 64) + offA[i]);
        constexpr int DPJ = (PER_WAVE + NT - 1) / NT;  // DMA pieces issued per n-tile of MFMAs
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            bf16x8 b[3];
#pragma unroll
            for (int s = 0; s < 3; ++s) b[s] = *reinterpret_cast<const bf16x8*>(img + s * (RT * 64) + offB[j]);
            if (STAGES == 2 && more) {
#pragma unroll
                for (int d = 0; d < DPJ; ++d)
                    if (j * DPJ + d < PER_WAVE) dma_piece(next_stage, j * DPJ + d);
            }
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                f4 c = acc[i][j];
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][0], b[2], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][1], b[1], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][2], b[0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][0], b[1], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][1], b[0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][0], b[0], c, 0, 0, 0);
                acc[i][j] = c;
            }
        }
    };

    if (STAGES == 2) {
        int stage = 0;
        if (kb < ke) {
#pragma unroll
            for (int k = 0; k < PER_WAVE; ++k) dma_piece(0, k);
        }
        __syncthreads();
        for (int k0 = kb; k0 < ke; k0 += 32) {
            multiply(smem + stage * STAGE, stage ^ 1, k0 + 32 < ke);
            __syncthreads();  // retires the DMA of the next stage (compiler's vmcnt(0)) and frees this one
            stage ^= 1;
        }
    } else {
        for (int k0 = kb; k0 < ke; k0 += 32) {
#pragma unroll
            for (int k = 0; k < PER_WAVE; ++k) dma_piece(0, k);
            __syncthreads();
            multiply(smem, 0, false);
            __syncthreads();
        }
    }

    const bool atomic = gridDim.z > 1;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int n = n0 + j * 16 + r;
        if (n >= g.N) continue;
        const float bv = (blockIdx.z == 0 && g.bias) ? g.bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int m = m0 + (wave * MT + i) * 16 + 4 * kg + e;
                if (m >= g.M) continue;
                float* dst = g.C + (size_t)m * g.ldc + n;
                const float v = acc[i][j][e] + bv;
                if (atomic)
                    atomicAdd(dst, v);
                else if (g.accumulate)
                    *dst += v;
                else
                    *dst = v;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------ host side
int planes_split_rows(const SplitRowsArgs& a, hipStream_t stream) {
    const long n = (long)a.R * (a.Cp / 8);
    ProfScope ps_(MMB_K_SPLIT, stream);
    hipLaunchKernelGGL(split_rows_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, a);
    MMB_HIP(hipGetLastError());
    return MMB_OK;
}

int planes_split_transpose(const SplitTArgs& a, hipStream_t stream) {
    ProfScope ps_(MMB_K_SPLIT, stream);
    hipLaunchKernelGGL(split_transpose_kernel, dim3((a.Rp + 31) / 32, (a.Ctot + 63) / 64), dim3(256), 0, stream, a);
    MMB_HIP(hipGetLastError());
    return MMB_OK;
}

template <int MT, int NT, int STAGES>
static int launch_planes(const PlanesGemmArgs& g, int splitk, hipStream_t stream) {
    constexpr int BM = 4 * MT * 16, BN = NT * 16;
    const size_t lds = (size_t)STAGES * 3 * (BM + BN) * 64;
    auto kern = gemm_planes_kernel<MT, NT, STAGES>;
    static bool attr_set = false;
    if (!attr_set) {
        MMB_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    dim3 grid((g.N + BN - 1) / BN, (g.M + BM - 1) / BM, splitk);
    int kchunk = (g.K + splitk - 1) / splitk;
    kchunk = (kchunk + 31) / 32 * 32;
    ProfScope ps_(MMB_K_GEMM, stream);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, stream, g, kchunk);
    MMB_HIP(hipGetLastError());
    return MMB_OK;
}

int planes_gemm(const PlanesGemmArgs& g, hipStream_t stream) {
    static int tune = -2;   // MMB_PLANES_TUNE = <bm: 0 auto | 1 force 64 | 2 force 128><stages: 1 | 2>, e.g. "21" (tuning aid)
    if (tune == -2) {
        const char* e = getenv("MMB_PLANES_TUNE");
        tune = e ? atoi(e) : -1;
    }
    const bool narrow = g.N <= 112;
    const int bn = narrow ? 112 : 208;
    const long tiles128 = (long)((g.M + 127) / 128) * ((g.N + bn - 1) / bn);
    bool small = tiles128 < 200 && g.K < 4096;
    int stages = 2;
    if (tune >= 0) {
        if (tune / 10 == 1) small = true;
        if (tune / 10 == 2) small = false;
        stages = (tune % 10 == 1) ? 1 : 2;
    }
    const int bm = small ? 64 : 128;
    const long tiles = (long)((g.M + bm - 1) / bm) * ((g.N + bn - 1) / bn);
    int splitk = 1;
    if (tiles < 160 && g.K >= 1024) {
        long s = (256 + tiles - 1) / tiles;
        const long smax = g.K / 512;
        if (s > smax) s = smax;
        if (s > 32) s = 32;
        splitk = s < 1 ? 1 : (int)s;
    }
    if (splitk > 1 && !g.accumulate)
        MMB_HIP(hipMemset2DAsync(g.C, (size_t)g.ldc * sizeof(float), 0, (size_t)g.N * sizeof(float), g.M, stream));
    const int cfg = (small ? 2 : 0) + (narrow ? 1 : 0);
    if (stages == 1) {
        switch (cfg) {
            case 0: return launch_planes<2, 13, 1>(g, splitk, stream);
            case 1: return launch_planes<2, 7, 1>(g, splitk, stream);
            case 2: return launch_planes<1, 13, 1>(g, splitk, stream);
            default: return launch_planes<1, 7, 1>(g, splitk, stream);
        }
    }
    switch (cfg) {
        case 0: return launch_planes<2, 13, 2>(g, splitk, stream);
        case 1: return launch_planes<2, 7, 2>(g, splitk, stream);
        case 2: return launch_planes<1, 13, 2>(g, splitk, stream);
        default: return launch_planes<1, 7, 2>(g, splitk, stream);
    }
}

}  // namespace mmb

// C = A (M,K) . B (N,K)^T + bias through the split passes and the planes kernel (tests / tools): ws needs
// 6 * (M + N) * roundup(K, 32) bytes.
extern "C" int mmb_gemm_nt_planes(const float* A, const float* Bm, float* C, const float* bias, int M, int N, int K,
                                  void* ws, size_t ws_bytes, int device, void* stream_) {
    using namespace mmb;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    MMB_REQUIRE(A && Bm && C && ws && M > 0 && N > 0 && K > 0 && K % 4 == 0, "mmb_gemm_nt_planes: bad argument");
    const int Kp = (K + 31) / 32 * 32;
    MMB_REQUIRE(ws_bytes >= (size_t)6 * (M + N) * Kp, "mmb_gemm_nt_planes: workspace too small");
    MMB_HIP(hipSetDevice(device));
    bf16_t* aP = static_cast<bf16_t*>(ws);
    bf16_t* bP = aP + (size_t)3 * M * Kp;
    SplitRowsArgs sa{};
    sa.src1 = A; sa.src2 = A; sa.R1 = M; sa.R = M; sa.C = K; sa.ld = K; sa.Cp = Kp; sa.planes = aP; sa.plane_stride = (size_t)M * Kp;
    if (int rc = planes_split_rows(sa, stream)) return rc;
    SplitRowsArgs sb{};
    sb.src1 = Bm; sb.src2 = Bm; sb.R1 = N; sb.R = N; sb.C = K; sb.ld = K; sb.Cp = Kp; sb.planes = bP; sb.plane_stride = (size_t)N * Kp;
    if (int rc = planes_split_rows(sb, stream)) return rc;
    PlanesGemmArgs g{};
    g.A = aP; g.a_plane = sa.plane_stride; g.lda = Kp; g.B = bP; g.b_plane = sb.plane_stride; g.ldb = Kp;
    g.C = C; g.ldc = N; g.bias = bias; g.M = M; g.N = N; g.K = Kp;
    return planes_gemm(g, stream);
}
