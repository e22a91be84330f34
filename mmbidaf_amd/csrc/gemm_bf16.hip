// fp32-accurate GEMM on the bf16 matrix cores of gfx950 (v_mfma_f32_16x16x32_bf16, fp32 accumulate).
//
// Each fp32 operand element is split EXACTLY into three bf16 terms  x = x0 + x1 + x2  (8+8+8 mantissa bits:
// x0 = bf16(x), x1 = bf16(x - x0), x2 = bf16(x - x0 - x1); the two subtractions are exact in fp32), and the product
// is accumulated from the six cross terms of order <= 2:
//     a.b ~= a0b2 + a1b1 + a2b0 + a0b1 + a1b0 + a0b0          (dropped terms <= 2^-24 |a||b|)
// Every bf16 x bf16 product is exact in the fp32 accumulator, so the result carries fp32-level error (the same
// ~1e-7 relative error as an fp32 FMA chain; tests/test_gpu_parity.py::test_gemm) at 16/6 = 2.7x the peak rate of
// the exact-f32 MFMA.  NS = 2 keeps two terms and three products (error ~2^-16): available, not the default.
//
// Structure: workgroup = WAVES waves x (MT*16) rows x (NT*16) columns, K tile = 32 = one MFMA K.  fp32 tiles are
// prefetched global -> registers while the previous tile is multiplied; they are split to bf16 on the way into LDS
// (one [rows][32] bf16 image per term, XOR-swizzled so the b128 fragment reads are conflict-free).  An operand whose
// contiguous dimension is M/N is transposed in registers by giving each thread an 8(k) x 4(m) patch.
#include <stdlib.h>

#include "common.h"

namespace mmb {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int KT = 32;       // K tile
constexpr int ROWB = 64;     // LDS bytes per tile row (32 bf16)

__device__ __forceinline__ int swz_off(int row, int oct) {  // byte offset of k-octet `oct` of tile row `row`
    return row * ROWB + ((oct ^ (((row >> 3) & 1) << 1)) << 4);
}

template <int NS>
__device__ __forceinline__ void split_store(char* const (&img)[3], int off, const float* x) {
    bf16x8 h0, h1, h2;
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
    *reinterpret_cast<bf16x8*>(img[0] + off) = h0;
    if (NS >= 2) *reinterpret_cast<bf16x8*>(img[1] + off) = h1;
    if (NS == 3) *reinterpret_cast<bf16x8*>(img[2] + off) = h2;
}

// TA/TB as in mmb_gemm_f32: A is (M,K) [TA=0, K contiguous] or (K,M) [TA=1, M contiguous];
//                           B is (N,K) [TB=1, K contiguous] or (K,N) [TB=0, N contiguous].
template <int WAVES, int MT, int NT, bool TA, bool TB, int NS>
__global__ __launch_bounds__(WAVES * 64) void gemm_bf16_kernel(const GemmArgs g, const int kchunk) {
    // batched mode (g.batch > 1, strided operands): blockIdx.z = product, no K split
    const int bz = g.batch > 1 ? blockIdx.z : 0, kz = g.batch > 1 ? 0 : blockIdx.z;
    const float* gA = g.A + (size_t)bz * g.sA;
    const float* gB = g.B + (size_t)bz * g.sB;
    float* gC = g.C + (size_t)bz * g.sC;
    constexpr int BM = WAVES * MT * 16, BN = NT * 16, NTHR = WAVES * 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* imgA[3];
    char* imgB[3];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        imgA[s] = smem + s * (BM * ROWB);
        imgB[s] = smem + NS * (BM * ROWB) + s * (BN * ROWB);
    }

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, kg = lane >> 4;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int kb = kz * kchunk;
    const int ke = min(g.K, kb + kchunk);

    // ---- prefetch registers: "units" of 8 k-values per tile row (K-contiguous) or 8k x 4m patches (M/N-contiguous)
    constexpr int A_UNITS = TA ? BM : BM * 4, B_UNITS = TB ? BN * 4 : BN;
    constexpr int A_PT = (A_UNITS + NTHR - 1) / NTHR, B_PT = (B_UNITS + NTHR - 1) / NTHR;
    constexpr int A_F = TA ? 32 : 8, B_F = TB ? 8 : 32;  // floats per unit
    float ra[A_PT][A_F], rb[B_PT][B_F];

    auto load_tile = [&](int k0) {
#pragma unroll
        for (int i = 0; i < A_PT; ++i) {
            const int c = tid + i * NTHR;
            if (A_UNITS % NTHR == 0 || c < A_UNITS) {
                if constexpr (!TA) {
                    const int row = c >> 2, oct = c & 3;
                    const float* src = gA + (size_t)min(m0 + row, g.M - 1) * g.lda + k0 + 8 * oct;
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        f4 t = f4{0.f, 0.f, 0.f, 0.f};
                        if (k0 + 8 * oct + 4 * h < ke) t = *reinterpret_cast<const f4*>(src + 4 * h);
                        ra[i][4 * h + 0] = t.x; ra[i][4 * h + 1] = t.y; ra[i][4 * h + 2] = t.z; ra[i][4 * h + 3] = t.w;
                    }
                } else {
                    const int m4 = c % (BM / 4), oct = c / (BM / 4);
                    const float* src = gA + (size_t)(k0 + 8 * oct) * g.lda + min(m0 + 4 * m4, g.M - 4);
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        f4 t = f4{0.f, 0.f, 0.f, 0.f};
                        if (k0 + 8 * oct + j < ke) t = *reinterpret_cast<const f4*>(src + (size_t)j * g.lda);
                        ra[i][4 * j + 0] = t.x; ra[i][4 * j + 1] = t.y; ra[i][4 * j + 2] = t.z; ra[i][4 * j + 3] = t.w;
                    }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < B_PT; ++i) {
            const int c = tid + i * NTHR;
            if (B_UNITS % NTHR == 0 || c < B_UNITS) {
                if constexpr (TB) {
                    const int row = c >> 2, oct = c & 3;
                    const int gn = min(n0 + row, g.N - 1);
                    int srow = gn;
                    if (g.gate_H > 0) {  // output column u*4+g of each 4H block comes from weight row g*H+u
                        const int H = g.gate_H, blk = gn / (4 * H), rem = gn % (4 * H);
                        srow = blk * 4 * H + (rem & 3) * H + (rem >> 2);
                    }
                    const float* src = gB + (size_t)srow * g.ldb + k0 + 8 * oct;
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        f4 t = f4{0.f, 0.f, 0.f, 0.f};
                        if (k0 + 8 * oct + 4 * h < ke) t = *reinterpret_cast<const f4*>(src + 4 * h);
                        rb[i][4 * h + 0] = t.x; rb[i][4 * h + 1] = t.y; rb[i][4 * h + 2] = t.z; rb[i][4 * h + 3] = t.w;
                    }
                } else {
                    const int n4 = c % (BN / 4), oct = c / (BN / 4);
                    int col = min(n0 + 4 * n4, g.N - 4), ld = g.ldb, shift = g.shiftB;
                    const float* base = gB;
                    if (g.nseg > 0) {  // virtual concatenation [x | y_fwd | y_rev]: pick this patch's segment
                        int sg = 0;
                        while (sg < g.nseg - 1 && col >= g.seg_cols[sg]) col -= g.seg_cols[sg++];
                        base = g.seg_ptr[sg];
                        ld = g.seg_ld[sg];
                        shift = g.seg_shift[sg];
                    }
                    const float* src = base + (size_t)(k0 + 8 * oct) * ld + col;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int gk = k0 + 8 * oct + j;
                        bool ok = gk < ke;
                        if (shift != 0) {  // read row gk+shift, zero when that leaves the period (sample) of gk
                            const int tt = gk % g.periodB + shift;
                            ok = ok && tt >= 0 && tt < g.periodB;
                        }
                        f4 t = f4{0.f, 0.f, 0.f, 0.f};
                        if (ok) t = *reinterpret_cast<const f4*>(src + ((ptrdiff_t)j + shift) * ld);
                        rb[i][4 * j + 0] = t.x; rb[i][4 * j + 1] = t.y; rb[i][4 * j + 2] = t.z; rb[i][4 * j + 3] = t.w;
                    }
                }
            }
        }
    };

    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < A_PT; ++i) {
            const int c = tid + i * NTHR;
            if (A_UNITS % NTHR == 0 || c < A_UNITS) {
                if constexpr (!TA) {
                    split_store<NS>(imgA, swz_off(c >> 2, c & 3), ra[i]);
                } else {
                    const int m4 = c % (BM / 4), oct = c / (BM / 4);
#pragma unroll
                    for (int mm = 0; mm < 4; ++mm) {
                        float x[8];
#pragma unroll
                        for (int j = 0; j < 8; ++j) x[j] = ra[i][4 * j + mm];
                        split_store<NS>(imgA, swz_off(4 * m4 + mm, oct), x);
                    }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < B_PT; ++i) {
            const int c = tid + i * NTHR;
            if (B_UNITS % NTHR == 0 || c < B_UNITS) {
                if constexpr (TB) {
                    split_store<NS>(imgB, swz_off(c >> 2, c & 3), rb[i]);
                } else {
                    const int n4 = c % (BN / 4), oct = c / (BN / 4);
#pragma unroll
                    for (int mm = 0; mm < 4; ++mm) {
                        float x[8];
#pragma unroll
                        for (int j = 0; j < 8; ++j) x[j] = rb[i][4 * j + mm];
                        split_store<NS>(imgB, swz_off(4 * n4 + mm, oct), x);
                    }
                }
            }
        }
    };

    f4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};

    // fragment read offsets: lane (r,kg) reads 8 bf16 of tile row (.. + r), k-octet kg
    int offA[MT], offB[NT];
#pragma unroll
    for (int i = 0; i < MT; ++i) offA[i] = swz_off((wave * MT + i) * 16 + r, kg);
#pragma unroll
    for (int j = 0; j < NT; ++j) offB[j] = swz_off(j * 16 + r, kg);

    if (kb < ke) load_tile(kb);
    for (int k0 = kb; k0 < ke; k0 += KT) {
        store_tile();
        __syncthreads();
        if (k0 + KT < ke) load_tile(k0 + KT);
        bf16x8 a[MT][NS];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int s = 0; s < NS; ++s) a[i][s] = *reinterpret_cast<const bf16x8*>(imgA[s] + offA[i]);
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            bf16x8 b[NS];
#pragma unroll
            for (int s = 0; s < NS; ++s) b[s] = *reinterpret_cast<const bf16x8*>(imgB[s] + offB[j]);
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                f4 c = acc[i][j];
                if (NS == 3) {
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][0], b[NS == 3 ? 2 : 0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][NS == 3 ? 1 : 0], b[NS == 3 ? 1 : 0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][NS == 3 ? 2 : 0], b[0], c, 0, 0, 0);
                }
                if (NS >= 2) {
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][0], b[NS >= 2 ? 1 : 0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][NS >= 2 ? 1 : 0], b[0], c, 0, 0, 0);
                }
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][0], b[0], c, 0, 0, 0);
                acc[i][j] = c;
            }
        }
        __syncthreads();
    }

    // ---- epilogue: lane (r,kg) holds C[m = 4kg+e][n = r] of each 16x16 tile
    const bool atomic = g.batch <= 1 && gridDim.z > 1;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int n = n0 + j * 16 + r;
        if (n >= g.N) continue;
        float bv = 0.f;
        if (kz == 0 && (g.bias || g.bias2)) {
            int srow = n;
            if (g.gate_H > 0) {
                const int H = g.gate_H, blk = n / (4 * H), rem = n % (4 * H);
                srow = blk * 4 * H + (rem & 3) * H + (rem >> 2);
            }
            if (g.bias) bv += g.bias[srow];
            if (g.bias2) bv += g.bias2[srow];
        }
#pragma unroll
        for (int i = 0; i < MT; ++i) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int m = m0 + (wave * MT + i) * 16 + 4 * kg + e;
                if (m >= g.M) continue;
                float* dst = gC + (size_t)m * g.ldc + n;
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

template <int WAVES, int MT, int NT, int NS>
static int launch_bf16(const GemmArgs& g, int splitk, hipStream_t stream) {
    constexpr int BM = WAVES * MT * 16, BN = NT * 16;
    dim3 grid((g.N + BN - 1) / BN, (g.M + BM - 1) / BM, g.batch > 1 ? g.batch : splitk);
    int kchunk = (g.K + splitk - 1) / splitk;
    kchunk = (kchunk + KT - 1) / KT * KT;
    const size_t lds = (size_t)NS * (BM + BN) * ROWB;
    dim3 block(WAVES * 64);
    ProfScope ps_(MMB_K_GEMM, stream);
#define MMB_L(TA_, TB_)                                                                                               \
    do {                                                                                                              \
        auto kern = gemm_bf16_kernel<WAVES, MT, NT, TA_, TB_, NS>;                                                    \
        static PerDeviceOnce attr_set;                                                                                \
        if (attr_set.pending()) {                                                                                     \
            MMB_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                        (int)lds));                                                                   \
            attr_set.mark();                                                                                          \
        }                                                                                                             \
        hipLaunchKernelGGL(kern, grid, block, lds, stream, g, kchunk);                                                \
    } while (0)
    if (!g.ta && g.tb)
        MMB_L(false, true);
    else if (!g.ta && !g.tb)
        MMB_L(false, false);
    else if (g.ta && !g.tb)
        MMB_L(true, false);
    else
        MMB_L(true, true);
#undef MMB_L
    MMB_HIP(hipGetLastError());
    return MMB_OK;
}

bool gemm_bf16_eligible(const GemmArgs& g) {
    if (g.batch > 1 && (g.use_ptrs || g.nseg > 0 || g.sA % 4 || g.sB % 4 || g.batch > 65535)) return false;   // batched: strided operands only
    const bool al = (reinterpret_cast<uintptr_t>(g.A) % 16 == 0) && (g.nseg > 0 || reinterpret_cast<uintptr_t>(g.B) % 16 == 0);
    if (!gemm_segments_ok(g)) return false;
    if (!al || g.lda % 4 || (g.nseg == 0 && g.ldb % 4) || g.K % 4) return false;
    if (g.ta && (g.M % 4 || g.M < 4)) return false;
    if (!g.tb && (g.N % 4 || g.N < 4)) return false;
    return true;
}

int gemm_bf16_launch(const GemmArgs& g, int ns, hipStream_t stream) {
    // column tile: 112 for narrow outputs and where it wastes less padding than 208 (N = 256: 336 instead of 416 columns)
    const bool narrow = g.N <= 112 || ((g.N + 111) / 112) * 112 < ((g.N + 207) / 208) * 208;
    const int bn = narrow ? 112 : 208;
    const long nb = g.batch > 1 ? g.batch : 1;
    const long tiles128 = (long)((g.M + 127) / 128) * ((g.N + bn - 1) / bn) * nb;
    const bool small = tiles128 < 320 && g.K < 4096;  // too few 128-row tiles (and no split-K) to fill 256 CUs twice
    const int bm = small ? 64 : 128;
    const long tiles = (long)((g.M + bm - 1) / bm) * ((g.N + bn - 1) / bn) * nb;
    int splitk = 1;
    if (g.batch <= 1 && tiles < 192 && g.K >= 512) {
        long s = (512 + tiles - 1) / tiles;
        const long smax = g.K / 256;
        if (s > smax) s = smax;
        if (s > 64) s = 64;
        splitk = s < 1 ? 1 : (int)s;
    }
    if (splitk > 1 && !g.accumulate)
        MMB_HIP(hipMemset2DAsync(g.C, (size_t)g.ldc * sizeof(float), 0, (size_t)g.N * sizeof(float), g.M, stream));
    const int cfg = (small ? 2 : 0) + (narrow ? 1 : 0);
    if (ns == 3) {
        switch (cfg) {
            case 0: return launch_bf16<4, 2, 13, 3>(g, splitk, stream);
            case 1: return launch_bf16<4, 2, 7, 3>(g, splitk, stream);
            case 2: return launch_bf16<4, 1, 13, 3>(g, splitk, stream);
            default: return launch_bf16<4, 1, 7, 3>(g, splitk, stream);
        }
    }
    if (ns == 1) {      // one bf16 term, one product: the bf16 operand mode's batched attention products (tolerance 3e-2 of scale)
        switch (cfg) {
            case 0: return launch_bf16<4, 2, 13, 1>(g, splitk, stream);
            case 1: return launch_bf16<4, 2, 7, 1>(g, splitk, stream);
            case 2: return launch_bf16<4, 1, 13, 1>(g, splitk, stream);
            default: return launch_bf16<4, 1, 7, 1>(g, splitk, stream);
        }
    }
    switch (cfg) {
        case 0: return launch_bf16<4, 2, 13, 2>(g, splitk, stream);
        case 1: return launch_bf16<4, 2, 7, 2>(g, splitk, stream);
        case 2: return launch_bf16<4, 1, 13, 2>(g, splitk, stream);
        default: return launch_bf16<4, 1, 7, 2>(g, splitk, stream);
    }
}

}  // namespace mmb
