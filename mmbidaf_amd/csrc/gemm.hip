// fp32 GEMM on the exact-f32 matrix cores of gfx950 (v_mfma_f32_16x16x4_f32).
//
// Used by the LSTM layer for the hoisted input projection  Gx = x . W_ih^T + b  (reference:
// inside torch.nn.LSTM, layers/encoding.py:79,96) and for the BPTT weight / input gradients.
// One workgroup = WAVES waves stacked along M; each wave owns MT x NT tiles of 16x16.
//   A operand of the MFMA: lane l supplies A[m = l&15][k = l>>4]
//   B operand            : lane l supplies B[k = l>>4][n = l&15]
//   C/D                  : lane l holds    C[m = 4*(l>>4) + e][n = l&15], e = 0..3
// Operand tiles go through LDS in the orientation they have in memory (no transpose on the way
// in): a K-contiguous tile is stored [rows][BK+1] (conflict-free ds_read_b32 down a column), an
// M/N-contiguous tile is stored [BK][cols] with cols % 32 == 16 so that the two 16-lane halves of a
// 32-lane bank group land on disjoint banks.
#include "common.h"

namespace mmb {

constexpr int BK = 16;

__host__ __device__ constexpr int pad16mod32(int c) { return (c % 32 == 16) ? c : c + ((16 - c % 32) + 32) % 32; }

template <int WAVES, int MT, int NT, bool TA, bool TB>
__global__ __launch_bounds__(WAVES * 64) void gemm_kernel(const GemmArgs g, const int kchunk) {
    constexpr int BM = WAVES * MT * 16;
    constexpr int BN = NT * 16;
    constexpr int NTHR = WAVES * 64;
    constexpr int BMP = pad16mod32(BM);
    constexpr int BNP = pad16mod32(BN);
    constexpr int A_ELEMS = TA ? BK * BMP : BM * (BK + 1);
    constexpr int B_ELEMS = TB ? BN * (BK + 1) : BK * BNP;
    __shared__ __attribute__((aligned(16))) float As[A_ELEMS];
    __shared__ __attribute__((aligned(16))) float Bs[B_ELEMS];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, kg = lane >> 4;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int kb = blockIdx.z * kchunk;
    const int ke = min(g.K, kb + kchunk);

    const bool a_vec = (g.lda % 4 == 0) && ((reinterpret_cast<uintptr_t>(g.A) & 15) == 0);
    const bool b_vec = (g.ldb % 4 == 0) && ((reinterpret_cast<uintptr_t>(g.B) & 15) == 0);

    f4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};

    for (int k0 = kb; k0 < ke; k0 += BK) {
        // ---- stage A
        if (!TA) {  // (M,K) row-major: tile rows = m, 16 k's contiguous
            for (int c = tid; c < BM * 4; c += NTHR) {
                const int m = c >> 2, k4 = (c & 3) * 4;
                const int gm = m0 + m, gk = k0 + k4;
                float v[4] = {0.f, 0.f, 0.f, 0.f};
                if (gm < g.M) {
                    const float* src = g.A + (size_t)gm * g.lda + gk;
                    if (a_vec && gk + 3 < ke) {
                        const f4 t = *reinterpret_cast<const f4*>(src);
                        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (gk + e < ke) v[e] = src[e];
                    }
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) As[m * (BK + 1) + k4 + e] = v[e];
            }
        } else {  // (K,M) row-major: tile rows = k, m contiguous
            for (int c = tid; c < BK * (BM / 4); c += NTHR) {
                const int k = c / (BM / 4), m4 = (c % (BM / 4)) * 4;
                const int gk = k0 + k, gm = m0 + m4;
                f4 t = f4{0.f, 0.f, 0.f, 0.f};
                if (gk < ke) {
                    const float* src = g.A + (size_t)gk * g.lda + gm;
                    if (a_vec && gm + 3 < g.M) {
                        t = *reinterpret_cast<const f4*>(src);
                    } else {
                        if (gm + 0 < g.M) t.x = src[0];
                        if (gm + 1 < g.M) t.y = src[1];
                        if (gm + 2 < g.M) t.z = src[2];
                        if (gm + 3 < g.M) t.w = src[3];
                    }
                }
                *reinterpret_cast<f4*>(&As[k * BMP + m4]) = t;
            }
        }
        // ---- stage B
        if (TB) {  // (N,K) row-major
            for (int c = tid; c < BN * 4; c += NTHR) {
                const int n = c >> 2, k4 = (c & 3) * 4;
                const int gn = n0 + n, gk = k0 + k4;
                float v[4] = {0.f, 0.f, 0.f, 0.f};
                if (gn < g.N) {
                    const float* src = g.B + (size_t)gn * g.ldb + gk;
                    if (b_vec && gk + 3 < ke) {
                        const f4 t = *reinterpret_cast<const f4*>(src);
                        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (gk + e < ke) v[e] = src[e];
                    }
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) Bs[n * (BK + 1) + k4 + e] = v[e];
            }
        } else {  // (K,N) row-major, optional row shift inside periods of periodB rows
            for (int c = tid; c < BK * (BN / 4); c += NTHR) {
                const int k = c / (BN / 4), n4 = (c % (BN / 4)) * 4;
                int gk = k0 + k;
                const int gn = n0 + n4;
                bool ok = gk < ke;
                if (g.shiftB != 0) {
                    const int t = gk % g.periodB + g.shiftB;
                    ok = ok && t >= 0 && t < g.periodB;
                    gk += g.shiftB;
                }
                f4 t = f4{0.f, 0.f, 0.f, 0.f};
                if (ok) {
                    const float* src = g.B + (size_t)gk * g.ldb + gn;
                    if (b_vec && gn + 3 < g.N) {
                        t = *reinterpret_cast<const f4*>(src);
                    } else {
                        if (gn + 0 < g.N) t.x = src[0];
                        if (gn + 1 < g.N) t.y = src[1];
                        if (gn + 2 < g.N) t.z = src[2];
                        if (gn + 3 < g.N) t.w = src[3];
                    }
                }
                *reinterpret_cast<f4*>(&Bs[k * BNP + n4]) = t;
            }
        }
        __syncthreads();
        // ---- MFMA over the 16-deep tile: 4 k-steps of 4
#pragma unroll
        for (int ks = 0; ks < BK / 4; ++ks) {
            const int kk = ks * 4 + kg;
            float a[MT], b[NT];
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const int m = (wave * MT + i) * 16 + r;
                a[i] = TA ? As[kk * BMP + m] : As[m * (BK + 1) + kk];
            }
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int n = j * 16 + r;
                b[j] = TB ? Bs[n * (BK + 1) + kk] : Bs[kk * BNP + n];
            }
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] = mfma16(a[i], b[j], acc[i][j]);
        }
        __syncthreads();
    }

    // ---- epilogue
    const bool atomic = gridDim.z > 1;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int n = n0 + j * 16 + r;
        if (n >= g.N) continue;
        float bv = 0.f;
        if (blockIdx.z == 0) {
            if (g.bias) bv += g.bias[n];
            if (g.bias2) bv += g.bias2[n];
        }
        int nc = n;
        if (g.gate_H > 0) {
            const int H = g.gate_H, blk = n / (4 * H), rem = n % (4 * H);
            nc = blk * 4 * H + (rem % H) * 4 + rem / H;
        }
#pragma unroll
        for (int i = 0; i < MT; ++i) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int m = m0 + (wave * MT + i) * 16 + 4 * kg + e;
                if (m >= g.M) continue;
                float* dst = g.C + (size_t)m * g.ldc + nc;
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

template <int WAVES, int MT, int NT>
static int launch_cfg(const GemmArgs& g, int splitk, hipStream_t stream) {
    constexpr int BM = WAVES * MT * 16, BN = NT * 16;
    dim3 grid((g.N + BN - 1) / BN, (g.M + BM - 1) / BM, splitk);
    int kchunk = (g.K + splitk - 1) / splitk;
    kchunk = (kchunk + BK - 1) / BK * BK;
    dim3 block(WAVES * 64);
    ProfScope ps_(MMB_K_GEMM, stream);
    if (!g.ta && g.tb)
        hipLaunchKernelGGL((gemm_kernel<WAVES, MT, NT, false, true>), grid, block, 0, stream, g, kchunk);
    else if (!g.ta && !g.tb)
        hipLaunchKernelGGL((gemm_kernel<WAVES, MT, NT, false, false>), grid, block, 0, stream, g, kchunk);
    else if (g.ta && !g.tb)
        hipLaunchKernelGGL((gemm_kernel<WAVES, MT, NT, true, false>), grid, block, 0, stream, g, kchunk);
    else
        hipLaunchKernelGGL((gemm_kernel<WAVES, MT, NT, true, true>), grid, block, 0, stream, g, kchunk);
    MMB_HIP(hipGetLastError());
    return MMB_OK;
}

static int pick_splitk(const GemmArgs& g, int bm, int bn) {
    const long tiles = (long)((g.M + bm - 1) / bm) * ((g.N + bn - 1) / bn);
    if (tiles >= 192 || g.K < 512) return 1;
    long s = (512 + tiles - 1) / tiles;            // aim at ~2 workgroups per CU
    const long smax = g.K / 128;                   // keep >= 128-deep K slices
    if (s > smax) s = smax;
    if (s > 64) s = 64;
    return s < 1 ? 1 : (int)s;
}

static void pick_tile(const GemmArgs& g, int* bm, int* bn, int* cfg) {
    // N tile: 208 (13 tiles) unless N is small enough for 112 (7 tiles)
    const bool narrow = g.N <= 112;
    // M tile: 128 (4 waves x 2) for tall problems, 160 (5 x 2) when M is a multiple of 160-ish LSTM gate blocks
    const bool m160 = (g.M % 160 == 0) || (g.M <= 800 && g.M > 128);
    *bn = narrow ? 112 : 208;
    *bm = m160 ? 160 : 128;
    *cfg = (m160 ? 2 : 0) + (narrow ? 1 : 0);
}

int gemm_splitk_for(const GemmArgs& g) {
    int bm, bn, cfg;
    pick_tile(g, &bm, &bn, &cfg);
    return pick_splitk(g, bm, bn);
}

int gemm_launch(const GemmArgs& g, hipStream_t stream) {
    if (g.M <= 0 || g.N <= 0) return MMB_OK;
    int bm, bn, cfg;
    pick_tile(g, &bm, &bn, &cfg);
    const int splitk = pick_splitk(g, bm, bn);
    if (splitk > 1 && !g.accumulate)
        MMB_HIP(hipMemset2DAsync(g.C, (size_t)g.ldc * sizeof(float), 0, (size_t)g.N * sizeof(float), g.M, stream));
    switch (cfg) {
        case 0: return launch_cfg<4, 2, 13>(g, splitk, stream);
        case 1: return launch_cfg<4, 2, 7>(g, splitk, stream);
        case 2: return launch_cfg<5, 2, 13>(g, splitk, stream);
        default: return launch_cfg<5, 2, 7>(g, splitk, stream);
    }
}

}  // namespace mmb

extern "C" int mmb_gemm_f32(const float* A, const float* Bm, float* C, const float* bias, int M, int N, int K,
                            int lda, int ldb, int ldc, int ta, int tb, int accumulate, int device, void* stream) {
    MMB_REQUIRE(A && Bm && C, "mmb_gemm_f32: null pointer");
    MMB_REQUIRE(M >= 0 && N >= 0 && K > 0, "mmb_gemm_f32: bad sizes M=%d N=%d K=%d", M, N, K);
    MMB_HIP(hipSetDevice(device));
    mmb::GemmArgs g{};
    g.A = A; g.B = Bm; g.C = C; g.bias = bias; g.bias2 = nullptr;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.ta = ta; g.tb = tb; g.accumulate = accumulate; g.gate_H = 0; g.shiftB = 0; g.periodB = 1;
    return mmb::gemm_launch(g, static_cast<hipStream_t>(stream));
}
