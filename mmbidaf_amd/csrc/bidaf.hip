// Fused BiDAF attention for gfx950 (forward + backward), fp32 on the exact-f32 matrix cores.
// Replaces BiDAFAttention.forward / get_similarity_matrix / masked_softmax of the reference
// (layers/attention.py:37-98) and their autograd.  The (B,T,M) similarity matrix, both softmaxes
// and the (B,T,T) product s1.s2^T of the reference are never materialised:
//
//   S_ij  = r_i + c_j + sum_d text_d[i,d] w_tm[d] mod_d[j,d]      r = text_d.w_t + bias, c = mod_d.w_m
//   P1    = softmax_j(mask_mod ? S : -1e30)   P2 = softmax_i(mask_text ? S : -1e30)   (blend, attention.py:94)
//   q     = P2^T text      a = P1 mod      b = P1 q      out = [text, a, text*a, text*b]
//
// Tile engine (per wave): v_mfma_f32_16x16x4_f32; A[i=l&15][k=l>>4], B[k=l>>4][j=l&15],
// C[row=4*(l>>4)+e][col=l&15].  A wave owns 16 rows "n" of one side (the LANE side: n = l&15) and
// streams the other side ("m" rows) through LDS panels:
//   S-type  C[m][n] = sum_d P[m][d] side[n][d]     A = LDS panel row (b128 reads), B = lane-side registers;
//           feature d = 16*s + 4*kg + e is the e-th component of the s-th b128 read of lane group kg.
//   PV-type O[d][n] += sum_m V[m][d] W[m][n]       A = LDS panel column reads, B = the S-type accumulator itself
//           (lane (r,kg) holds W[m=4kg+e][n=r], exactly the B operand of k-group kg): no transpose, no LDS
//           round trip for the probabilities.  The result lands as O[n=r][d=16*dt+4kg+e], the SAME layout as
//           the lane-side registers, so an accumulated gradient (dq) is reused directly as an S-type operand.
// All softmax statistics are lane-local in n (replicated over the 4 k-groups).
#include <math.h>

#include "common.h"

namespace mmb {

constexpr int DT = 13;           // 16-wide feature tiles: D <= 208
constexpr int LDP = 212;         // LDS panel row stride (floats): 848 B, 16-B aligned, odd multiple of 16 B
constexpr int PR = 32;           // m rows per staged panel
constexpr int NW = 4;            // waves per workgroup
constexpr int NTHR = NW * 64;
constexpr float NEG = -1e30f;    // attention.py:94

using side_t = f4[DT];

__device__ __forceinline__ void zero_side(side_t& s) {
#pragma unroll
    for (int i = 0; i < DT; ++i) s[i] = f4{0.f, 0.f, 0.f, 0.f};
}

// lane-side registers: side[s] = src[(n), 16s + 4kg .. +3]  (zero outside N x D), optionally scaled by w[d]
__device__ __forceinline__ void load_side(side_t& side, const float* src_b, int n, int N, int D, int kg, const float* w) {
#pragma unroll
    for (int s = 0; s < DT; ++s) {
        const int d = 16 * s + 4 * kg;
        f4 v = f4{0.f, 0.f, 0.f, 0.f};
        if (n < N && d < D) {
            v = *reinterpret_cast<const f4*>(src_b + (size_t)n * D + d);
            if (w) {
                const f4 ww = *reinterpret_cast<const f4*>(w + d);
                v *= ww;
            }
        }
        side[s] = v;
    }
}

// stage rows [row0,row0+PR) of a (R,D) matrix into an LDS panel [PR][LDP], zero-filled outside
__device__ __forceinline__ void stage_panel(float* panel, const float* src_b, int row0, int R, int D, int tid) {
    for (int i = tid; i < PR * (DT * 4); i += NTHR) {
        const int rr = i / (DT * 4), c = i % (DT * 4);
        f4 v = f4{0.f, 0.f, 0.f, 0.f};
        if (row0 + rr < R && 4 * c < D) v = *reinterpret_cast<const f4*>(src_b + (size_t)(row0 + rr) * D + 4 * c);
        *reinterpret_cast<f4*>(panel + rr * LDP + 4 * c) = v;
    }
}

// two independent S-type chains (m blocks 0 and 1 of the panel) against the same lane-side registers
__device__ __forceinline__ void sprod2(const float* panel, int r, int kg, const side_t& side, f4& c0, f4& c1) {
    const float* p0 = panel + r * LDP + 4 * kg;
    const float* p1 = p0 + 16 * LDP;
#pragma unroll
    for (int s = 0; s < DT; ++s) {
        const f4 a0 = *reinterpret_cast<const f4*>(p0 + 16 * s);
        const f4 a1 = *reinterpret_cast<const f4*>(p1 + 16 * s);
        const f4 b = side[s];
        c0 = mfma16(a0.x, b.x, c0); c1 = mfma16(a1.x, b.x, c1);
        c0 = mfma16(a0.y, b.y, c0); c1 = mfma16(a1.y, b.y, c1);
        c0 = mfma16(a0.z, b.z, c0); c1 = mfma16(a1.z, b.z, c1);
        c0 = mfma16(a0.w, b.w, c0); c1 = mfma16(a1.w, b.w, c1);
    }
}

// PV-type: O[dt] += V[m = mb*16 + 4kg + e][d = 16dt + r] * W[e]   for the m block mb of the panel
__device__ __forceinline__ void pvprod(const float* panel, int mb, int r, int kg, const f4 w, side_t& O) {
    const float* v = panel + (mb * 16 + 4 * kg) * LDP + r;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
        O[dt] = mfma16(v[0 * LDP + 16 * dt], w.x, O[dt]);
        O[dt] = mfma16(v[1 * LDP + 16 * dt], w.y, O[dt]);
        O[dt] = mfma16(v[2 * LDP + 16 * dt], w.z, O[dt]);
        O[dt] = mfma16(v[3 * LDP + 16 * dt], w.w, O[dt]);
    }
}

__device__ __forceinline__ float kg_allsum(float v) {  // over the 4 k-groups (lanes r, r+16, r+32, r+48)
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
}
__device__ __forceinline__ float kg_allmax(float v) {
    v = fmaxf(v, __shfl_xor(v, 16));
    v = fmaxf(v, __shfl_xor(v, 32));
    return v;
}
__device__ __forceinline__ float r_allsum(float v) {  // over the 16 lanes of a k-group
    v += __shfl_xor(v, 1);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 4);
    v += __shfl_xor(v, 8);
    return v;
}
__device__ __forceinline__ float f4sum(const f4 v) { return (v.x + v.y) + (v.z + v.w); }
__device__ __forceinline__ float f4max(const f4 v) { return fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)); }

// ------------------------------------------------------------------------------------------ rank-1 terms
// rterm[b,i] = text_d[b,i].w_t + bias ; cterm[b,j] = mod_d[b,j].w_m        (one wave per row)
__global__ __launch_bounds__(256) void att_rank1_kernel(const float* __restrict__ text_d, const float* __restrict__ mod_d,
                                                        const float* __restrict__ w_t, const float* __restrict__ w_m,
                                                        const float* __restrict__ bias, float* __restrict__ rterm,
                                                        float* __restrict__ cterm, int BT, int BM, int D) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= BT + BM) return;
    const bool is_t = row < BT;
    const float* src = is_t ? text_d + (size_t)row * D : mod_d + (size_t)(row - BT) * D;
    const float* w = is_t ? w_t : w_m;
    float acc = 0.f;
    for (int d = lane * 4; d < D; d += 256) {
        const f4 v = *reinterpret_cast<const f4*>(src + d);
        const f4 ww = *reinterpret_cast<const f4*>(w + d);
        acc += f4sum(v * ww);
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) {
        if (is_t) rterm[row] = acc + bias[0];
        else cterm[row - BT] = acc;
    }
}

// ------------------------------------------------------------------------------------------ forward
struct AttFwdArgs {
    const float* side_src;  // (B,N,D) lane-side S operand (dropped copy), scaled by w_tm on load
    const float* w_tm;      // (D)
    const float* mS;        // (B,R,D) m-side S operand (dropped copy)
    const float* mV0;       // (B,R,D) first  value panel
    const float* mV1;       // (B,R,D) second value panel (row pass only)
    const uint8_t* m_mask;  // (B,R)
    const float* m_term;    // (B,R)
    const float* n_term;    // (B,N)
    float* stat;            // (B,N,2) {max,sum}            (splits == 1)
    float* part_o;          // (B,splits,N,D) unnormalised  (col pass, splits > 1)
    float* part_stat;       // (B,splits,N,2)
    float* q;               // (B,N,D)                      (col pass, splits == 1)
    const float* text;      // (B,N,D)                      (row pass epilogue)
    float* out;             // (B,N,4D)
    float* bsave;           // (B,N,D)
    int N, R, D, splits, rows_per_split;
};

// NV = 1: column pass (lane side = modality rows j, streams text rows i), produces q and the column stats.
// NV = 2: row pass    (lane side = text rows i, streams modality rows j with values [mod | q]), produces out.
template <int NV>
__global__ __launch_bounds__(NTHR) void att_fwd_kernel(const AttFwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, kg = lane >> 4;
    const int b = blockIdx.z, split = blockIdx.y;
    const int N = a.N, R = a.R, D = a.D;
    const int n = (blockIdx.x * NW + wave) * 16 + r;

    const bool sep_s = a.mS != a.mV0;  // dropped copy differs from the clean value panel
    float* pV0 = smem;
    float* pV1 = pV0 + PR * LDP;                        // only touched when NV == 2
    float* pS = sep_s ? (pV0 + NV * PR * LDP) : pV0;
    float* mterm_s = smem + (NV + 1) * PR * LDP;        // [PR]
    int* mcode_s = reinterpret_cast<int*>(mterm_s + PR);  // [PR] 0 = beyond R, 1 = masked, 2 = live

    const float* mS_b = a.mS + (size_t)b * R * D;
    const float* mV0_b = a.mV0 + (size_t)b * R * D;
    const float* mV1_b = NV == 2 ? a.mV1 + (size_t)b * R * D : nullptr;

    side_t side;
    load_side(side, a.side_src + (size_t)b * N * D, n, N, D, kg, a.w_tm);
    const float nterm = n < N ? a.n_term[(size_t)b * N + n] : 0.f;

    side_t O0, O1;
    zero_side(O0);
    zero_side(O1);
    float m_run = -INFINITY, l_run = 0.f;

    const int row_begin = split * a.rows_per_split;
    const int row_end = min(R, row_begin + a.rows_per_split);
    for (int p0 = row_begin; p0 < row_end; p0 += PR) {
        __syncthreads();
        stage_panel(pV0, mV0_b, p0, row_end, D, tid);
        if (NV == 2) stage_panel(pV1, mV1_b, p0, row_end, D, tid);
        if (sep_s) stage_panel(pS, mS_b, p0, row_end, D, tid);
        if (tid < PR) {
            const int m = p0 + tid;
            const bool in = m < row_end;
            mterm_s[tid] = in ? a.m_term[(size_t)b * R + m] : 0.f;
            mcode_s[tid] = in ? (a.m_mask[(size_t)b * R + m] ? 2 : 1) : 0;
        }
        __syncthreads();

        f4 s0 = f4{0.f, 0.f, 0.f, 0.f}, s1 = s0;
        sprod2(pS, r, kg, side, s0, s1);
        f4 v[2] = {s0, s1};
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int ml = mb * 16 + 4 * kg + e;
                const int code = mcode_s[ml];
                const float x = v[mb][e] + mterm_s[ml] + nterm;
                v[mb][e] = code == 2 ? x : (code == 1 ? NEG : -INFINITY);
            }
        const float bmax = kg_allmax(fmaxf(f4max(v[0]), f4max(v[1])));
        const float m_new = fmaxf(m_run, bmax);
        const float alpha = expf(m_run - m_new);
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int e = 0; e < 4; ++e) v[mb][e] = expf(v[mb][e] - m_new);
        l_run = l_run * alpha + f4sum(v[0]) + f4sum(v[1]);
        if (__any(alpha != 1.0f)) {
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                O0[dt] *= alpha;
                if (NV == 2) O1[dt] *= alpha;
            }
        }
        m_run = m_new;
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
            pvprod(pV0, mb, r, kg, v[mb], O0);
            if (NV == 2) pvprod(pV1, mb, r, kg, v[mb], O1);
        }
    }

    const float l = kg_allsum(l_run);
    if (n >= N) return;
    if (NV == 1 && a.splits > 1) {
        float* po = a.part_o + (((size_t)b * a.splits + split) * N + n) * D;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const int d = 16 * dt + 4 * kg;
            if (d < D) *reinterpret_cast<f4*>(po + d) = O0[dt];
        }
        if (kg == 0) {
            float* ps = a.part_stat + (((size_t)b * a.splits + split) * N + n) * 2;
            ps[0] = m_run;
            ps[1] = l;
        }
        return;
    }
    const float inv = 1.0f / l;
    if (kg == 0) {
        a.stat[((size_t)b * N + n) * 2 + 0] = m_run;
        a.stat[((size_t)b * N + n) * 2 + 1] = l;
    }
    if (NV == 1) {
        float* qo = a.q + ((size_t)b * N + n) * D;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const int d = 16 * dt + 4 * kg;
            if (d < D) *reinterpret_cast<f4*>(qo + d) = O0[dt] * inv;
        }
    } else {
        const float* tx = a.text + ((size_t)b * N + n) * D;
        float* oo = a.out + ((size_t)b * N + n) * 4 * D;
        float* bo = a.bsave + ((size_t)b * N + n) * D;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const int d = 16 * dt + 4 * kg;
            if (d < D) {
                const f4 t = *reinterpret_cast<const f4*>(tx + d);
                const f4 av = O0[dt] * inv, bv = O1[dt] * inv;
                *reinterpret_cast<f4*>(oo + d) = t;
                *reinterpret_cast<f4*>(oo + D + d) = av;
                *reinterpret_cast<f4*>(oo + 2 * D + d) = t * av;
                *reinterpret_cast<f4*>(oo + 3 * D + d) = t * bv;
                *reinterpret_cast<f4*>(bo + d) = bv;
            }
        }
    }
}

// merge the per-split partial column softmaxes: q = sum_p O_p e^{m_p-m} / sum_p l_p e^{m_p-m}
__global__ __launch_bounds__(256) void att_combine_kernel(const float* __restrict__ part_o, const float* __restrict__ part_stat,
                                                          float* __restrict__ q, float* __restrict__ stat, int B, int N, int D,
                                                          int splits) {
    const int d4 = D / 4;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)B * N * d4) return;
    const int c = idx % d4;
    const size_t bn = idx / d4;
    const int n = bn % N, b = bn / N;
    float m = -INFINITY;
    for (int p = 0; p < splits; ++p) m = fmaxf(m, part_stat[(((size_t)b * splits + p) * N + n) * 2]);
    float l = 0.f;
    f4 acc = f4{0.f, 0.f, 0.f, 0.f};
    for (int p = 0; p < splits; ++p) {
        const size_t o = ((size_t)b * splits + p) * N + n;
        const float sc = expf(part_stat[o * 2] - m);
        l += part_stat[o * 2 + 1] * sc;
        acc += *reinterpret_cast<const f4*>(part_o + o * D + 4 * c) * sc;
    }
    *reinterpret_cast<f4*>(q + bn * D + 4 * c) = acc * (1.0f / l);
    if (c == 0) {
        stat[bn * 2] = m;
        stat[bn * 2 + 1] = l;
    }
}

// ------------------------------------------------------------------------------------------ backward
// elementwise prologue over text rows (one wave per row):
//   da = g1 + g2*text ; db = g3*text ; delta1 = g1.a + g2.(text*a) + g3.(text*b) ; d_text = g0 + g2*a + g3*b
__global__ __launch_bounds__(256) void att_bwd_pre_kernel(const float* __restrict__ d_out, const float* __restrict__ out,
                                                          const float* __restrict__ text, const float* __restrict__ bsave,
                                                          float* __restrict__ da, float* __restrict__ db,
                                                          float* __restrict__ delta1, float* __restrict__ d_text,
                                                          float* __restrict__ d_w_t, float* __restrict__ d_w_m,
                                                          float* __restrict__ d_w_tm, float* __restrict__ d_bias, int rows, int D) {
    if (blockIdx.x == 0) {  // the parameter gradients are accumulated with atomics by the later kernels
        for (int i = threadIdx.x; i < D; i += 256) d_w_t[i] = d_w_m[i] = d_w_tm[i] = 0.f;
        if (threadIdx.x == 0) d_bias[0] = 0.f;
    }
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* g = d_out + (size_t)row * 4 * D;
    const float* o = out + (size_t)row * 4 * D;
    float acc = 0.f;
    for (int d = lane * 4; d < D; d += 256) {
        const f4 g0 = *reinterpret_cast<const f4*>(g + d), g1 = *reinterpret_cast<const f4*>(g + D + d);
        const f4 g2 = *reinterpret_cast<const f4*>(g + 2 * D + d), g3 = *reinterpret_cast<const f4*>(g + 3 * D + d);
        const f4 o1 = *reinterpret_cast<const f4*>(o + D + d), o2 = *reinterpret_cast<const f4*>(o + 2 * D + d);
        const f4 o3 = *reinterpret_cast<const f4*>(o + 3 * D + d);
        const f4 t = *reinterpret_cast<const f4*>(text + (size_t)row * D + d);
        const f4 bv = *reinterpret_cast<const f4*>(bsave + (size_t)row * D + d);
        *reinterpret_cast<f4*>(da + (size_t)row * D + d) = g1 + g2 * t;
        *reinterpret_cast<f4*>(db + (size_t)row * D + d) = g3 * t;
        *reinterpret_cast<f4*>(d_text + (size_t)row * D + d) = g0 + g2 * o1 + g3 * bv;
        acc += f4sum(g1 * o1 + g2 * o2 + g3 * o3);
    }
#pragma unroll
    for (int o_ = 32; o_ >= 1; o_ >>= 1) acc += __shfl_xor(acc, o_);
    if (lane == 0) delta1[row] = acc;
}

struct AttBwdArgs {
    const float *text, *mod, *text_d, *mod_d;       // (B,T,D) / (B,M,D)
    const uint8_t *text_mask, *mod_mask;            // (B,T) / (B,M)
    const float *w_t, *w_m, *w_tm;
    const float *q, *rterm, *cterm, *row_stat, *col_stat;
    const float *da, *db, *delta1;                   // workspace (B,T,D),(B,T,D),(B,T)
    float *dq, *delta2;                              // workspace (B,M,D),(B,M)
    float *d_mod, *d_mod_d, *d_text, *d_text_d;      // outputs
    float *d_w_t, *d_w_m, *d_w_tm, *d_bias;          // outputs, zeroed by the prologue, accumulated with atomics
    // per-split partial sums of the j-side sweeps, (B,splits,M,D) / (B,splits,M)
    float *p_dq, *p_dmc, *p_dmd1, *p_dmd2, *p_dc1, *p_dc2;
    int T, M, D, splits, rows_per_split;
    int fold;                                        // 1: no dropped copies, d_*_d folded into d_*
};

__device__ __forceinline__ void store_side(float* dst_row, const side_t& v, int D, int kg) {
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
        const int d = 16 * dt + 4 * kg;
        if (d < D) *reinterpret_cast<f4*>(dst_row + d) = v[dt];
    }
}

// j-side sweep 1 (lane side = modality rows j, streams a slice of the text rows i):
//   dq_j += sum_i P1_ij db_i ; dmodc_j += sum_i P1_ij da_i ; dS1 = P1 (dP1 - delta1_i) mask_j
//   dmodd_j += sum_i dS1_ij text_d_i (scaled by w_tm later) ; dc_j += sum_i dS1_ij
__global__ __launch_bounds__(NTHR) void att_bwd_j1_kernel(const AttBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, kg = lane >> 4;
    const int b = blockIdx.z, split = blockIdx.y;
    const int T = a.T, M = a.M, D = a.D;
    const int n = (blockIdx.x * NW + wave) * 16 + r;  // modality row j

    float* pTd = smem;
    float* pDa = pTd + PR * LDP;
    float* pDb = pDa + PR * LDP;
    float* rt_s = pDb + PR * LDP;   // [PR] rterm
    float* rmax_s = rt_s + PR;      // [PR]
    float* rinv_s = rmax_s + PR;    // [PR] 1/rowsum, 0 beyond the slice
    float* dl1_s = rinv_s + PR;     // [PR]

    side_t sideS, sideM, sideQ;
    load_side(sideS, a.mod_d + (size_t)b * M * D, n, M, D, kg, a.w_tm);
    load_side(sideM, a.mod + (size_t)b * M * D, n, M, D, kg, nullptr);
    load_side(sideQ, a.q + (size_t)b * M * D, n, M, D, kg, nullptr);
    const bool nin = n < M;
    const float cterm = nin ? a.cterm[(size_t)b * M + n] : 0.f;
    const bool mm = nin ? a.mod_mask[(size_t)b * M + n] != 0 : false;
    const float mmf = mm ? 1.f : 0.f;

    side_t dq, dmc, dmd;
    zero_side(dq);
    zero_side(dmc);
    zero_side(dmd);
    float dc = 0.f;

    const float* td_b = a.text_d + (size_t)b * T * D;
    const float* da_b = a.da + (size_t)b * T * D;
    const float* db_b = a.db + (size_t)b * T * D;
    const int row_begin = split * a.rows_per_split, row_end = min(T, row_begin + a.rows_per_split);
    for (int p0 = row_begin; p0 < row_end; p0 += PR) {
        __syncthreads();
        stage_panel(pTd, td_b, p0, row_end, D, tid);
        stage_panel(pDa, da_b, p0, row_end, D, tid);
        stage_panel(pDb, db_b, p0, row_end, D, tid);
        if (tid < PR) {
            const int i = p0 + tid;
            const bool in = i < row_end;
            rt_s[tid] = in ? a.rterm[(size_t)b * T + i] : 0.f;
            rmax_s[tid] = in ? a.row_stat[((size_t)b * T + i) * 2] : 0.f;
            rinv_s[tid] = in ? 1.0f / a.row_stat[((size_t)b * T + i) * 2 + 1] : 0.f;
            dl1_s[tid] = in ? a.delta1[(size_t)b * T + i] : 0.f;
        }
        __syncthreads();
        f4 s[2] = {f4{0.f, 0.f, 0.f, 0.f}, f4{0.f, 0.f, 0.f, 0.f}};
        f4 dp[2] = {f4{0.f, 0.f, 0.f, 0.f}, f4{0.f, 0.f, 0.f, 0.f}};
        sprod2(pTd, r, kg, sideS, s[0], s[1]);
        sprod2(pDa, r, kg, sideM, dp[0], dp[1]);
        sprod2(pDb, r, kg, sideQ, dp[0], dp[1]);
        f4 p1[2], ds[2];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int ml = mb * 16 + 4 * kg + e;
                const float x = mm ? s[mb][e] + rt_s[ml] + cterm : NEG;
                const float p = expf(x - rmax_s[ml]) * rinv_s[ml];  // rinv = 0 beyond the slice
                p1[mb][e] = p;
                const float g = p * (dp[mb][e] - dl1_s[ml]) * mmf;
                ds[mb][e] = g;
                dc += g;
            }
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
            pvprod(pDb, mb, r, kg, p1[mb], dq);
            pvprod(pDa, mb, r, kg, p1[mb], dmc);
            pvprod(pTd, mb, r, kg, ds[mb], dmd);
        }
    }
    dc = kg_allsum(dc);
    if (!nin) return;
    const size_t prow = ((size_t)b * a.splits + split) * M + n;
    store_side(a.p_dq + prow * D, dq, D, kg);
    store_side(a.p_dmc + prow * D, dmc, D, kg);
    store_side(a.p_dmd1 + prow * D, dmd, D, kg);
    if (kg == 0) a.p_dc1[prow] = dc;
}

// j-side sweep 2 (needs the complete dq = sum of the sweep-1 partials):
//   dS2 = P2 (dP2 - delta2_j) mask_i, dP2_ij = text_i . dq_j ; dmodd_j += sum_i dS2_ij text_d_i ; dc_j += sum_i dS2_ij
//   split 0 also publishes dq_j and delta2_j = q_j . dq_j for the i-side pass
__global__ __launch_bounds__(NTHR) void att_bwd_j2_kernel(const AttBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, kg = lane >> 4;
    const int b = blockIdx.z, split = blockIdx.y;
    const int T = a.T, M = a.M, D = a.D;
    const int n = (blockIdx.x * NW + wave) * 16 + r;

    const bool sep = a.text_d != a.text;
    float* pT = smem;
    float* pTd = sep ? pT + PR * LDP : pT;
    float* rt_s = smem + 2 * PR * LDP;
    int* code_s = reinterpret_cast<int*>(rt_s + PR);  // 0 beyond slice, 1 masked, 2 live

    side_t sideS, sideDq;
    load_side(sideS, a.mod_d + (size_t)b * M * D, n, M, D, kg, a.w_tm);
    zero_side(sideDq);
    for (int p = 0; p < a.splits; ++p) {
        side_t t;
        load_side(t, a.p_dq + ((size_t)b * a.splits + p) * M * D, n, M, D, kg, nullptr);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) sideDq[dt] += t[dt];
    }
    const bool nin = n < M;
    const float cterm = nin ? a.cterm[(size_t)b * M + n] : 0.f;
    const float cmax = nin ? a.col_stat[((size_t)b * M + n) * 2] : 0.f;
    const float cinv = nin ? 1.0f / a.col_stat[((size_t)b * M + n) * 2 + 1] : 0.f;
    float delta2;
    {
        side_t sq;
        load_side(sq, a.q + (size_t)b * M * D, n, M, D, kg, nullptr);
        float acc = 0.f;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) acc += f4sum(sq[dt] * sideDq[dt]);
        delta2 = kg_allsum(acc);
    }
    if (split == 0 && nin) {
        store_side(a.dq + ((size_t)b * M + n) * D, sideDq, D, kg);
        if (kg == 0) a.delta2[(size_t)b * M + n] = delta2;
    }

    side_t dmd;
    zero_side(dmd);
    float dc = 0.f;
    const float* td_b = a.text_d + (size_t)b * T * D;
    const float* t_b = a.text + (size_t)b * T * D;
    const int row_begin = split * a.rows_per_split, row_end = min(T, row_begin + a.rows_per_split);
    for (int p0 = row_begin; p0 < row_end; p0 += PR) {
        __syncthreads();
        stage_panel(pT, t_b, p0, row_end, D, tid);
        if (sep) stage_panel(pTd, td_b, p0, row_end, D, tid);
        if (tid < PR) {
            const int i = p0 + tid;
            const bool in = i < row_end;
            rt_s[tid] = in ? a.rterm[(size_t)b * T + i] : 0.f;
            code_s[tid] = in ? (a.text_mask[(size_t)b * T + i] ? 2 : 1) : 0;
        }
        __syncthreads();
        f4 s[2] = {f4{0.f, 0.f, 0.f, 0.f}, f4{0.f, 0.f, 0.f, 0.f}};
        f4 dp[2] = {f4{0.f, 0.f, 0.f, 0.f}, f4{0.f, 0.f, 0.f, 0.f}};
        sprod2(pTd, r, kg, sideS, s[0], s[1]);
        sprod2(pT, r, kg, sideDq, dp[0], dp[1]);
        f4 ds[2];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int ml = mb * 16 + 4 * kg + e;
                const int code = code_s[ml];
                const float x = code == 2 ? s[mb][e] + rt_s[ml] + cterm : NEG;
                const float p = code ? expf(x - cmax) * cinv : 0.f;
                const float g = code == 2 ? p * (dp[mb][e] - delta2) : 0.f;
                ds[mb][e] = g;
                dc += g;
            }
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) pvprod(pTd, mb, r, kg, ds[mb], dmd);
    }
    dc = kg_allsum(dc);
    if (!nin) return;
    const size_t prow = ((size_t)b * a.splits + split) * M + n;
    store_side(a.p_dmd2 + prow * D, dmd, D, kg);
    if (kg == 0) a.p_dc2[prow] = dc;
}

// j-side epilogue: one wave per JF_ROWS modality rows, lane = 4 features.  Sums the split partials, writes
//   d_mod_d_j = dc_j w_m + w_tm * dmodd_j ;  d_mod_j = dmodc_j (+ d_mod_d_j when folded)
// and accumulates d_w_m += sum_j dc_j mod_d[j,:] (registers, then LDS across the 4 waves, then one atomic per feature).
constexpr int JF_ROWS = 2;
__global__ __launch_bounds__(256) void att_bwd_jfin_kernel(const AttBwdArgs a, int B) {
    __shared__ f4 wred[4][64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int chunk = blockIdx.x * 4 + wave;
    const int M = a.M, D = a.D, S = a.splits;
    const int rows = B * M;
    const int d = lane * 4;
    f4 wacc = f4{0.f, 0.f, 0.f, 0.f};
    const bool din = d < D;
    const f4 wm = din ? *reinterpret_cast<const f4*>(a.w_m + d) : f4{0.f, 0.f, 0.f, 0.f};
    const f4 wtm = din ? *reinterpret_cast<const f4*>(a.w_tm + d) : f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int rr = 0; rr < JF_ROWS; ++rr) {
        const int row = chunk * JF_ROWS + rr;
        if (row < rows) {
            const int b = row / M, n = row % M;
            float dc = 0.f;
            f4 c = f4{0.f, 0.f, 0.f, 0.f}, dd = c;
            for (int p = 0; p < S; ++p) {
                const size_t prow = ((size_t)b * S + p) * M + n;
                dc += a.p_dc1[prow] + a.p_dc2[prow];
                if (din) {
                    c += *reinterpret_cast<const f4*>(a.p_dmc + prow * D + d);
                    dd += *reinterpret_cast<const f4*>(a.p_dmd1 + prow * D + d);
                    dd += *reinterpret_cast<const f4*>(a.p_dmd2 + prow * D + d);
                }
            }
            if (din) {
                const f4 gd = wm * dc + wtm * dd;
                if (a.fold) {
                    *reinterpret_cast<f4*>(a.d_mod + (size_t)row * D + d) = c + gd;
                } else {
                    *reinterpret_cast<f4*>(a.d_mod + (size_t)row * D + d) = c;
                    *reinterpret_cast<f4*>(a.d_mod_d + (size_t)row * D + d) = gd;
                }
                wacc += *reinterpret_cast<const f4*>(a.mod_d + (size_t)row * D + d) * dc;
            }
        }
    }
    wred[wave][lane] = wacc;
    __syncthreads();
    if (wave == 0 && din) {
        const f4 t = wred[0][lane] + wred[1][lane] + wred[2][lane] + wred[3][lane];
        atomicAdd(a.d_w_m + d + 0, t.x);
        atomicAdd(a.d_w_m + d + 1, t.y);
        atomicAdd(a.d_w_m + d + 2, t.z);
        atomicAdd(a.d_w_m + d + 3, t.w);
    }
}

// i-side pass (lane side = text rows i, streams all modality rows j):
//   dS = P1 (dP1 - delta1_i) mask_j + P2 (dP2 - delta2_j) mask_i
//   d_text_i += sum_j P2_ij dq_j ; dX_i = sum_j dS_ij mod_d_j ; dr_i = sum_j dS_ij
//   d_text_d_i = dr_i w_t + w_tm * dX_i ; d_w_t += dr_i text_d_i ; d_w_tm += dX_i * text_d_i ; d_bias += dr_i
constexpr int PI_STRIDE = 2 * DT * 16 + 16;  // per-wave partial: [d_w_t 208 | d_w_tm 208 | d_bias 1 ...]
__global__ __launch_bounds__(NTHR) void att_bwd_i_kernel(const AttBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, kg = lane >> 4;
    const int b = blockIdx.z;
    const int T = a.T, M = a.M, D = a.D;
    const int n = (blockIdx.x * NW + wave) * 16 + r;  // text row i

    const bool sep = a.mod_d != a.mod;
    float* pM = smem;
    float* pQ = pM + PR * LDP;
    float* pDq = pQ + PR * LDP;
    float* pMd = sep ? pDq + PR * LDP : pM;
    float* ct_s = smem + 4 * PR * LDP;  // [PR] cterm
    float* cmax_s = ct_s + PR;
    float* cinv_s = cmax_s + PR;        // 0 beyond M
    float* dl2_s = cinv_s + PR;
    float* mmf_s = dl2_s + PR;          // modality mask as float, -1 beyond M

    side_t sideS, sideDa, sideDb, sideT;
    load_side(sideS, a.text_d + (size_t)b * T * D, n, T, D, kg, a.w_tm);
    load_side(sideDa, a.da + (size_t)b * T * D, n, T, D, kg, nullptr);
    load_side(sideDb, a.db + (size_t)b * T * D, n, T, D, kg, nullptr);
    load_side(sideT, a.text + (size_t)b * T * D, n, T, D, kg, nullptr);
    const bool nin = n < T;
    const float rterm = nin ? a.rterm[(size_t)b * T + n] : 0.f;
    const float rmax = nin ? a.row_stat[((size_t)b * T + n) * 2] : 0.f;
    const float rinv = nin ? 1.0f / a.row_stat[((size_t)b * T + n) * 2 + 1] : 0.f;
    const float dl1 = nin ? a.delta1[(size_t)b * T + n] : 0.f;
    const bool tm = nin ? a.text_mask[(size_t)b * T + n] != 0 : false;
    const float tmf = tm ? 1.f : 0.f;

    side_t dtx, dX;
    zero_side(dtx);
    zero_side(dX);
    float dr = 0.f;
    const float* m_b = a.mod + (size_t)b * M * D;
    const float* md_b = a.mod_d + (size_t)b * M * D;
    const float* q_b = a.q + (size_t)b * M * D;
    const float* dq_b = a.dq + (size_t)b * M * D;
    for (int p0 = 0; p0 < M; p0 += PR) {
        __syncthreads();
        stage_panel(pM, m_b, p0, M, D, tid);
        stage_panel(pQ, q_b, p0, M, D, tid);
        stage_panel(pDq, dq_b, p0, M, D, tid);
        if (sep) stage_panel(pMd, md_b, p0, M, D, tid);
        if (tid < PR) {
            const int j = p0 + tid;
            const bool in = j < M;
            ct_s[tid] = in ? a.cterm[(size_t)b * M + j] : 0.f;
            cmax_s[tid] = in ? a.col_stat[((size_t)b * M + j) * 2] : 0.f;
            cinv_s[tid] = in ? 1.0f / a.col_stat[((size_t)b * M + j) * 2 + 1] : 0.f;
            dl2_s[tid] = in ? a.delta2[(size_t)b * M + j] : 0.f;
            mmf_s[tid] = in ? (a.mod_mask[(size_t)b * M + j] ? 1.f : 0.f) : -1.f;
        }
        __syncthreads();
        f4 s[2] = {f4{0.f, 0.f, 0.f, 0.f}, f4{0.f, 0.f, 0.f, 0.f}};
        f4 dp1[2] = {f4{0.f, 0.f, 0.f, 0.f}, f4{0.f, 0.f, 0.f, 0.f}};
        f4 dp2[2] = {f4{0.f, 0.f, 0.f, 0.f}, f4{0.f, 0.f, 0.f, 0.f}};
        sprod2(pMd, r, kg, sideS, s[0], s[1]);
        sprod2(pM, r, kg, sideDa, dp1[0], dp1[1]);
        sprod2(pQ, r, kg, sideDb, dp1[0], dp1[1]);
        sprod2(pDq, r, kg, sideT, dp2[0], dp2[1]);
        f4 p2[2], ds[2];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int ml = mb * 16 + 4 * kg + e;
                const float mf = mmf_s[ml];
                const float x = s[mb][e] + rterm + ct_s[ml];
                const float P1 = mf >= 0.f ? expf((mf > 0.f ? x : NEG) - rmax) * rinv : 0.f;
                const float P2 = mf >= 0.f ? expf((tm ? x : NEG) - cmax_s[ml]) * cinv_s[ml] : 0.f;
                const float g1 = mf > 0.f ? P1 * (dp1[mb][e] - dl1) : 0.f;
                const float g2 = P2 * (dp2[mb][e] - dl2_s[ml]) * tmf;
                p2[mb][e] = P2;
                ds[mb][e] = g1 + g2;
                dr += g1 + g2;
            }
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
            pvprod(pDq, mb, r, kg, p2[mb], dtx);
            pvprod(pMd, mb, r, kg, ds[mb], dX);
        }
    }
    dr = kg_allsum(dr);
    // ---- epilogue: gradients of this text row; parameter-gradient partials reduced over the workgroup in LDS
    __syncthreads();  // panels are dead: reuse their memory
    float* part = smem + wave * PI_STRIDE;
    const float* td_row = a.text_d + ((size_t)b * T + n) * D;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
        const int d = 16 * dt + 4 * kg;
        f4 pt = f4{0.f, 0.f, 0.f, 0.f}, ptm = pt;
        if (nin && d < D) {
            const f4 wt = *reinterpret_cast<const f4*>(a.w_t + d), wtm = *reinterpret_cast<const f4*>(a.w_tm + d);
            const f4 td = *reinterpret_cast<const f4*>(td_row + d);
            const f4 gd = wt * dr + wtm * dX[dt];
            float* dtp = a.d_text + ((size_t)b * T + n) * D + d;
            const f4 prev = *reinterpret_cast<const f4*>(dtp);  // g0 + g2*a + g3*b from the prologue
            if (a.fold) {
                *reinterpret_cast<f4*>(dtp) = prev + dtx[dt] + gd;
            } else {
                *reinterpret_cast<f4*>(dtp) = prev + dtx[dt];
                *reinterpret_cast<f4*>(a.d_text_d + ((size_t)b * T + n) * D + d) = gd;
            }
            pt = td * dr;
            ptm = td * dX[dt];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float s1 = r_allsum(pt[e]), s2 = r_allsum(ptm[e]);
            if (r == 0) {
                part[16 * dt + 4 * kg + e] = s1;
                part[DT * 16 + 16 * dt + 4 * kg + e] = s2;
            }
        }
    }
    const float sb = r_allsum(nin ? dr : 0.f);
    if (lane == 0) part[2 * DT * 16] = sb;
    __syncthreads();
    for (int i = tid; i < 2 * DT * 16 + 1; i += NTHR) {
        float acc = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) acc += smem[w * PI_STRIDE + i];
        if (i < DT * 16) {
            if (i < D) atomicAdd(a.d_w_t + i, acc);
        } else if (i < 2 * DT * 16) {
            if (i - DT * 16 < D) atomicAdd(a.d_w_tm + (i - DT * 16), acc);
        } else {
            atomicAdd(a.d_bias, acc);
        }
    }
}

// ------------------------------------------------------------------------------------------ host side
static int pick_splits(int B, int N, int R) {
    const int waves = B * ((N + 15) / 16);
    int s = (1536 + waves - 1) / waves;  // aim at ~1.5 waves per SIMD (1024 SIMDs)
    const int smax = (R + PR - 1) / PR;
    if (s > smax) s = smax;
    return s < 1 ? 1 : s;
}
static int rows_per_split(int R, int splits) {
    int rp = (R + splits - 1) / splits;
    return (rp + PR - 1) / PR * PR;
}

struct BwdWs {
    size_t da, db, delta1, dq, delta2, p_dq, p_dmc, p_dmd1, p_dmd2, p_dc1, p_dc2, total;
    int splits;
};
static BwdWs bwd_layout(int B, int T, int M, int D) {
    BwdWs w{};
    w.splits = pick_splits(B, M, T);
    const size_t S = w.splits;
    size_t o = 0;
    auto take = [&](size_t nfloat) { size_t at = o; o += (nfloat + 3) / 4 * 4; return at; };
    w.da = take((size_t)B * T * D);
    w.db = take((size_t)B * T * D);
    w.delta1 = take((size_t)B * T);
    w.dq = take((size_t)B * M * D);
    w.delta2 = take((size_t)B * M);
    w.p_dq = take(S * B * M * D);
    w.p_dmc = take(S * B * M * D);
    w.p_dmd1 = take(S * B * M * D);
    w.p_dmd2 = take(S * B * M * D);
    w.p_dc1 = take(S * B * M);
    w.p_dc2 = take(S * B * M);
    w.total = o;
    return w;
}

template <typename K>
static int allow_lds(K kernel, size_t bytes) {
    MMB_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return MMB_OK;
}

}  // namespace mmb

using namespace mmb;

static int check_att_dims(int B, int T, int M, int D) {
    MMB_REQUIRE(B >= 1 && T >= 1 && M >= 1, "bidaf: bad sizes B=%d T=%d M=%d", B, T, M);
    MMB_REQUIRE(D >= 4 && D % 4 == 0 && D <= MMB_ATT_MAX_D, "bidaf: D=%d must be a multiple of 4 and <= %d", D, MMB_ATT_MAX_D);
    return MMB_OK;
}

extern "C" int mmb_bidaf_fwd(const float* text, const float* mod, const uint8_t* text_mask, const uint8_t* mod_mask,
                             const float* text_d, const float* mod_d, const float* w_t, const float* w_m,
                             const float* w_tm, const float* bias, float* out, float* q, float* bsave, float* rterm,
                             float* cterm, float* row_stat, float* col_stat, int B, int T, int M, int D, int device,
                             void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (int rc = check_att_dims(B, T, M, D)) return rc;
    MMB_REQUIRE(text && mod && text_mask && mod_mask && w_t && w_m && w_tm && bias && out && q && bsave && rterm &&
                    cterm && row_stat && col_stat, "mmb_bidaf_fwd: null pointer");
    MMB_HIP(hipSetDevice(device));
    if (!text_d) text_d = text;
    if (!mod_d) mod_d = mod;

    { ProfScope ps_(MMB_K_ATT_RANK1, stream); hipLaunchKernelGGL(att_rank1_kernel, dim3((B * T + B * M + 3) / 4), dim3(256), 0, stream, text_d, mod_d, w_t, w_m, bias,
                       rterm, cterm, B * T, B * M, D); }
    MMB_HIP(hipGetLastError());

    // ---- column pass: lane side = modality rows, streams text.  (`out` is used as scratch for the split
    //      partials: it is (B,T,4D) and is only written by the row pass afterwards.)
    {
        AttFwdArgs a{};
        a.side_src = mod_d; a.w_tm = w_tm; a.mS = text_d; a.mV0 = text; a.mV1 = nullptr;
        a.m_mask = text_mask; a.m_term = rterm; a.n_term = cterm; a.stat = col_stat; a.q = q;
        a.N = M; a.R = T; a.D = D;
        int splits = pick_splits(B, M, T);
        while (splits > 1 && (size_t)splits * M * (D + 2) > (size_t)T * 4 * D) --splits;
        a.splits = splits;
        a.rows_per_split = rows_per_split(T, splits);
        a.part_o = out;
        a.part_stat = out + (size_t)B * splits * M * D;
        const size_t lds = ((size_t)(1 + 1) * PR * LDP + 2 * PR) * sizeof(float);
        if (int rc = allow_lds(att_fwd_kernel<1>, lds)) return rc;
        dim3 grid((M + 16 * NW - 1) / (16 * NW), splits, B);
        { ProfScope ps_(MMB_K_ATT_COL, stream); hipLaunchKernelGGL(att_fwd_kernel<1>, grid, dim3(NTHR), lds, stream, a); }
        MMB_HIP(hipGetLastError());
        if (splits > 1) {
            const size_t nthr = (size_t)B * M * (D / 4);
            { ProfScope ps_(MMB_K_ATT_COMBINE, stream); hipLaunchKernelGGL(att_combine_kernel, dim3((nthr + 255) / 256), dim3(256), 0, stream, a.part_o, a.part_stat, q,
                               col_stat, B, M, D, splits); }
            MMB_HIP(hipGetLastError());
        }
    }
    // ---- row pass: lane side = text rows, streams [mod | q]
    {
        AttFwdArgs a{};
        a.side_src = text_d; a.w_tm = w_tm; a.mS = mod_d; a.mV0 = mod; a.mV1 = q;
        a.m_mask = mod_mask; a.m_term = cterm; a.n_term = rterm; a.stat = row_stat;
        a.text = text; a.out = out; a.bsave = bsave;
        a.N = T; a.R = M; a.D = D; a.splits = 1; a.rows_per_split = rows_per_split(M, 1);
        const size_t lds = ((size_t)(2 + 1) * PR * LDP + 2 * PR) * sizeof(float);
        if (int rc = allow_lds(att_fwd_kernel<2>, lds)) return rc;
        dim3 grid((T + 16 * NW - 1) / (16 * NW), 1, B);
        { ProfScope ps_(MMB_K_ATT_ROW, stream); hipLaunchKernelGGL(att_fwd_kernel<2>, grid, dim3(NTHR), lds, stream, a); }
        MMB_HIP(hipGetLastError());
    }
    return MMB_OK;
}

extern "C" size_t mmb_bidaf_bwd_workspace_bytes(int B, int T, int M, int D) {
    if (B < 1 || T < 1 || M < 1 || D < 4) return 0;
    return bwd_layout(B, T, M, D).total * sizeof(float);
}

extern "C" int mmb_bidaf_bwd(const float* d_out, const float* out, const float* text, const float* mod,
                             const uint8_t* text_mask, const uint8_t* mod_mask, const float* text_d, const float* mod_d,
                             const float* w_t, const float* w_m, const float* w_tm, const float* q, const float* bsave,
                             const float* rterm, const float* cterm, const float* row_stat, const float* col_stat,
                             float* d_text, float* d_mod, float* d_text_d, float* d_mod_d, float* d_w_t, float* d_w_m,
                             float* d_w_tm, float* d_bias, float* workspace, size_t workspace_bytes, int B, int T, int M,
                             int D, int device, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (int rc = check_att_dims(B, T, M, D)) return rc;
    MMB_REQUIRE(d_out && out && text && mod && text_mask && mod_mask && w_t && w_m && w_tm && q && bsave && rterm && cterm &&
                    row_stat && col_stat && d_text && d_mod && d_w_t && d_w_m && d_w_tm && d_bias && workspace,
                "mmb_bidaf_bwd: null pointer");
    const bool drop_t = text_d != nullptr, drop_m = mod_d != nullptr;
    MMB_REQUIRE(drop_t == drop_m, "mmb_bidaf_bwd: text_d and mod_d must both be given or both be NULL");
    MMB_REQUIRE(drop_t ? (d_text_d && d_mod_d) : (!d_text_d && !d_mod_d),
                "mmb_bidaf_bwd: d_text_d/d_mod_d must be given exactly when text_d/mod_d are");
    const BwdWs L = bwd_layout(B, T, M, D);
    MMB_REQUIRE(workspace_bytes >= L.total * sizeof(float), "mmb_bidaf_bwd: workspace too small (%zu < %zu)", workspace_bytes,
                L.total * sizeof(float));
    MMB_HIP(hipSetDevice(device));

    AttBwdArgs a{};
    a.text = text; a.mod = mod; a.text_d = drop_t ? text_d : text; a.mod_d = drop_m ? mod_d : mod;
    a.text_mask = text_mask; a.mod_mask = mod_mask; a.w_t = w_t; a.w_m = w_m; a.w_tm = w_tm;
    a.q = q; a.rterm = rterm; a.cterm = cterm; a.row_stat = row_stat; a.col_stat = col_stat;
    a.da = workspace + L.da; a.db = workspace + L.db; a.delta1 = workspace + L.delta1;
    a.dq = workspace + L.dq; a.delta2 = workspace + L.delta2;
    a.p_dq = workspace + L.p_dq; a.p_dmc = workspace + L.p_dmc; a.p_dmd1 = workspace + L.p_dmd1;
    a.p_dmd2 = workspace + L.p_dmd2; a.p_dc1 = workspace + L.p_dc1; a.p_dc2 = workspace + L.p_dc2;
    a.d_mod = d_mod; a.d_mod_d = d_mod_d; a.d_text = d_text; a.d_text_d = d_text_d;
    a.d_w_t = d_w_t; a.d_w_m = d_w_m; a.d_w_tm = d_w_tm; a.d_bias = d_bias;
    a.T = T; a.M = M; a.D = D; a.fold = drop_t ? 0 : 1;
    a.splits = L.splits;
    a.rows_per_split = rows_per_split(T, a.splits);

    { ProfScope ps_(MMB_K_ATT_BWD_PRE, stream); hipLaunchKernelGGL(att_bwd_pre_kernel, dim3((B * T + 3) / 4), dim3(256), 0, stream, d_out, out, text, bsave,
                       workspace + L.da, workspace + L.db, workspace + L.delta1, d_text, d_w_t, d_w_m, d_w_tm, d_bias, B * T, D); }
    MMB_HIP(hipGetLastError());
    {
        const size_t lds = ((size_t)3 * PR * LDP + 4 * PR) * sizeof(float);
        if (int rc = allow_lds(att_bwd_j1_kernel, lds)) return rc;
        dim3 grid((M + 16 * NW - 1) / (16 * NW), a.splits, B);
        { ProfScope ps_(MMB_K_ATT_BWD_J1, stream); hipLaunchKernelGGL(att_bwd_j1_kernel, grid, dim3(NTHR), lds, stream, a); }
        MMB_HIP(hipGetLastError());
    }
    {
        const size_t lds = ((size_t)2 * PR * LDP + 2 * PR) * sizeof(float);
        if (int rc = allow_lds(att_bwd_j2_kernel, lds)) return rc;
        dim3 grid((M + 16 * NW - 1) / (16 * NW), a.splits, B);
        { ProfScope ps_(MMB_K_ATT_BWD_J2, stream); hipLaunchKernelGGL(att_bwd_j2_kernel, grid, dim3(NTHR), lds, stream, a); }
        MMB_HIP(hipGetLastError());
    }
    {
        const int chunks = (B * M + JF_ROWS - 1) / JF_ROWS;
        { ProfScope ps_(MMB_K_ATT_BWD_JFIN, stream); hipLaunchKernelGGL(att_bwd_jfin_kernel, dim3((chunks + 3) / 4), dim3(256), 0, stream, a, B); }
        MMB_HIP(hipGetLastError());
    }
    {
        const size_t lds = ((size_t)4 * PR * LDP + 5 * PR) * sizeof(float);
        if (int rc = allow_lds(att_bwd_i_kernel, lds)) return rc;
        dim3 grid((T + 16 * NW - 1) / (16 * NW), 1, B);
        { ProfScope ps_(MMB_K_ATT_BWD_I, stream); hipLaunchKernelGGL(att_bwd_i_kernel, grid, dim3(NTHR), lds, stream, a); }
        MMB_HIP(hipGetLastError());
    }
    return MMB_OK;
}
