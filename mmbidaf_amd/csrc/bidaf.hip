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
// Workgroup geometry is a template parameter of the tile kernels: NWv waves (16 lane-side rows each) share
// PRv-row panels of the streamed side.  Small workgroups + short panels trade panel re-reads (served by the XCD's
// L2, see decode_block) for more resident workgroups per CU, which is what hides staging and barrier latency here.
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

// Stage rows [row0,row0+PR) of a (R,D) matrix into an LDS panel [PR][LDP] by LDS-DMA (global_load_lds_dwordx4:
// 16 B per lane straight into LDS, no VGPR round trip, no per-chunk VALU work in the panel loop).  The LDS image is
// lane-linear in 16-B chunks: chunk c = tid + k*NTHR is row c / 53, columns 4*(c % 53)...; the per-lane SOURCE address
// is free, so rows beyond R and columns beyond D (incl. the pad chunk) are simply clamped onto valid, finite data:
// every consumer ignores them (rows via its softmax code / zero weights, columns via zero lane-side registers and
// d < D store guards).
template <int NTHR, int PR>
__device__ __forceinline__ void stage_panel(float* panel, const float* src_b, int row0, int R, int D, int tid) {
    constexpr int CPR = LDP / 4;                  // 16-B chunks per LDS row (53)
    constexpr int NCH = PR * CPR;                 // chunks per panel
    constexpr int NIT = (NCH + NTHR - 1) / NTHR;  // wave-instructions per wave
    float* wave_dst = panel + (tid & ~63) * 4;    // wave-uniform: this wave's 1-KiB slot of each NTHR*16-B stripe
#pragma unroll 1   // keep the address math inside the loop: hoisted, it costs ~50 VGPRs in the register-bound kernels
    for (int k = 0; k < NIT; ++k) {
        const int c = tid + k * NTHR;
        if (NCH % NTHR == 0 || c < NCH) {
            const int row = c / CPR, col = min(4 * (c - row * CPR), D - 4);
            const float* src = src_b + (size_t)min(row0 + row, R - 1) * D + col;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(wave_dst + k * NTHR * 4), 16, 0, 0);
        }
    }
}

// NB independent S-type chains (the NB 16-row m blocks of the panel) against the same lane-side registers
template <int NB>
__device__ __forceinline__ void sprodN(const float* panel, int r, int kg, const side_t& side, f4 (&c)[NB]) {
    const float* p0 = panel + r * LDP + 4 * kg;
#pragma unroll
    for (int s = 0; s < DT; ++s) {
        const f4 b = side[s];
        f4 a[NB];
#pragma unroll
        for (int q = 0; q < NB; ++q) a[q] = *reinterpret_cast<const f4*>(p0 + q * 16 * LDP + 16 * s);
#pragma unroll
        for (int q = 0; q < NB; ++q) c[q] = mfma16(a[q].x, b.x, c[q]);
#pragma unroll
        for (int q = 0; q < NB; ++q) c[q] = mfma16(a[q].y, b.y, c[q]);
#pragma unroll
        for (int q = 0; q < NB; ++q) c[q] = mfma16(a[q].z, b.z, c[q]);
#pragma unroll
        for (int q = 0; q < NB; ++q) c[q] = mfma16(a[q].w, b.w, c[q]);
    }
}

// block index -> (lane-side tile, split, sample).  Blocks are dealt round-robin over the 8 XCDs (id % 8 labels the
// XCD group), so all blocks of one sample are given ids with equal id % 8: the panels they all stream then stay in
// that XCD's L2.  Purely a speed choice; any mapping is correct.
__device__ __forceinline__ void decode_block(int tiles, int splits, int B, int& tile, int& split, int& b) {
    const int id = blockIdx.x;
    int slot;
    if (B % 8 == 0) {
        const int xcd = id & 7;
        slot = id >> 3;
        b = xcd + 8 * (slot % (B / 8));
        slot /= (B / 8);
    } else {
        b = id % B;
        slot = id / B;
    }
    tile = slot % tiles;
    split = slot / tiles;
}

// PV-type: O[dt] += V[m = mb*16 + 4kg + e][d = 16dt + r] * W[e]   for the m block mb of the panel.
// Software-pipelined by hand: the 13 LDS reads of row e+1 are issued before the 13 MFMAs of row e (at one wave per
// SIMD hipcc otherwise keeps only 1-2 reads in flight and every ~100-cycle LDS latency lands on the MFMA stream).
__device__ __forceinline__ void pvprod(const float* panel, int mb, int r, int kg, const f4 w, side_t& O) {
    const float* v = panel + (mb * 16 + 4 * kg) * LDP + r;
    float cur[DT], nxt[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) cur[dt] = v[16 * dt];
    __builtin_amdgcn_sched_group_barrier(0x100, DT, 0);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (e < 3) {
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) nxt[dt] = v[(e + 1) * LDP + 16 * dt];
        }
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) O[dt] = mfma16(cur[dt], w[e], O[dt]);
        if (e < 3) {
            // one MFMA, then one read of the next row, ... : reads ride under the MFMAs
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) cur[dt] = nxt[dt];
        } else {
            __builtin_amdgcn_sched_group_barrier(0x008, DT, 0);
        }
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
    int N, R, D, B, splits, rows_per_split;
    int dbg;   // timing-only ablations (MMB_ATT_DBG, never set by the product path): 1 = skip staging, 2 = skip S, 4 = skip PV, 8 = skip the epilogue
};

// NV = 1: column pass = att_col_kernel (lane side = modality rows j, streams text rows i), produces q and the column stats.
// NV = 2: row pass = att_row_kernel    (lane side = text rows i, streams modality rows j with values [mod | q]), produces out.
// DB: two LDS stages -- the LDS-DMA of panel p+1 is in flight under the MFMAs of panel p (one barrier per panel)
template <int NV, int NW, int PR, bool DB>
__device__ __forceinline__ void att_fwd_body(const AttFwdArgs& a, float* smem) {
    constexpr int NTHR = NW * 64, NB = PR / 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, kg = lane >> 4;
    const int N = a.N, R = a.R, D = a.D;
    int tile, split, b;
    decode_block((N + 16 * NW - 1) / (16 * NW), a.splits, a.B, tile, split, b);
    const int n = (tile * NW + wave) * 16 + r;

    const bool sep_s = a.mS != a.mV0;  // dropped copy differs from the clean value panel
    constexpr int STAGE_F = (NV + 1) * PR * LDP;   // floats per LDS stage (panels)

    const float* mS_b = a.mS + (size_t)b * R * D;
    const float* mV0_b = a.mV0 + (size_t)b * R * D;
    const float* mV1_b = NV == 2 ? a.mV1 + (size_t)b * R * D : nullptr;

    side_t side;
    load_side(side, a.side_src + (size_t)b * N * D, n, N, D, kg, a.w_tm);
    const float nterm = n < N ? a.n_term[(size_t)b * N + n] : 0.f;

    if (NV == 2) {
        // the first quarter of `out` is a verbatim copy of text (attention.py:52): written here, whole rows per
        // wave-instruction, so that these stores overlap the main loop instead of joining the epilogue burst
        const float* tx = a.text + (size_t)b * N * D;
        float* oo = a.out + (size_t)b * N * 4 * D;
        for (int rr = wave; rr < 16 * NW; rr += NW) {
            const int gn = tile * NW * 16 + rr;
            if (gn < N && 4 * lane < D)
                *reinterpret_cast<f4*>(oo + (size_t)gn * 4 * D + 4 * lane) = *reinterpret_cast<const f4*>(tx + (size_t)gn * D + 4 * lane);
        }
    }
    side_t O0, O1;
    zero_side(O0);
    zero_side(O1);
    float m_run = -INFINITY, l_run = 0.f;

    const int row_begin = split * a.rows_per_split;
    const int row_end = min(R, row_begin + a.rows_per_split);
    auto stage = [&](float* base, int p0) {
        float* pV0 = base;
        stage_panel<NTHR, PR>(pV0, mV0_b, p0, row_end, D, tid);
        if (NV == 2) stage_panel<NTHR, PR>(pV0 + PR * LDP, mV1_b, p0, row_end, D, tid);
        if (sep_s) stage_panel<NTHR, PR>(pV0 + NV * PR * LDP, mS_b, p0, row_end, D, tid);
    };
    // per-row scalars of the streamed side for the WHOLE split, loaded once (a per-panel global load in front of the
    // panel barrier costs a full memory round trip per panel)
    float* mterm_all = smem + (DB ? 2 : 1) * STAGE_F;                   // [rows_per_split]
    int* mcode_all = reinterpret_cast<int*>(mterm_all + a.rows_per_split);   // 0 = beyond R, 1 = masked, 2 = live
    for (int i = tid; i < a.rows_per_split; i += NTHR) {
        const int m = row_begin + i;
        const bool in = m < row_end;
        mterm_all[i] = in ? a.m_term[(size_t)b * R + m] : 0.f;
        mcode_all[i] = in ? (a.m_mask[(size_t)b * R + m] ? 2 : 1) : 0;
    }
    int cur = 0;
    if (DB) {
        if (row_begin < row_end) stage(smem, row_begin);
        __syncthreads();
    }
    for (int p0 = row_begin; p0 < row_end; p0 += PR) {
        float* base = smem + (DB ? cur * STAGE_F : 0);
        if (DB) {
            if (p0 + PR < row_end) stage(smem + (cur ^ 1) * STAGE_F, p0 + PR);
        } else {
            __syncthreads();
            if (!(a.dbg & 1) || p0 == row_begin) stage(base, p0);
            __syncthreads();
        }
        const float* pV0 = base;
        const float* pV1 = pV0 + PR * LDP;                       // only touched when NV == 2
        const float* pS = sep_s ? (pV0 + NV * PR * LDP) : pV0;
        const float* mterm_s = mterm_all + (p0 - row_begin);
        const int* mcode_s = mcode_all + (p0 - row_begin);

        f4 v[NB];
#pragma unroll
        for (int q = 0; q < NB; ++q) v[q] = f4{0.f, 0.f, 0.f, 0.f};
        if (!(a.dbg & 2)) sprodN<NB>(pS, r, kg, side, v);
        float bmax = -INFINITY;
#pragma unroll
        for (int mb = 0; mb < NB; ++mb)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int ml = mb * 16 + 4 * kg + e;
                const int code = mcode_s[ml];
                const float x = v[mb][e] + mterm_s[ml] + nterm;
                v[mb][e] = code == 2 ? x : (code == 1 ? NEG : -INFINITY);
                bmax = fmaxf(bmax, v[mb][e]);
            }
        bmax = kg_allmax(bmax);
        const float m_new = fmaxf(m_run, bmax);
        const float alpha = __expf(m_run - m_new);
        float psum = 0.f;
#pragma unroll
        for (int mb = 0; mb < NB; ++mb)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[mb][e] = __expf(v[mb][e] - m_new);
                psum += v[mb][e];
            }
        l_run = l_run * alpha + psum;
        if (__any(alpha != 1.0f)) {
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                O0[dt] *= alpha;
                if (NV == 2) O1[dt] *= alpha;
            }
        }
        m_run = m_new;
#pragma unroll
        for (int mb = 0; mb < NB; ++mb) {
            if (a.dbg & 4) continue;
            pvprod(pV0, mb, r, kg, v[mb], O0);
            if (NV == 2) pvprod(pV1, mb, r, kg, v[mb], O1);
        }
        if (DB) {
            __syncthreads();   // retires the DMA of the next stage and frees this one
            cur ^= 1;
        }
    }

    // ---- epilogue.  The accumulators hold 16 rows x 64-B pieces per store instruction; written directly that is 16
    // partial cache lines per instruction (measured: half of this kernel's time).  Instead each wave parks its tile in
    // LDS (the panels are dead) and the workgroup writes whole rows: one row per wave-instruction, lane = 16-B chunk.
    const float l = kg_allsum(l_run);
    const bool partial = NV == 1 && a.splits > 1;
    const float inv = partial ? 1.0f : 1.0f / l;
    if (n < N && kg == 0) {
        float* st = partial ? a.part_stat + (((size_t)b * a.splits + split) * N + n) * 2 : a.stat + ((size_t)b * N + n) * 2;
        st[0] = m_run;
        st[1] = l;
    }
    if (a.dbg & 8) return;   // timing-only: no epilogue
    float* et = smem;                                   // [16*NW][LDP]
    const int row0 = tile * NW * 16;                    // first lane-side row of this workgroup
    const int c4 = lane;                                // this lane's 16-B chunk of a row
    auto park = [&](const side_t& O) {
        __syncthreads();
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
            *reinterpret_cast<f4*>(et + (wave * 16 + r) * LDP + 16 * dt + 4 * kg) = O[dt] * inv;
        __syncthreads();
    };
    park(O0);
    if (NV == 1) {
        float* dst = partial ? a.part_o + ((size_t)b * a.splits + split) * N * D : a.q + (size_t)b * N * D;
        for (int rr = wave; rr < 16 * NW; rr += NW) {
            const int gn = row0 + rr;
            if (gn < N && 4 * c4 < D) *reinterpret_cast<f4*>(dst + (size_t)gn * D + 4 * c4) = *reinterpret_cast<const f4*>(et + rr * LDP + 4 * c4);
        }
    } else {
        const float* tx = a.text + (size_t)b * N * D;
        float* oo = a.out + (size_t)b * N * 4 * D;
        float* bo = a.bsave + (size_t)b * N * D;
        for (int rr = wave; rr < 16 * NW; rr += NW) {
            const int gn = row0 + rr;
            if (gn < N && 4 * c4 < D) {
                const f4 t = *reinterpret_cast<const f4*>(tx + (size_t)gn * D + 4 * c4);
                const f4 av = *reinterpret_cast<const f4*>(et + rr * LDP + 4 * c4);
                float* o = oo + (size_t)gn * 4 * D + 4 * c4;
                *reinterpret_cast<f4*>(o + D) = av;
                *reinterpret_cast<f4*>(o + 2 * D) = t * av;
            }
        }
        park(O1);
        for (int rr = wave; rr < 16 * NW; rr += NW) {
            const int gn = row0 + rr;
            if (gn < N && 4 * c4 < D) {
                const f4 t = *reinterpret_cast<const f4*>(tx + (size_t)gn * D + 4 * c4);
                const f4 bv = *reinterpret_cast<const f4*>(et + rr * LDP + 4 * c4);
                *reinterpret_cast<f4*>(oo + (size_t)gn * 4 * D + 3 * D + 4 * c4) = t * bv;
                *reinterpret_cast<f4*>(bo + (size_t)gn * D + 4 * c4) = bv;
            }
        }
    }
}

template <int NW, int PR, bool DB = false>
__global__ __launch_bounds__(NW * 64) void att_col_kernel(const AttFwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    att_fwd_body<1, NW, PR, DB>(a, smem);
}
template <int NW, int PR, bool DB = false>
__global__ __launch_bounds__(NW * 64) void att_row_kernel(const AttFwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    att_fwd_body<2, NW, PR, DB>(a, smem);
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
        const float sc = __expf(part_stat[o * 2] - m);
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
    int B, T, M, D, splits, rows_per_split;
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
template <int NW, int PR>
__global__ __launch_bounds__(NW * 64) void att_bwd_j1_kernel(const AttBwdArgs a) {
    constexpr int NTHR = NW * 64, NB = PR / 16;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, kg = lane >> 4;
    const int T = a.T, M = a.M, D = a.D;
    int tile, split, b;
    decode_block((M + 16 * NW - 1) / (16 * NW), a.splits, a.B, tile, split, b);
    const int n = (tile * NW + wave) * 16 + r;  // modality row j

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
        stage_panel<NTHR, PR>(pTd, td_b, p0, row_end, D, tid);
        stage_panel<NTHR, PR>(pDa, da_b, p0, row_end, D, tid);
        stage_panel<NTHR, PR>(pDb, db_b, p0, row_end, D, tid);
        if (tid < PR) {
            const int i = p0 + tid;
            const bool in = i < row_end;
            rt_s[tid] = in ? a.rterm[(size_t)b * T + i] : 0.f;
            rmax_s[tid] = in ? a.row_stat[((size_t)b * T + i) * 2] : INFINITY;   // exp(x - inf) = 0 beyond the slice
            rinv_s[tid] = in ? 1.0f / a.row_stat[((size_t)b * T + i) * 2 + 1] : 0.f;
            dl1_s[tid] = in ? a.delta1[(size_t)b * T + i] : 0.f;
        }
        __syncthreads();
        f4 s[NB], dp[NB];
#pragma unroll
        for (int q = 0; q < NB; ++q) s[q] = dp[q] = f4{0.f, 0.f, 0.f, 0.f};
        sprodN<NB>(pTd, r, kg, sideS, s);
        sprodN<NB>(pDa, r, kg, sideM, dp);
        sprodN<NB>(pDb, r, kg, sideQ, dp);
        f4 p1[NB], ds[NB];
#pragma unroll
        for (int mb = 0; mb < NB; ++mb)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int ml = mb * 16 + 4 * kg + e;
                const float x = mm ? s[mb][e] + rt_s[ml] + cterm : NEG;
                const float p = __expf(x - rmax_s[ml]) * rinv_s[ml];  // rinv = 0 beyond the slice
                p1[mb][e] = p;
                const float g = p * (dp[mb][e] - dl1_s[ml]) * mmf;
                ds[mb][e] = g;
                dc += g;
            }
#pragma unroll
        for (int mb = 0; mb < NB; ++mb) {
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
template <int NW, int PR>
__global__ __launch_bounds__(NW * 64) void att_bwd_j2_kernel(const AttBwdArgs a) {
    constexpr int NTHR = NW * 64, NB = PR / 16;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, kg = lane >> 4;
    const int T = a.T, M = a.M, D = a.D;
    int tile, split, b;
    decode_block((M + 16 * NW - 1) / (16 * NW), a.splits, a.B, tile, split, b);
    const int n = (tile * NW + wave) * 16 + r;

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
        stage_panel<NTHR, PR>(pT, t_b, p0, row_end, D, tid);
        if (sep) stage_panel<NTHR, PR>(pTd, td_b, p0, row_end, D, tid);
        if (tid < PR) {
            const int i = p0 + tid;
            const bool in = i < row_end;
            rt_s[tid] = in ? a.rterm[(size_t)b * T + i] : 0.f;
            code_s[tid] = in ? (a.text_mask[(size_t)b * T + i] ? 2 : 1) : 0;
        }
        __syncthreads();
        f4 s[NB], dp[NB];
#pragma unroll
        for (int q = 0; q < NB; ++q) s[q] = dp[q] = f4{0.f, 0.f, 0.f, 0.f};
        sprodN<NB>(pTd, r, kg, sideS, s);
        sprodN<NB>(pT, r, kg, sideDq, dp);
        f4 ds[NB];
#pragma unroll
        for (int mb = 0; mb < NB; ++mb)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int ml = mb * 16 + 4 * kg + e;
                const int code = code_s[ml];
                const float x = code == 2 ? s[mb][e] + rt_s[ml] + cterm : NEG;
                const float p = code ? __expf(x - cmax) * cinv : 0.f;
                const float g = code == 2 ? p * (dp[mb][e] - delta2) : 0.f;
                ds[mb][e] = g;
                dc += g;
            }
#pragma unroll
        for (int mb = 0; mb < NB; ++mb) pvprod(pTd, mb, r, kg, ds[mb], dmd);
    }
    dc = kg_allsum(dc);
    if (!nin) return;
    const size_t prow = ((size_t)b * a.splits + split) * M + n;
    store_side(a.p_dmd2 + prow * D, dmd, D, kg);
    if (kg == 0) a.p_dc2[prow] = dc;
}

// j-side epilogue: one wave per JF_ROWS modality rows, lane = 4 features.  Sums the split partials, writes
//   d_mod_d_j = dc_j w_m + w_tm * dmodd_j ;  d_mod_j = dmodc_j (+ d_mod_d_j when folded)
// and accumulates d_w_m += sum_j dc_j mod_d[j,:]: registers over the wave's rows, LDS across the 4 waves, then ONE
// atomic per feature and workgroup with consecutive lanes on consecutive addresses.
constexpr int JF_ROWS = 4;
__global__ __launch_bounds__(256) void att_bwd_jfin_kernel(const AttBwdArgs a, int B) {
    __shared__ float wred[4][256];
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
#pragma unroll 4
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
    *reinterpret_cast<f4*>(&wred[wave][d]) = wacc;
    __syncthreads();
    const int t = threadIdx.x;
    if (t < D) atomicAdd(a.d_w_m + t, (wred[0][t] + wred[1][t]) + (wred[2][t] + wred[3][t]));
}

// i-side pass (lane side = text rows i, streams all modality rows j):
//   dS = P1 (dP1 - delta1_i) mask_j + P2 (dP2 - delta2_j) mask_i
//   d_text_i += sum_j P2_ij dq_j ; dX_i = sum_j dS_ij mod_d_j ; dr_i = sum_j dS_ij
//   d_text_d_i = dr_i w_t + w_tm * dX_i ; d_w_t += dr_i text_d_i ; d_w_tm += dX_i * text_d_i ; d_bias += dr_i
constexpr int PI_STRIDE = 2 * DT * 16 + 16;  // per-wave partial: [d_w_t 208 | d_w_tm 208 | d_bias 1 ...]
template <int NW, int PR>
__global__ __launch_bounds__(NW * 64) void att_bwd_i_kernel(const AttBwdArgs a) {
    constexpr int NTHR = NW * 64, NB = PR / 16;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, kg = lane >> 4;
    const int T = a.T, M = a.M, D = a.D;
    int tile, split, b;
    decode_block((T + 16 * NW - 1) / (16 * NW), 1, a.B, tile, split, b);
    const int n = (tile * NW + wave) * 16 + r;  // text row i

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
        stage_panel<NTHR, PR>(pM, m_b, p0, M, D, tid);
        stage_panel<NTHR, PR>(pQ, q_b, p0, M, D, tid);
        stage_panel<NTHR, PR>(pDq, dq_b, p0, M, D, tid);
        if (sep) stage_panel<NTHR, PR>(pMd, md_b, p0, M, D, tid);
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
        f4 s[NB], dp1[NB], dp2[NB];
#pragma unroll
        for (int q = 0; q < NB; ++q) s[q] = dp1[q] = dp2[q] = f4{0.f, 0.f, 0.f, 0.f};
        sprodN<NB>(pMd, r, kg, sideS, s);
        sprodN<NB>(pM, r, kg, sideDa, dp1);
        sprodN<NB>(pQ, r, kg, sideDb, dp1);
        sprodN<NB>(pDq, r, kg, sideT, dp2);
        f4 p2[NB], ds[NB];
#pragma unroll
        for (int mb = 0; mb < NB; ++mb)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int ml = mb * 16 + 4 * kg + e;
                const float mf = mmf_s[ml];
                const float x = s[mb][e] + rterm + ct_s[ml];
                const float P1 = mf >= 0.f ? __expf((mf > 0.f ? x : NEG) - rmax) * rinv : 0.f;
                const float P2 = mf >= 0.f ? __expf((tm ? x : NEG) - cmax_s[ml]) * cinv_s[ml] : 0.f;
                const float g1 = mf > 0.f ? P1 * (dp1[mb][e] - dl1) : 0.f;
                const float g2 = P2 * (dp2[mb][e] - dl2_s[ml]) * tmf;
                p2[mb][e] = P2;
                ds[mb][e] = g1 + g2;
                dr += g1 + g2;
            }
#pragma unroll
        for (int mb = 0; mb < NB; ++mb) {
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
// workgroup geometries that are compiled (NW waves x PR-row panels); chosen per kernel below / by MMB_ATT_GEOM
// (measured on MI355X, cfg2: 4 waves x 32-row panels beats 2x16, 1x16 and 2x32 on every kernel)
enum { GEOM_4x32 = 0, GEOM_2x16 = 1, GEOM_4x16DB = 2, GEOM_COUNT };   // DB = double-buffered panels (forward kernels)
static const int kGeomNW[GEOM_COUNT] = {4, 2, 4};
static const int kGeomPR[GEOM_COUNT] = {32, 16, 16};

static int geom_for(int kernel_id, int deflt) {
    // MMB_ATT_GEOM = "<col><row><j1><j2><i>" one digit per kernel (tuning aid), e.g. 11111
    static int forced[5] = {-2, -2, -2, -2, -2};
    if (forced[0] == -2) {
        const char* e = getenv("MMB_ATT_GEOM");
        for (int i = 0; i < 5; ++i) forced[i] = (e && (int)strlen(e) > i && e[i] >= '0' && e[i] < '0' + GEOM_COUNT) ? e[i] - '0' : -1;
    }
    return forced[kernel_id] >= 0 ? forced[kernel_id] : deflt;
}

// How many ways to split the streamed side (R rows, PR-row panels) of a sweep whose lane side has N rows per sample:
// the workgroups (4 waves = 64 lane-side rows each) run in rounds of `slots` (256 CUs x workgroups that fit a CU's LDS),
// so the cost is rounds x (panels per split + fixed per-workgroup work) plus the traffic of the per-split partials.
// (The earlier "1.5 waves per SIMD" rule gave 384 workgroups for both cfg2 attentions: 1.5 rounds.)
static int pick_splits(int B, int N, int R, int PR, int slots = 256) {
    const long tiles = (long)B * ((N + 63) / 64);
    const int smax = (R + PR - 1) / PR;
    int best = 1;
    double best_cost = 1e30;
    for (int s = 1; s <= smax; ++s) {
        const long rounds = (tiles * s + slots - 1) / slots;
        const int panels = ((R + s - 1) / s + PR - 1) / PR;
        const double cost = (double)rounds * (panels + 1.5) + 0.15 * s;
        if (cost < best_cost) { best_cost = cost; best = s; }
    }
    return best;
}
static int rows_per_split(int R, int splits, int PR) {
    int rp = (R + splits - 1) / splits;
    return (rp + PR - 1) / PR * PR;
}

struct BwdWs {
    size_t da, db, delta1, dq, delta2, p_dq, p_dmc, p_dmd1, p_dmd2, p_dc1, p_dc2, total;
    int splits;
};
static BwdWs bwd_layout(int B, int T, int M, int D) {
    BwdWs w{};
    w.splits = pick_splits(B, M, T, 16) > pick_splits(B, M, T, 32) ? pick_splits(B, M, T, 16) : pick_splits(B, M, T, 32);   // upper bound over the compiled geometries
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

// launch `KERNEL<..., NW, PR>` for geometry `geom` on a 1-D grid of tiles(N) * splits * B workgroups
#define MMB_ATT_LAUNCH(KID, GEOM, N_LANE, SPLITS, BATCH, LDS_FLOATS_EXPR, ARGS, ...)                                   \
    do {                                                                                                               \
        int rc_ = MMB_OK;                                                                                              \
        auto go = [&](auto kern, int NW, int PR) {                                                                     \
            const size_t lds = (size_t)(LDS_FLOATS_EXPR) * sizeof(float);                                              \
            if ((rc_ = allow_lds(kern, lds))) return;                                                                  \
            const int tiles = ((N_LANE) + 16 * NW - 1) / (16 * NW);                                                    \
            ProfScope ps_(KID, stream);                                                                                \
            hipLaunchKernelGGL(kern, dim3(tiles * (SPLITS) * (BATCH)), dim3(NW * 64), lds, stream, ARGS);              \
        };                                                                                                             \
        switch (GEOM) {                                                                                                \
            case GEOM_2x16: go(__VA_ARGS__<2, 16>, 2, 16); break;                                                      \
            default: go(__VA_ARGS__<4, 32>, 4, 32); break;                                                             \
        }                                                                                                              \
        if (rc_) return rc_;                                                                                           \
        MMB_HIP(hipGetLastError());                                                                                    \
    } while (0)

}  // namespace mmb

using namespace mmb;

static int check_att_dims(int B, int T, int M, int D) {
    MMB_REQUIRE(B >= 1 && T >= 1 && M >= 1, "bidaf: bad sizes B=%d T=%d M=%d", B, T, M);
    MMB_REQUIRE(D >= 4 && D % 4 == 0 && D <= MMB_ATT_GENERAL_MAX_D, "bidaf: D=%d must be a multiple of 4 and <= %d", D,
                MMB_ATT_GENERAL_MAX_D);
    return MMB_OK;
}

extern "C" size_t mmb_bidaf_fwd_workspace_bytes(int B, int T, int M, int D) {
    if (B < 1 || T < 1 || M < 1 || D < 4 || D <= MMB_ATT_MAX_D) return 0;   // the fused kernels need no scratch
    return bidaf_big_fwd_ws_floats(B, T, M, D) * sizeof(float);
}

extern "C" int mmb_bidaf_fwd(const float* text, const float* mod, const uint8_t* text_mask, const uint8_t* mod_mask,
                             const float* text_d, const float* mod_d, const float* w_t, const float* w_m,
                             const float* w_tm, const float* bias, float* out, float* q, float* bsave, float* rterm,
                             float* cterm, float* row_stat, float* col_stat, float* workspace, size_t workspace_bytes,
                             int B, int T, int M, int D, int device, void* stream_) {
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
    if (D > MMB_ATT_MAX_D) {   // general-size path (bidaf_big.hip): similarity matrix materialised in the workspace
        MMB_REQUIRE(workspace && workspace_bytes >= mmb_bidaf_fwd_workspace_bytes(B, T, M, D),
                    "mmb_bidaf_fwd: D=%d > %d needs a workspace of mmb_bidaf_fwd_workspace_bytes()", D, MMB_ATT_MAX_D);
        return bidaf_big_fwd(text, mod, text_mask, mod_mask, text_d, mod_d, w_tm, out, q, bsave, rterm, cterm, row_stat, col_stat,
                             workspace, B, T, M, D, stream);
    }

    // ---- column pass: lane side = modality rows, streams text.  (`out` is used as scratch for the split
    //      partials: it is (B,T,4D) and is only written by the row pass afterwards.)
    {
        const int geom = geom_for(0, GEOM_4x32);
        const int PRg = kGeomPR[geom];
        AttFwdArgs a{};
        a.side_src = mod_d; a.w_tm = w_tm; a.mS = text_d; a.mV0 = text; a.mV1 = nullptr;
        a.m_mask = text_mask; a.m_term = rterm; a.n_term = cterm; a.stat = col_stat; a.q = q;
        a.N = M; a.R = T; a.D = D; a.B = B;
        int splits = pick_splits(B, M, T, PRg, 512);   // 54 KB of LDS: two workgroups per CU
        while (splits > 1 && (size_t)splits * M * (D + 2) > (size_t)T * 4 * D) --splits;
        a.splits = splits;
        a.rows_per_split = rows_per_split(T, splits, PRg);
        a.part_o = out;
        a.part_stat = out + (size_t)B * splits * M * D;
        if (geom == GEOM_4x16DB) {
            const size_t lds = ((size_t)2 * (1 + 1) * 16 * LDP + 2 * a.rows_per_split) * sizeof(float);
            if (int rc = allow_lds(att_col_kernel<4, 16, true>, lds)) return rc;
            ProfScope ps_(MMB_K_ATT_COL, stream);
            hipLaunchKernelGGL((att_col_kernel<4, 16, true>), dim3(((M + 63) / 64) * splits * B), dim3(256), lds, stream, a);
            MMB_HIP(hipGetLastError());
        } else {
            MMB_ATT_LAUNCH(MMB_K_ATT_COL, geom, M, splits, B, (1 + 1) * PR * LDP + 2 * a.rows_per_split, a, att_col_kernel);
        }
        if (splits > 1) {
            const size_t nthr = (size_t)B * M * (D / 4);
            { ProfScope ps_(MMB_K_ATT_COMBINE, stream); hipLaunchKernelGGL(att_combine_kernel, dim3((nthr + 255) / 256), dim3(256), 0, stream, a.part_o, a.part_stat, q,
                               col_stat, B, M, D, splits); }
            MMB_HIP(hipGetLastError());
        }
    }
    // ---- row pass: lane side = text rows, streams [mod | q]
    {
        const int geom = geom_for(1, GEOM_4x32);
        AttFwdArgs a{};
        a.side_src = text_d; a.w_tm = w_tm; a.mS = mod_d; a.mV0 = mod; a.mV1 = q;
        a.m_mask = mod_mask; a.m_term = cterm; a.n_term = rterm; a.stat = row_stat;
        a.text = text; a.out = out; a.bsave = bsave;
        a.N = T; a.R = M; a.D = D; a.B = B; a.splits = 1; a.rows_per_split = rows_per_split(M, 1, kGeomPR[geom]);
        { const char* e = getenv("MMB_ATT_DBG"); a.dbg = e ? atoi(e) : 0; }
        if (geom == GEOM_4x16DB) {
            const size_t lds = ((size_t)2 * (2 + 1) * 16 * LDP + 2 * a.rows_per_split) * sizeof(float);
            if (int rc = allow_lds(att_row_kernel<4, 16, true>, lds)) return rc;
            ProfScope ps_(MMB_K_ATT_ROW, stream);
            hipLaunchKernelGGL((att_row_kernel<4, 16, true>), dim3(((T + 63) / 64) * B), dim3(256), lds, stream, a);
            MMB_HIP(hipGetLastError());
        } else {
            MMB_ATT_LAUNCH(MMB_K_ATT_ROW, geom, T, 1, B, (2 + 1) * PR * LDP + 2 * a.rows_per_split, a, att_row_kernel);
        }
    }
    return MMB_OK;
}

extern "C" size_t mmb_bidaf_bwd_workspace_bytes(int B, int T, int M, int D) {
    if (B < 1 || T < 1 || M < 1 || D < 4) return 0;
    if (D > MMB_ATT_MAX_D) return bidaf_big_bwd_ws_floats(B, T, M, D) * sizeof(float);
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
    MMB_REQUIRE(workspace_bytes >= mmb_bidaf_bwd_workspace_bytes(B, T, M, D), "mmb_bidaf_bwd: workspace too small (%zu < %zu)",
                workspace_bytes, mmb_bidaf_bwd_workspace_bytes(B, T, M, D));
    MMB_HIP(hipSetDevice(device));
    if (D > MMB_ATT_MAX_D)
        return bidaf_big_bwd(d_out, out, text, mod, text_mask, mod_mask, text_d, mod_d, w_t, w_m, w_tm, q, bsave, rterm, cterm,
                             row_stat, col_stat, d_text, d_mod, d_text_d, d_mod_d, d_w_t, d_w_m, d_w_tm, d_bias, workspace, B, T, M,
                             D, stream);
    const BwdWs L = bwd_layout(B, T, M, D);

    const int geom_j1 = geom_for(2, GEOM_4x32), geom_j2 = geom_for(3, GEOM_4x32), geom_i = geom_for(4, GEOM_4x32);
    // both j sweeps must agree on how the text rows are split (the partial buffers are indexed by split)
    const int PRj = kGeomPR[geom_j1] > kGeomPR[geom_j2] ? kGeomPR[geom_j1] : kGeomPR[geom_j2];

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
    a.B = B; a.T = T; a.M = M; a.D = D; a.fold = drop_t ? 0 : 1;
    a.splits = pick_splits(B, M, T, PRj);
    if (a.splits > L.splits) a.splits = L.splits;
    a.rows_per_split = rows_per_split(T, a.splits, PRj);

    { ProfScope ps_(MMB_K_ATT_BWD_PRE, stream); hipLaunchKernelGGL(att_bwd_pre_kernel, dim3((B * T + 3) / 4), dim3(256), 0, stream, d_out, out, text, bsave,
                       workspace + L.da, workspace + L.db, workspace + L.delta1, d_text, d_w_t, d_w_m, d_w_tm, d_bias, B * T, D); }
    MMB_HIP(hipGetLastError());
    MMB_ATT_LAUNCH(MMB_K_ATT_BWD_J1, geom_j1, M, a.splits, B, 3 * PR * LDP + 4 * PR, a, att_bwd_j1_kernel);
    MMB_ATT_LAUNCH(MMB_K_ATT_BWD_J2, geom_j2, M, a.splits, B, 2 * PR * LDP + 2 * PR, a, att_bwd_j2_kernel);
    {
        const int chunks = (B * M + JF_ROWS - 1) / JF_ROWS;
        { ProfScope ps_(MMB_K_ATT_BWD_JFIN, stream); hipLaunchKernelGGL(att_bwd_jfin_kernel, dim3((chunks + 3) / 4), dim3(256), 0, stream, a, B); }
        MMB_HIP(hipGetLastError());
    }
    MMB_ATT_LAUNCH(MMB_K_ATT_BWD_I, geom_i, T, 1, B, 4 * PR * LDP + 5 * PR + NW * PI_STRIDE, a, att_bwd_i_kernel);
    return MMB_OK;
}
