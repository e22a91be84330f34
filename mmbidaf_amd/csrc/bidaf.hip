// Fused BiDAF attention for gfx950 (forward + backward) on the 16-bit matrix cores at fp32 accuracy.
// Replaces BiDAFAttention.forward / get_similarity_matrix / masked_softmax of the reference
// (layers/attention.py:37-98) and their autograd.  The (B,T,M) similarity matrix, both softmaxes
// and the (B,T,T) product s1.s2^T of the reference are never materialised:
//
//   S_ij  = r_i + c_j + sum_d text_d[i,d] w_tm[d] mod_d[j,d]      r = text_d.w_t + bias, c = mod_d.w_m
//   P1    = softmax_j(mask_mod ? S : -1e30)   P2 = softmax_i(mask_text ? S : -1e30)   (blend, attention.py:94)
//   q     = P2^T text      a = P1 mod      b = P1 q      out = [text, a, text*a, text*b]
//
// Arithmetic.  Every contraction runs on v_mfma_f32_16x16x32_f16 from an error-compensated split of the fp32
// operands: x * s = h0 + h1 with h0 = fp16(x s), h1 = fp16(x s - h0), s a power of two per operand ROW chosen so
// that the row maximum lands in [2^13, 2^14) (exact scaling; 22 significant bits); a product is the three cross
// terms a0 b1 + a1 b0 + a0 b0 accumulated in fp32 (the dropped a1 b1 is below 2^-22), i.e. 3 MFMAs of K = 32 in 48
// cycles where the exact-f32 MFMA needs 256: the same scheme as the LSTM GEMMs (planes.hip), 5.3x the fp32 rate at
// fp32 error (~1e-6 of the operand scale).  Streamed operands are split ONCE by the producing kernel into "planes"
// (tiled exactly as the LDS image, below) together with one inverse scale per row; probabilities and softmax
// gradients are split in registers (v_cvt_pkrtz pairs) straight out of the accumulators.
//
// Tile engine (per wave).  A wave owns 16 rows "n" of one side (the LANE side: n = l & 15, k-group g = l >> 4) and
// streams the other side ("m" rows) through 32-row LDS panels:
//   S-type  C[m][n] = sum_d P[m][d] side[n][d]     A = panel rows (ds_read_b128: 8 consecutive d of one plane),
//           B = lane-side registers (7 k-tiles x 2 planes x 4 VGPRs); accumulator lane (n, g) holds m = 4g + e.
//   PV-type O[d][n] += sum_m V[m][d] W[m][n]       A = V^T through ds_read_b64_tr_b16 (the hardware transpose read:
//           the same row-major panel image serves both kinds of product), B = the S-type accumulators of the
//           panel's two 16-row blocks themselves: lane (n, g) holds rows 4g..4g+3 of each block, and the MFMA's
//           k index is simply DEFINED as k = 8g + j <-> row (j < 4 ? 4g + j : 16 + 4g + j - 4), which the transpose
//           reads follow.  Probabilities therefore never leave registers.  O lands as O[n][d = 16 dt + 4g + e].
// Scales.  S-type: acc * inv_m[m] * inv_n.  PV-type: the row scale of V sits inside the K sum, so it is folded
// into W before the split: W' = W * inv_V[m] * c with c a power of two that maps the largest possible |W inv_V|
// of this lane's column to 2^14 (W = probabilities: c = 2^14 / max inv_V; W = softmax gradients: an a-priori
// Cauchy-Schwarz bound from the operands' row scales -- the two-term split keeps 2^-25 absolute precision over 40
// binary orders below the maximum, so a loose bound costs nothing); the accumulator is divided by c at the end.
//
// Planes layout (one tensor, one sample, R rows padded to a multiple of 32): 1-KiB chunks
// [row block of 16][k tile of 32 features][plane 0|1], chunk = 16 rows x 64 B with the four 16-B slots of a row
// XOR-ed by 2 * bit 2 of the row: conflict-free for the b128 row reads AND for the transpose reads (a 32-lane half
// then covers all 64 banks exactly once).  A 32-row panel is one contiguous 28-KiB run: 28 LDS-DMA
// wave-instructions (global_load_lds_dwordx4), no VGPR round trip, no per-element work in the panel loop.
#include <math.h>
#include <stdlib.h>

#include "common.h"

namespace mmb {

constexpr int KT = 7;              // 32-deep k tiles: features padded to 224
constexpr int DT = 13;             // 16-wide output feature tiles: D <= 208
constexpr int PR = 32;             // rows per panel
constexpr int NW = 4;              // waves per workgroup (64 lane-side rows)
constexpr int NTHR = NW * 64;
constexpr int PCH = 2048;          // bytes of one (row block, k tile): two 1-KiB planes
constexpr int PRB = KT * PCH;      // bytes of one 16-row block of a tensor
constexpr int PANEL_B = 2 * PRB;   // bytes of a 32-row panel (28 KiB)
constexpr int LDP = 212;           // row stride (floats) of the epilogue's LDS staging tile
constexpr float NEG = -1e30f;      // attention.py:94
constexpr float WMAX = 16384.0f;   // 2^14: where the largest split operand is mapped

// timing-only ablations (mmb_set_att_debug / MMB_ATT_DBG; never set by the product path; results are then WRONG):
// 1 = stage only the first panel, 2 = no S-type products, 4 = no PV-type products, 8 = no epilogue stores,
// 16 = no panel loop at all (prologue + epilogue only)
static int g_att_dbg = -1;
static int att_dbg() {
    if (g_att_dbg < 0) {
        const char* e = getenv("MMB_ATT_DBG");
        g_att_dbg = e ? atoi(e) : 0;
    }
    return g_att_dbg;
}

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef short v4s __attribute__((__vector_size__(4 * sizeof(short))));
typedef short s8v __attribute__((ext_vector_type(8)));
typedef unsigned u4v __attribute__((ext_vector_type(4)));

struct side_t {
    half8 h[KT][2];                // [k tile][plane]: features 32 kt + 8 g .. + 7 of row n
};
using acc_t = f4[DT];

__device__ __forceinline__ void zero_acc(acc_t& s) {
#pragma unroll
    for (int i = 0; i < DT; ++i) s[i] = f4{0.f, 0.f, 0.f, 0.f};
}

__host__ __device__ __forceinline__ int pad32(int r) { return (r + 31) / 32 * 32; }
// bytes of one sample's planes of an R-row tensor
__host__ __device__ __forceinline__ size_t planes_sample_bytes(int R) { return (size_t)(pad32(R) / 16) * PRB; }

__device__ __forceinline__ int att_swz(int row) { return ((row >> 2) & 1) << 1; }
// byte offset of (row, 16-B octet 0..27) of plane 0 inside one sample's planes; plane 1 follows at +1024
__device__ __forceinline__ int pl_off_att(int row, int oct) {
    const int rl = row & 15;
    return (row >> 4) * PRB + (oct >> 2) * PCH + rl * 64 + (((oct & 3) ^ att_swz(rl)) << 4);
}

// power of two s with s * amax in [2^13, 2^14)  (1 for amax = 0 or out of range)
__device__ __forceinline__ float a_pow2_scale(float amax) {
    const unsigned u = __float_as_uint(amax);
    const int e = (int)((u >> 23) & 0xFF) - 127;
    if (amax <= 0.0f || e > 100 || e < -100) return 1.0f;
    return __uint_as_float((unsigned)(13 - e + 127) << 23);
}
// largest power of two <= x (x > 0 finite)
__device__ __forceinline__ float pow2_floor(float x) { return __uint_as_float(__float_as_uint(x) & 0x7F800000u); }

// two-term fp16 split of 8 already scaled values (round to nearest; clamped so a violated bound saturates)
__device__ __forceinline__ void a_split2h(const float* x, half8& h0, half8& h1) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float v = fminf(fmaxf(x[j], -60000.0f), 60000.0f);
        const _Float16 a = (_Float16)v;
        h0[j] = a;
        h1[j] = (_Float16)(v - (float)a);
    }
}

// reductions over the 4 k-groups of a row (lanes r, r+16, r+32, r+48) with the gfx950 half / row swaps: one VALU
// instruction per step instead of a ds_bpermute round trip.  v_permlane32_swap(x, x) leaves [lo | lo] and [hi | hi],
// v_permlane16_swap(x, x) leaves rows [0 0 2 2] and [1 1 3 3]: combining the pair is the xor-32 / xor-16 exchange.
// Inline asm, not __builtin_amdgcn_permlane{16,32}_swap: hipcc (ROCm 7.2) copy-propagates across the builtin as if
// its second operand were not written (a swap of two copies of one value then yields "p + p"); the two v_nop are the
// wait states a VALU write of an operand needs before the swap reads it.
__device__ __forceinline__ void kg_pairs(float v, int step, float& p, float& q) {
    p = v;
    q = v;
    if (step == 32) asm volatile("v_nop\n\tv_nop\n\tv_permlane32_swap_b32 %0, %1" : "+v"(p), "+v"(q));
    else asm volatile("v_nop\n\tv_nop\n\tv_permlane16_swap_b32 %0, %1" : "+v"(p), "+v"(q));
}
__device__ __forceinline__ float kg_allsum(float v) {
    float p, q;
    kg_pairs(v, 32, p, q);
    v = p + q;
    kg_pairs(v, 16, p, q);
    return p + q;
}
__device__ __forceinline__ float kg_allmax(float v) {
    float p, q;
    kg_pairs(v, 32, p, q);
    v = fmaxf(p, q);
    kg_pairs(v, 16, p, q);
    return fmaxf(p, q);
}
__device__ __forceinline__ float r_allsum(float v) {  // over the 16 lanes of a k-group
    v += __shfl_xor(v, 1);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 4);
    v += __shfl_xor(v, 8);
    return v;
}
__device__ __forceinline__ float half_allsum(float v) {  // over the 32 lanes of a half wave
    v += __shfl_xor(v, 16);
    return r_allsum(v);
}
__device__ __forceinline__ float half_allmax(float v) {
#pragma unroll
    for (int o = 16; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float f4sum(const f4 v) { return (v.x + v.y) + (v.z + v.w); }
__device__ __forceinline__ float f4amax(const f4 v) { return fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))); }

// ---- lane-side operand of row n from an fp32 row (optionally scaled feature-wise by w): the row's own power-of-two
// scale from its maximum over the lane's 56 values and the 4 k-groups; rows n >= N come out as zeros with inv = 0
__device__ __forceinline__ void side_from_regs(float (&x)[KT][8], side_t& sd, float& inv_n) {
    float amax = 0.f;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int j = 0; j < 8; ++j) amax = fmaxf(amax, fabsf(x[kt][j]));
    amax = kg_allmax(amax);
    const float s = a_pow2_scale(amax);
    inv_n = amax > 0.f ? 1.0f / s : 0.f;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
#pragma unroll
        for (int j = 0; j < 8; ++j) x[kt][j] *= s;
        a_split2h(x[kt], sd.h[kt][0], sd.h[kt][1]);
    }
}
__device__ __forceinline__ void load_row_regs(float (&x)[KT][8], const float* src_b, int n, int N, int D, int g, const float* w) {
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int d = 32 * kt + 8 * g + 4 * h;
            f4 v = f4{0.f, 0.f, 0.f, 0.f};
            if (n < N && d < D) {
                v = *reinterpret_cast<const f4*>(src_b + (size_t)n * D + d);
                if (w) v *= *reinterpret_cast<const f4*>(w + d);
            }
            x[kt][4 * h] = v.x; x[kt][4 * h + 1] = v.y; x[kt][4 * h + 2] = v.z; x[kt][4 * h + 3] = v.w;
        }
}
__device__ __forceinline__ void load_side_f32(side_t& sd, float& inv_n, const float* src_b, int n, int N, int D, int g, const float* w) {
    float x[KT][8];
    load_row_regs(x, src_b, n, N, D, g, w);
    side_from_regs(x, sd, inv_n);
}
// ---- lane-side operand straight from planes (already split by their producer): 14 16-B loads
__device__ __forceinline__ void load_side_planes(side_t& sd, float& inv_n, const char* planes_b, const float* inv_b, int n, int N, int g) {
    const int nn = min(n, pad32(N) - 1);
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
        const char* p = planes_b + pl_off_att(nn, 4 * kt + g);
        sd.h[kt][0] = *reinterpret_cast<const half8*>(p);
        sd.h[kt][1] = *reinterpret_cast<const half8*>(p + 1024);
    }
    inv_n = n < N ? inv_b[n] : 0.f;
}
// value of the lane's features of a planes row, reconstructed to fp32 (x = (h0 + h1) * inv)
__device__ __forceinline__ float side_dot_regs(const side_t& sd, float inv_n, const float (&x)[KT][8]) {
    float acc = 0.f;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += ((float)sd.h[kt][0][j] + (float)sd.h[kt][1][j]) * x[kt][j];
    return acc * inv_n;
}

// ---- stage the 32-row panel starting at row p0 (a multiple of 32) of one sample's planes: 28 contiguous KiB,
// 7 LDS-DMA wave-instructions per wave, LDS image lane-linear = the global image
__device__ __forceinline__ void stage_panel(char* panel, const char* planes_b, int p0, int tid) {
    const char* src = planes_b + (size_t)(p0 >> 4) * PRB + (tid & 63) * 16;
    const int wave = tid >> 6;
#pragma unroll
    for (int k = 0; k < PANEL_B / 1024 / NW; ++k) {
        const int piece = wave + NW * k;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + piece * 1024),
                                         (__attribute__((address_space(3))) void*)(panel + piece * 1024), 16, 0, 0);
    }
}

__device__ __forceinline__ f4 mfma_h(const half8 a, const half8 b, const f4 c) {
    // v_mfma_f32_16x16x32_f16: A[row l&15][k = 8(l>>4)+j], B[k][col l&15], C[row 4(l>>4)+e][col l&15]
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

// S-type product of the panel's two 16-row blocks against the lane-side registers: c[mb][e] = row 16 mb + 4g + e.
// Software-pipelined by hand: the 4 b128 reads of k tile kt+1 are issued before the 6 MFMAs of k tile kt and interleaved
// with them (at one wave per SIMD hipcc otherwise waits out every LDS latency in front of the MFMAs that need it).
template <bool PIPE = true>
__device__ __forceinline__ void sprod2(const char* panel, int r, int g, const side_t& side, f4 (&c)[2]) {
    const char* p = panel + r * 64 + ((g ^ att_swz(r)) << 4);
    if (!PIPE) {   // register-starved kernels (3-4 lane-side operands): plain order, the compiler schedules
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            half8 a0[2], a1[2];
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                a0[mb] = *reinterpret_cast<const half8*>(p + (mb * KT + kt) * PCH);
                a1[mb] = *reinterpret_cast<const half8*>(p + (mb * KT + kt) * PCH + 1024);
            }
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) c[mb] = mfma_h(a0[mb], side.h[kt][1], c[mb]);
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) c[mb] = mfma_h(a1[mb], side.h[kt][0], c[mb]);
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) c[mb] = mfma_h(a0[mb], side.h[kt][0], c[mb]);
        }
        return;
    }
    half8 fr[2][4];   // [buffer][mb * 2 + plane]
    auto ld = [&](int kt, half8 (&x)[4]) {
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
            x[2 * mb] = *reinterpret_cast<const half8*>(p + (mb * KT + kt) * PCH);
            x[2 * mb + 1] = *reinterpret_cast<const half8*>(p + (mb * KT + kt) * PCH + 1024);
        }
    };
    ld(0, fr[0]);
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
        const half8(&x)[4] = fr[kt & 1];
        if (kt + 1 < KT) ld(kt + 1, fr[(kt + 1) & 1]);
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) c[mb] = mfma_h(x[2 * mb], side.h[kt][1], c[mb]);
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) c[mb] = mfma_h(x[2 * mb + 1], side.h[kt][0], c[mb]);
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) c[mb] = mfma_h(x[2 * mb], side.h[kt][0], c[mb]);
        if (kt + 1 < KT) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        }
    }
}

// per-lane byte offsets of the transpose reads: lane 4q+p of a 16-lane group supplies row 4g+q, features 4p..4p+3 of the
// 16-feature tile; tr[dt & 1] is the offset inside the (row block, k tile = dt >> 1) chunk of plane 0
struct tr_off {
    int o[2];
};
__device__ __forceinline__ tr_off make_tr_off(int lane) {
    const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    const int row = 4 * g + q;
    tr_off t;
#pragma unroll
    for (int hlf = 0; hlf < 2; ++hlf) t.o[hlf] = row * 64 + (((2 * hlf + (p >> 1)) ^ att_swz(row)) << 4) + (p & 1) * 8;
    return t;
}
__device__ __forceinline__ v4s tr16(const char* lds_addr) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(lds_addr));
}
__device__ __forceinline__ half8 cat44(const v4s lo, const v4s hi) {
    const s8v t = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(half8, t);
}

// PV-type: O[dt] += V^T[16 dt ..][32 rows] . W   with W already split into (W0, W1) (k = 8g+j <-> row as above).
// The 4 transpose reads of feature tile dt+PVD run ahead of the 3 MFMAs of tile dt (ring of PVD+1 fragment sets).
template <int PVD = 3>
__device__ __forceinline__ void pvprod(const char* panel, const tr_off& tr, const half8 W0, const half8 W1, acc_t& O) {
    v4s fr[PVD + 1][4];
    auto ld = [&](int dt, v4s (&x)[4]) {
        const char* base = panel + (dt >> 1) * PCH + tr.o[dt & 1];
        x[0] = tr16(base);
        x[1] = tr16(base + KT * PCH);
        x[2] = tr16(base + 1024);
        x[3] = tr16(base + KT * PCH + 1024);
    };
#pragma unroll
    for (int d = 0; d < PVD; ++d) ld(d, fr[d]);
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
        if (dt + PVD < DT) ld(dt + PVD, fr[(dt + PVD) % (PVD + 1)]);
        const v4s(&x)[4] = fr[dt % (PVD + 1)];
        const half8 A0 = cat44(x[0], x[1]), A1 = cat44(x[2], x[3]);
        O[dt] = mfma_h(A0, W1, O[dt]);
        O[dt] = mfma_h(A1, W0, O[dt]);
        O[dt] = mfma_h(A0, W0, O[dt]);
        if (dt + PVD < DT) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        }
    }
}

// two-term split of the 8 accumulator values a lane holds for the panel (w0: block 0, w1: block 1), truncating
// conversions (v_cvt_pkrtz_f16_f32: 6 VALU per pair); |w| <= ~2^14 by construction
__device__ __forceinline__ void split_w(const f4 w0, const f4 w1, half8& H0, half8& H1) {
    u4v hh, ll;
    auto pk = [](float a, float b, unsigned& h, unsigned& l) {
        const auto h2 = __builtin_amdgcn_cvt_pkrtz(a, b);
        const auto l2 = __builtin_amdgcn_cvt_pkrtz(a - (float)h2[0], b - (float)h2[1]);
        h = __builtin_bit_cast(unsigned, h2);
        l = __builtin_bit_cast(unsigned, l2);
    };
    unsigned h, l;
    pk(w0.x, w0.y, h, l); hh[0] = h; ll[0] = l;
    pk(w0.z, w0.w, h, l); hh[1] = h; ll[1] = l;
    pk(w1.x, w1.y, h, l); hh[2] = h; ll[2] = l;
    pk(w1.z, w1.w, h, l); hh[3] = h; ll[3] = l;
    H0 = __builtin_bit_cast(half8, hh);
    H1 = __builtin_bit_cast(half8, ll);
}

// workgroup-wide maximum of up to 4 non-negative per-thread values (red: >= 4 * NW floats of LDS); all threads get it
template <int NV_>
__device__ __forceinline__ void wg_allmax(float (&v)[NV_], float* red, int tid) {
#pragma unroll
    for (int k = 0; k < NV_; ++k) {
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) v[k] = fmaxf(v[k], __shfl_xor(v[k], o));
    }
    __syncthreads();
    if ((tid & 63) == 0) {
#pragma unroll
        for (int k = 0; k < NV_; ++k) red[(tid >> 6) * 4 + k] = v[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NV_; ++k) {
        float m = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) m = fmaxf(m, red[w * 4 + k]);
        v[k] = m;
    }
}
// power of two c with c * imax = 2^14 (imax = largest inverse scale, itself a power of two); 1 when there is none
__device__ __forceinline__ float cmap(float imax) { return imax > 0.f ? WMAX / imax : 1.0f; }
// power of two c <= 2^14 / (imax * bound): maps the largest possible |W inv_V| of a softmax-gradient operand to <= 2^14
__device__ __forceinline__ float cmap_bound(float imax, float bound) {
    const float den = imax * bound;
    if (!(den > 0.f) || !(den < 1e37f)) return 1.0f;
    const float c = WMAX / den;
    return (c > 1e-30f && c < 1e30f) ? pow2_floor(c) : 1.0f;
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

// mask code of streamed row m: 0 = beyond the range, 1 = masked, 2 = live.  Prefix masks come from the lengths
// (models.py:86-92: mask[b, m] = m < len[b]) when given, arbitrary 0/1 masks from the u8 tensor.
__device__ __forceinline__ int mask_code(bool in, const uint8_t* mask, const int* len, int b, int R, int m) {
    if (!in) return 0;
    const bool live = len ? (m < len[b]) : (mask[(size_t)b * R + m] != 0);
    return live ? 2 : 1;
}
__device__ __forceinline__ bool mask_live(const uint8_t* mask, const int* len, int b, int R, int m) {
    return len ? (m < len[b]) : (mask[(size_t)b * R + m] != 0);
}

// ------------------------------------------------------------------------------------------ split passes
// One wave per row, lane = 4 features (coalesced 16 B per lane).  Up to 4 tensors per launch (blockIdx.y).  Also the rank-1 terms of the similarity: term[b,row] = src[b,row] . w + bias.
struct SplitSrc {
    const float* src;     // (B,R,D)
    char* planes;         // B x planes_sample_bytes(R)
    float* inv;           // (B, pad32(R)) inverse row scales, 0 for all-zero and padding rows
    const float* w;       // (D) or null: rank-1 weight of `term`
    const float* bias;    // (1) or null
    float* term;          // (B,R) or null
    const float* mul;     // (D) or null: the planes hold src * mul feature-wise (the w_tm-folded lane-side operands)
    int R;
};
struct PrepArgs {
    SplitSrc t[6];
    int n, D, B;
};
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float wave_allmax(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float wave_allsum(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
// lane c (< 8 KT) holds features 4c..4c+3 of `row`; amax = the row's max |x| (wave-uniform)
__device__ __forceinline__ void store_split_row(char* planes_b, float* inv_row, int row, int c, f4 x, float amax) {
    const float s = a_pow2_scale(amax);
    if (c < 8 * KT) {
        x = x * s;
        half4 h0, h1;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float v = fminf(fmaxf(x[j], -60000.0f), 60000.0f);
            const _Float16 a = (_Float16)v;
            h0[j] = a;
            h1[j] = (_Float16)(v - (float)a);
        }
        char* d = planes_b + pl_off_att(row, c >> 1) + (c & 1) * 8;
        *reinterpret_cast<half4*>(d) = h0;
        *reinterpret_cast<half4*>(d + 1024) = h1;
    }
    if (c == 0) *inv_row = amax > 0.f ? 1.0f / s : 0.f;
}
// One wave per row (fully coalesced 16 B per lane), 4 rows per workgroup; blockIdx.y = tensor
__global__ __launch_bounds__(256) void att_prep_kernel(const PrepArgs a) {
    const SplitSrc s = a.t[blockIdx.y];
    const int Rp = pad32(s.R);
    const long rowi = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (rowi >= (long)a.B * Rp) return;
    const int b = rowi / Rp, row = rowi - (long)b * Rp, c = threadIdx.x & 63, d = 4 * c;
    f4 x = f4{0.f, 0.f, 0.f, 0.f};
    float dot = 0.f;
    if (row < s.R && d < a.D) {
        x = *reinterpret_cast<const f4*>(s.src + ((size_t)b * s.R + row) * a.D + d);
        if (s.w) dot = f4sum(x * *reinterpret_cast<const f4*>(s.w + d));
        if (s.mul) x = x * *reinterpret_cast<const f4*>(s.mul + d);
    }
    const float amax = wave_allmax(f4amax(x));
    if (s.term) {
        dot = wave_allsum(dot);
        if (c == 0 && row < s.R) s.term[(size_t)b * s.R + row] = dot + (s.bias ? s.bias[0] : 0.f);
    }
    store_split_row(s.planes + (size_t)b * planes_sample_bytes(s.R), s.inv + (size_t)b * Rp + row, row, c, x, amax);
}

// rank-1 terms alone (general-width path, bidaf_big.hip): rterm[b,i] = text_d[b,i].w_t + bias ; cterm[b,j] = mod_d[b,j].w_m
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
    for (int d = lane * 4; d < D; d += 256) acc += f4sum(*reinterpret_cast<const f4*>(src + d) * *reinterpret_cast<const f4*>(w + d));
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) {
        if (is_t) rterm[row] = acc + bias[0];
        else cterm[row - BT] = acc;
    }
}

// ------------------------------------------------------------------------------------------ forward
struct AttFwdArgs {
    const char* side_p;     // planes of the lane-side S operand (dropped copy * w_tm)   + inverse scales (B, pad32(N))
    const float* side_i;
    const char* mS;         // planes of the streamed S operand (dropped copy)      + inverse scales (B, pad32(R))
    const float* iS;
    const char* mV0;        // planes of the first value tensor (== mS without dropout)
    const float* iV0;
    const char* mV1;        // planes of the second value tensor (row pass only)
    const float* iV1;
    const uint8_t* m_mask;  // (B,R) or null when m_len is given
    const int* m_len;       // (B) prefix lengths or null
    const float* m_term;    // (B,R)
    const float* n_term;    // (B,N)
    float* stat;            // (B,N,2) {max,sum}                     (row pass)
    float* part_o;          // (B,splits,N,D) unnormalised partials  (col pass)
    float* part_stat;       // (B,splits,N,2)
    const float* text;      // (B,N,D)                               (row pass epilogue)
    float* out;             // (B,N,4D)
    float* bsave;           // (B,N,D)
    int N, R, D, B, splits, rows_per_split;
    int nstage;             // LDS stages of the panel loop (1 or 2)
    int dbg;
};

// NV = 1: column pass = att_col_kernel (lane side = modality rows j, streams text rows i), produces the partials of q.
// NV = 2: row pass = att_row_kernel    (lane side = text rows i, streams modality rows j with values [mod | q]), produces out.
template <int NV, bool DBG>
__device__ __forceinline__ void att_fwd_body(const AttFwdArgs& a, char* smem) {
    const int dbg = DBG ? a.dbg : 0;   // timing-only ablations are compiled into their own kernel instances
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int N = a.N, R = a.R, D = a.D, Rp = pad32(R);
    int tile, split, b;
    decode_block((N + 16 * NW - 1) / (16 * NW), a.splits, a.B, tile, split, b);
    const int n = (tile * NW + wave) * 16 + r;
    const int rps = a.rows_per_split;

    const bool sep_s = a.mS != a.mV0;  // dropped copy differs from the clean value panel
    // LDS: nstage stages of [V0 | V1 (NV == 2) | S (dropped copy)] panels, then the per-row scalars of the WHOLE split
    const int npan = NV + (sep_s ? 1 : 0);
    const int stage_b = npan * PANEL_B;
    const bool db = a.nstage == 2;     // two stages: the LDS-DMA of panel p+1 is in flight under the MFMAs of panel p
    float* mterm_all = reinterpret_cast<float*>(smem + a.nstage * stage_b);
    int* mcode_all = reinterpret_cast<int*>(mterm_all + rps);
    float* sS_all = mterm_all + 2 * rps;              // inverse row scale of the S operand
    float* sV0_all = mterm_all + 3 * rps;             // inverse row scale of V0 times c0
    float* sV1_all = mterm_all + 4 * rps;
    float* red = mterm_all + 5 * rps;

    const size_t szR = planes_sample_bytes(R);
    const char* mS_b = a.mS + (size_t)b * szR;
    const char* mV0_b = a.mV0 + (size_t)b * szR;
    const char* mV1_b = NV == 2 ? a.mV1 + (size_t)b * szR : nullptr;

    side_t side;
    float inv_n;
    load_side_planes(side, inv_n, a.side_p + (size_t)b * planes_sample_bytes(N), a.side_i + (size_t)b * pad32(N), n, N, g);
    const float nterm = n < N ? a.n_term[(size_t)b * N + n] : 0.f;
    const tr_off tr = make_tr_off(lane);

    // row pass: the workgroup's 64 text rows, one row per wave-instruction (lane = 16-B chunk), are read ONCE into
    // registers (16 independent loads in flight), copied out at once as the first quarter of `out` (a verbatim copy of
    // text, attention.py:52 -- these stores overlap the main loop) and kept for the epilogue's text*a, text*b
    f4 trow[16];
    if (NV == 2) {
        const float* tx = a.text + (size_t)b * N * D;
        float* oo = a.out + (size_t)b * N * 4 * D;
        const bool cin = 4 * lane < D;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int gn = tile * NW * 16 + wave + NW * k;
            trow[k] = (gn < N && cin) ? *reinterpret_cast<const f4*>(tx + (size_t)gn * D + 4 * lane) : f4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int gn = tile * NW * 16 + wave + NW * k;
            if (gn < N && cin) *reinterpret_cast<f4*>(oo + (size_t)gn * 4 * D + 4 * lane) = trow[k];
        }
    }
    acc_t O0, O1;
    zero_acc(O0);
    zero_acc(O1);
    float m_run = -INFINITY, l_run = 0.f;

    const int row_begin = split * rps;
    const int row_end = (dbg & 16) ? row_begin : min(R, row_begin + rps);
    float im[2] = {0.f, 0.f};
    for (int i = tid; i < rps; i += NTHR) {
        const int m = row_begin + i;
        const bool in = m < min(R, row_begin + rps);
        mterm_all[i] = in ? a.m_term[(size_t)b * R + m] : 0.f;
        mcode_all[i] = mask_code(in, a.m_mask, a.m_len, b, R, m);
        sS_all[i] = in ? a.iS[(size_t)b * Rp + m] : 0.f;
        const float v0 = in ? a.iV0[(size_t)b * Rp + m] : 0.f;
        sV0_all[i] = v0;
        im[0] = fmaxf(im[0], v0);
        if (NV == 2) {
            const float v1 = in ? a.iV1[(size_t)b * Rp + m] : 0.f;
            sV1_all[i] = v1;
            im[1] = fmaxf(im[1], v1);
        }
    }
    wg_allmax(im, red, tid);
    const float c0 = cmap(im[0]), c1 = cmap(im[1]);
    for (int i = tid; i < rps; i += NTHR) {   // each thread rescales the entries it wrote itself
        sV0_all[i] *= c0;
        if (NV == 2) sV1_all[i] *= c1;
    }

    auto stage = [&](char* base, int p0) {
        if ((dbg & 1) && p0 != row_begin) return;
        stage_panel(base, mV0_b, p0, tid);
        if (NV == 2) stage_panel(base + PANEL_B, mV1_b, p0, tid);
        if (sep_s) stage_panel(base + NV * PANEL_B, mS_b, p0, tid);
    };
    int cur = 0;
    if (db) {
        if (row_begin < row_end) stage(smem, row_begin);
        __syncthreads();
    }
    for (int p0 = row_begin; p0 < row_end; p0 += PR) {
        char* base = smem + (db ? cur * stage_b : 0);
        if (db) {
            if (p0 + PR < row_end) stage(smem + (cur ^ 1) * stage_b, p0 + PR);
        } else {
            __syncthreads();
            stage(base, p0);
            __syncthreads();
        }
        const char* pV0 = base;
        const char* pV1 = base + PANEL_B;                 // only touched when NV == 2
        const char* pS = sep_s ? base + NV * PANEL_B : pV0;
        const int i0 = p0 - row_begin;

        f4 v[2] = {f4{0.f, 0.f, 0.f, 0.f}, f4{0.f, 0.f, 0.f, 0.f}};
        if (!(dbg & 2)) sprod2(pS, r, g, side, v);
        float bmax = -INFINITY;
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int ml = i0 + mb * 16 + 4 * g + e;
                const int code = mcode_all[ml];
                const float x = v[mb][e] * (sS_all[ml] * inv_n) + mterm_all[ml] + nterm;
                v[mb][e] = code == 2 ? x : (code == 1 ? NEG : -INFINITY);
                bmax = fmaxf(bmax, v[mb][e]);
            }
        bmax = kg_allmax(bmax);
        const float m_new = fmaxf(m_run, bmax);
        const float alpha = __expf(m_run - m_new);
        float psum = 0.f;
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
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
        {
            f4 w[2];
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int e = 0; e < 4; ++e) w[mb][e] = v[mb][e] * sV0_all[i0 + mb * 16 + 4 * g + e];
            half8 W0, W1;
            split_w(w[0], w[1], W0, W1);
            if (!(dbg & 4)) pvprod(pV0, tr, W0, W1, O0);
        }
        if (NV == 2) {
            f4 w[2];
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int e = 0; e < 4; ++e) w[mb][e] = v[mb][e] * sV1_all[i0 + mb * 16 + 4 * g + e];
            half8 W0, W1;
            split_w(w[0], w[1], W0, W1);
            if (!(dbg & 4)) pvprod(pV1, tr, W0, W1, O1);
        }
        if (db) {
            __syncthreads();   // retires the DMA of the next stage and frees this one
            cur ^= 1;
        }
    }

    // ---- epilogue.  The accumulators hold 16 rows x 64-B pieces per store instruction; written directly that is 16
    // partial cache lines per instruction.  Instead each wave parks its tile in LDS (the panels are dead) and the
    // workgroup writes whole rows: one row per wave-instruction, lane = 16-B chunk.
    const float l = kg_allsum(l_run);
    const bool partial = NV == 1;
    if (n < N && g == 0) {
        float* st = partial ? a.part_stat + (((size_t)b * a.splits + split) * N + n) * 2 : a.stat + ((size_t)b * N + n) * 2;
        st[0] = m_run;
        st[1] = l;
    }
    if (dbg & 8) return;
    float* et = reinterpret_cast<float*>(smem);          // [16*NW][LDP]
    const int row0 = tile * NW * 16;                    // first lane-side row of this workgroup
    const int c4 = lane;                                // this lane's 16-B chunk of a row
    auto park = [&](const acc_t& O, float scale) {
        __syncthreads();
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
            *reinterpret_cast<f4*>(et + (wave * 16 + r) * LDP + 16 * dt + 4 * g) = O[dt] * scale;
        __syncthreads();
    };
    park(O0, partial ? 1.0f / c0 : 1.0f / (l * c0));
    if (NV == 1) {
        float* dst = a.part_o + ((size_t)b * a.splits + split) * N * D;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int rr = wave + NW * k;
            const int gn = row0 + rr;
            if (gn < N && 4 * c4 < D) *reinterpret_cast<f4*>(dst + (size_t)gn * D + 4 * c4) = *reinterpret_cast<const f4*>(et + rr * LDP + 4 * c4);
        }
    } else {
        float* oo = a.out + (size_t)b * N * 4 * D;
        float* bo = a.bsave + (size_t)b * N * D;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int rr = wave + NW * k, gn = row0 + rr;
            if (gn < N && 4 * c4 < D) {
                const f4 av = *reinterpret_cast<const f4*>(et + rr * LDP + 4 * c4);
                float* o = oo + (size_t)gn * 4 * D + 4 * c4;
                *reinterpret_cast<f4*>(o + D) = av;
                *reinterpret_cast<f4*>(o + 2 * D) = trow[k] * av;
            }
        }
        park(O1, 1.0f / (l * c1));
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int rr = wave + NW * k, gn = row0 + rr;
            if (gn < N && 4 * c4 < D) {
                const f4 bv = *reinterpret_cast<const f4*>(et + rr * LDP + 4 * c4);
                *reinterpret_cast<f4*>(oo + (size_t)gn * 4 * D + 3 * D + 4 * c4) = trow[k] * bv;
                *reinterpret_cast<f4*>(bo + (size_t)gn * D + 4 * c4) = bv;
            }
        }
    }
}

template <bool DBG>
__global__ __launch_bounds__(NTHR) void att_col_kernel(const AttFwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    att_fwd_body<1, DBG>(a, smem);
}
template <bool DBG>
__global__ __launch_bounds__(NTHR) void att_row_kernel(const AttFwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    att_fwd_body<2, DBG>(a, smem);
}

// Row pass with TWO waves per SIMD (8 waves per workgroup): waves w and w+4 own the same 16 text rows and split the
// two value products between them -- role 0 accumulates a = P1.mod, role 1 accumulates b = P1.q -- so each wave needs
// one accumulator set (<= 256 registers) and the pair hides each other's LDS / softmax / barrier latencies; both compute
// S and the online softmax of the panel (42 of the 81 MFMAs a wave issues per panel are redundant: the price).
template <bool DBG>
__global__ __launch_bounds__(2 * NTHR) void att_row8_kernel(const AttFwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int dbg = DBG ? a.dbg : 0;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int role = __builtin_amdgcn_readfirstlane(wave >> 2), w4 = wave & 3;
    const int r = lane & 15, g = lane >> 4;
    const int N = a.N, R = a.R, D = a.D, Rp = pad32(R);
    int tile, split, b;
    decode_block((N + 16 * NW - 1) / (16 * NW), 1, a.B, tile, split, b);
    const int n = (tile * NW + w4) * 16 + r;
    const int rps = a.rows_per_split;   // = pad32(R): one split

    const bool sep_s = a.mS != a.mV0;
    const int npan = 2 + (sep_s ? 1 : 0);
    const int stage_b = npan * PANEL_B;
    const bool db = a.nstage == 2;
    float* mterm_all = reinterpret_cast<float*>(smem + a.nstage * stage_b);
    int* mcode_all = reinterpret_cast<int*>(mterm_all + rps);
    float* sS_all = mterm_all + 2 * rps;
    float* sV_all[2] = {mterm_all + 3 * rps, mterm_all + 4 * rps};
    float* red = mterm_all + 5 * rps;

    const size_t szR = planes_sample_bytes(R);
    const char* mS_b = a.mS + (size_t)b * szR;
    const char* mV0_b = a.mV0 + (size_t)b * szR;
    const char* mV1_b = a.mV1 + (size_t)b * szR;

    side_t side;
    float inv_n;
    load_side_planes(side, inv_n, a.side_p + (size_t)b * planes_sample_bytes(N), a.side_i + (size_t)b * pad32(N), n, N, g);
    const float nterm = n < N ? a.n_term[(size_t)b * N + n] : 0.f;
    const tr_off tr = make_tr_off(lane);

    // first quarter of `out` = verbatim copy of text (attention.py:52): 8 rows per wave, loads batched, stores at once
    {
        const float* tx = a.text + (size_t)b * N * D;
        float* oo = a.out + (size_t)b * N * 4 * D;
        const bool cin = 4 * lane < D;
        f4 t[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int gn = tile * NW * 16 + wave + 8 * k;
            t[k] = (gn < N && cin) ? *reinterpret_cast<const f4*>(tx + (size_t)gn * D + 4 * lane) : f4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int gn = tile * NW * 16 + wave + 8 * k;
            if (gn < N && cin) *reinterpret_cast<f4*>(oo + (size_t)gn * 4 * D + 4 * lane) = t[k];
        }
    }
    acc_t O;
    zero_acc(O);
    float m_run = -INFINITY, l_run = 0.f;

    const int row_end = (dbg & 16) ? 0 : R;
    float im[2] = {0.f, 0.f};
    for (int i = tid; i < rps; i += 2 * NTHR) {
        const bool in = i < R;
        mterm_all[i] = in ? a.m_term[(size_t)b * R + i] : 0.f;
        mcode_all[i] = mask_code(in, a.m_mask, a.m_len, b, R, i);
        sS_all[i] = in ? a.iS[(size_t)b * Rp + i] : 0.f;
        const float v0 = in ? a.iV0[(size_t)b * Rp + i] : 0.f, v1 = in ? a.iV1[(size_t)b * Rp + i] : 0.f;
        sV_all[0][i] = v0;
        sV_all[1][i] = v1;
        im[0] = fmaxf(im[0], v0);
        im[1] = fmaxf(im[1], v1);
    }
    {   // workgroup maximum over the 8 waves
#pragma unroll
        for (int k = 0; k < 2; ++k) im[k] = wave_allmax(im[k]);
        __syncthreads();
        if (lane == 0) { red[wave * 2] = im[0]; red[wave * 2 + 1] = im[1]; }
        __syncthreads();
        im[0] = im[1] = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) { im[0] = fmaxf(im[0], red[w * 2]); im[1] = fmaxf(im[1], red[w * 2 + 1]); }
    }
    const float cv[2] = {cmap(im[0]), cmap(im[1])};
    for (int i = tid; i < rps; i += 2 * NTHR) {
        sV_all[0][i] *= cv[0];
        sV_all[1][i] *= cv[1];
    }
    const float c_mine = role ? cv[1] : cv[0];
    const float* sV_mine = role ? sV_all[1] : sV_all[0];

    // staging: waves 0-3 bring the mod panel, waves 4-7 the q panel (7 pieces each); a separate S panel: all 8 waves
    auto stage = [&](char* base, int p0) {
        if ((dbg & 1) && p0 != 0) return;
        stage_panel(base + role * PANEL_B, role ? mV1_b : mV0_b, p0, tid & (NTHR - 1));
        if (sep_s) {
            const char* src = mS_b + (size_t)(p0 >> 4) * PRB + lane * 16;
            char* dst = base + 2 * PANEL_B;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int piece = wave + 8 * k;
                if (piece < PANEL_B / 1024)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + piece * 1024),
                                                     (__attribute__((address_space(3))) void*)(dst + piece * 1024), 16, 0, 0);
            }
        }
    };
    int cur = 0;
    if (db) {
        if (0 < row_end) stage(smem, 0);
        __syncthreads();
    }
    for (int p0 = 0; p0 < row_end; p0 += PR) {
        char* base = smem + (db ? cur * stage_b : 0);
        if (db) {
            if (p0 + PR < row_end) stage(smem + (cur ^ 1) * stage_b, p0 + PR);
        } else {
            __syncthreads();
            stage(base, p0);
            __syncthreads();
        }
        const char* pS = sep_s ? base + 2 * PANEL_B : base;
        const char* pV = base + role * PANEL_B;

        f4 v[2] = {f4{0.f, 0.f, 0.f, 0.f}, f4{0.f, 0.f, 0.f, 0.f}};
        if (!(dbg & 2)) sprod2(pS, r, g, side, v);
        float bmax = -INFINITY;
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int ml = p0 + mb * 16 + 4 * g + e;
                const int code = mcode_all[ml];
                const float x = v[mb][e] * (sS_all[ml] * inv_n) + mterm_all[ml] + nterm;
                v[mb][e] = code == 2 ? x : (code == 1 ? NEG : -INFINITY);
                bmax = fmaxf(bmax, v[mb][e]);
            }
        bmax = kg_allmax(bmax);
        const float m_new = fmaxf(m_run, bmax);
        const float alpha = __expf(m_run - m_new);
        float psum = 0.f;
        f4 w[2];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float pv = __expf(v[mb][e] - m_new);
                psum += pv;
                w[mb][e] = pv * sV_mine[p0 + mb * 16 + 4 * g + e];
            }
        l_run = l_run * alpha + psum;
        if (__any(alpha != 1.0f)) {
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) O[dt] *= alpha;
        }
        m_run = m_new;
        half8 W0, W1;
        split_w(w[0], w[1], W0, W1);
        if (!(dbg & 4)) pvprod(pV, tr, W0, W1, O);
        if (db) {
            __syncthreads();
            cur ^= 1;
        }
    }

    // ---- epilogue: every wave parks its tile (role 0: a, role 1: b) in its role's LDS region, then the workgroup writes
    // whole rows: role 0 the a and text*a quarters, role 1 the text*b quarter and the saved b
    const float l = kg_allsum(l_run);
    if (role == 0 && n < N && g == 0) {
        float* st = a.stat + ((size_t)b * N + n) * 2;
        st[0] = m_run;
        st[1] = l;
    }
    if (dbg & 8) return;
    float* et = reinterpret_cast<float*>(smem) + role * (16 * NW * LDP);
    const int row0 = tile * NW * 16;
    const int c4 = lane;
    const float* tx = a.text + (size_t)b * N * D;
    f4 trow[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int gn = row0 + w4 + NW * k;
        trow[k] = (gn < N && 4 * c4 < D) ? *reinterpret_cast<const f4*>(tx + (size_t)gn * D + 4 * c4) : f4{0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();
    {
        const float scale = 1.0f / (l * c_mine);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) *reinterpret_cast<f4*>(et + (w4 * 16 + r) * LDP + 16 * dt + 4 * g) = O[dt] * scale;
    }
    __syncthreads();
    float* oo = a.out + (size_t)b * N * 4 * D;
    float* bo = a.bsave + (size_t)b * N * D;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int rr = w4 + NW * k, gn = row0 + rr;
        if (gn < N && 4 * c4 < D) {
            const f4 ov = *reinterpret_cast<const f4*>(et + rr * LDP + 4 * c4);
            float* o = oo + (size_t)gn * 4 * D + 4 * c4;
            if (role == 0) {
                *reinterpret_cast<f4*>(o + D) = ov;
                *reinterpret_cast<f4*>(o + 2 * D) = trow[k] * ov;
            } else {
                *reinterpret_cast<f4*>(o + 3 * D) = trow[k] * ov;
                *reinterpret_cast<f4*>(bo + (size_t)gn * D + 4 * c4) = ov;
            }
        }
    }
}

// merge the per-split partial column softmaxes, q = sum_p O_p e^{m_p-m} / sum_p l_p e^{m_p-m}, and write q as planes
// (it is only ever a streamed / lane-side MFMA operand): one wave per modality row, lane = float4 chunk
__global__ __launch_bounds__(256) void att_combine_kernel(const float* __restrict__ part_o, const float* __restrict__ part_stat,
                                                          char* __restrict__ q_planes, float* __restrict__ q_inv, float* __restrict__ stat,
                                                          int B, int N, int D, int splits) {
    const int Np = pad32(N);
    const long rowi = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (rowi >= (long)B * Np) return;
    const int b = rowi / Np, n = rowi - (long)b * Np, c = threadIdx.x & 63, d = 4 * c;
    f4 v = f4{0.f, 0.f, 0.f, 0.f};
    if (n < N) {
        // four splits per trip: their statistics and partial rows are loaded together (independent of the running values),
        // lanes beyond D read column 0 and drop it -- no branch around the loads
        const int dsafe = d < D ? d : 0;
        float m = -INFINITY, l = 0.f;
        for (int p0 = 0; p0 < splits; p0 += 4) {
            float pm[4], pl[4];
            f4 po[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const size_t o = ((size_t)b * splits + min(p0 + k, splits - 1)) * N + n;
                pm[k] = part_stat[o * 2];
                pl[k] = part_stat[o * 2 + 1];
                po[k] = *reinterpret_cast<const f4*>(part_o + o * D + dsafe);
            }
            float mn = m;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (p0 + k < splits) mn = fmaxf(mn, pm[k]);
            const float resc = __expf(m - mn);   // exp(-inf) = 0 on the first trip
            l *= resc;
            v = v * resc;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (p0 + k < splits) {
                    const float sc = __expf(pm[k] - mn);
                    l += pl[k] * sc;
                    v += po[k] * sc;
                }
            m = mn;
        }
        if (d >= D) v = f4{0.f, 0.f, 0.f, 0.f};
        v = v * (1.0f / l);
        if (c == 0) {
            stat[((size_t)b * N + n) * 2] = m;
            stat[((size_t)b * N + n) * 2 + 1] = l;
        }
    }
    const float amax = wave_allmax(f4amax(v));
    store_split_row(q_planes + (size_t)b * planes_sample_bytes(N), q_inv + (size_t)b * Np + n, n, c, v, amax);
}

// ------------------------------------------------------------------------------------------ backward
// prologue over text rows (one wave per row, lane = float4 chunk: every access a coalesced 16 B per lane):
//   da = g1 + g2*text ; db = g3*text  (written as planes: they are only ever MFMA operands)
//   delta1 = da.a + db.b ; d_text = g0 + g2*a + g3*b      (a = out[:, D:2D], b = bsave)
__global__ __launch_bounds__(256) void att_bwd_pre_kernel(const float* __restrict__ d_out, const float* __restrict__ out,
                                                          const float* __restrict__ text, const float* __restrict__ bsave,
                                                          char* __restrict__ da_planes, float* __restrict__ da_inv,
                                                          char* __restrict__ db_planes, float* __restrict__ db_inv,
                                                          float* __restrict__ delta1, float* __restrict__ d_text,
                                                          float* __restrict__ d_w_t, float* __restrict__ d_w_m,
                                                          float* __restrict__ d_w_tm, float* __restrict__ d_bias, int B, int T, int D) {
    if (blockIdx.x == 0) {  // the parameter gradients are accumulated with atomics by the later kernels
        for (int i = threadIdx.x; i < D; i += 256) d_w_t[i] = d_w_m[i] = d_w_tm[i] = 0.f;
        if (threadIdx.x == 0) d_bias[0] = 0.f;
    }
    const int Tp = pad32(T);
    const long rowi = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (rowi >= (long)B * Tp) return;
    const int b = rowi / Tp, row = rowi - (long)b * Tp, c = threadIdx.x & 63, d = 4 * c;
    f4 xa = f4{0.f, 0.f, 0.f, 0.f}, xb = xa;
    float acc = 0.f;
    if (row < T && d < D) {
        const size_t rr = (size_t)b * T + row;
        const float* g = d_out + rr * 4 * D;
        const f4 g0 = *reinterpret_cast<const f4*>(g + d), g1 = *reinterpret_cast<const f4*>(g + D + d);
        const f4 g2 = *reinterpret_cast<const f4*>(g + 2 * D + d), g3 = *reinterpret_cast<const f4*>(g + 3 * D + d);
        const f4 av = *reinterpret_cast<const f4*>(out + rr * 4 * D + D + d);
        const f4 t = *reinterpret_cast<const f4*>(text + rr * D + d);
        const f4 bv = *reinterpret_cast<const f4*>(bsave + rr * D + d);
        xa = g1 + g2 * t;
        xb = g3 * t;
        *reinterpret_cast<f4*>(d_text + rr * D + d) = g0 + g2 * av + g3 * bv;
        acc = f4sum(xa * av + xb * bv);
    }
    const float amax_a = wave_allmax(f4amax(xa)), amax_b = wave_allmax(f4amax(xb));
    acc = wave_allsum(acc);
    if (c == 0 && row < T) delta1[(size_t)b * T + row] = acc;
    const size_t sz = planes_sample_bytes(T);
    store_split_row(da_planes + (size_t)b * sz, da_inv + (size_t)b * Tp + row, row, c, xa, amax_a);
    store_split_row(db_planes + (size_t)b * sz, db_inv + (size_t)b * Tp + row, row, c, xb, amax_b);
}

struct AttBwdArgs {
    const float *text, *mod, *text_d, *mod_d;       // (B,T,D) / (B,M,D) fp32 (lane-side loads)
    const uint8_t *text_mask, *mod_mask;            // (B,T) / (B,M) or null with the lengths
    const int *text_len, *mod_len;                  // (B) or null
    const float *w_t, *w_m, *w_tm;
    const float *rterm, *cterm, *row_stat, *col_stat;
    // planes + inverse scales (saved by the forward: text, text_d, mod, mod_d, q; workspace: da, db, dq)
    const char *pT, *pTd, *pM, *pMd, *pQ, *pDa, *pDb, *pTw, *pMw;   // pTw / pMw: text_d * w_tm, mod_d * w_tm (lane-side S operands)
    const float *iT, *iTd, *iM, *iMd, *iQ, *iDa, *iDb, *iTw, *iMw;
    char* pDq;
    float* iDq;
    const float* delta1;                             // (B,T)
    float* delta2;                                   // (B,M)
    float *d_mod, *d_mod_d, *d_text, *d_text_d;      // outputs
    float *d_w_t, *d_w_m, *d_w_tm, *d_bias;          // outputs, zeroed by the prologue, accumulated with atomics
    // per-split partial sums of the j-side sweeps, (B,splits,M,D) / (B,splits,M)
    float *p_dq, *p_dmc, *p_dmd1, *p_dmd2, *p_dc1, *p_dc2;
    int B, T, M, D, splits, rows_per_split;
    int fold;                                        // 1: no dropped copies, d_*_d folded into d_*
    int nstage_j2;                                   // LDS stages of the second j sweep (1 or 2)
    int jf_rows;                                     // rows per wave of the j-side epilogue (1, 2, 4 or 8)
    int i_blocks;                                    // workgroups of the i-side pass; ids beyond them run the j-side epilogue
    int dbg;
};

// Accumulator tiles hold 16 rows x 64-B pieces per store instruction (16 partial cache lines); instead every wave parks
// its tile in LDS and the workgroup stores its 64 rows whole: one row per wave-instruction, lane = 16-B chunk.
// et: [64][LDP] floats of LDS that no wave still reads (the panels are dead); dst: row `row0` of a (rows, D) matrix.
__device__ __forceinline__ void park_store(float* et, const acc_t& O, float scale, float* dst, int nrows, int D, int tid) {
    const int lane = tid & 63, wave = tid >> 6, r = lane & 15, g = lane >> 4;
    __syncthreads();
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) *reinterpret_cast<f4*>(et + (wave * 16 + r) * LDP + 16 * dt + 4 * g) = O[dt] * scale;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int rr = wave + NW * k;
        if (rr < nrows && 4 * lane < D) *reinterpret_cast<f4*>(dst + (size_t)rr * D + 4 * lane) = *reinterpret_cast<const f4*>(et + rr * LDP + 4 * lane);
    }
}
__device__ __forceinline__ void store_acc(float* dst_row, const acc_t& v, float scale, int D, int g) {
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
        const int d = 16 * dt + 4 * g;
        if (d < D) *reinterpret_cast<f4*>(dst_row + d) = v[dt] * scale;
    }
}

// j-side sweep 1 (lane side = modality rows j, streams a slice of the text rows i):
//   dq_j += sum_i P1_ij db_i ; dmodc_j += sum_i P1_ij da_i ; dS1 = P1 (dP1 - delta1_i) mask_j
//   dmodd_j += sum_i dS1_ij text_d_i (scaled by w_tm later) ; dc_j += sum_i dS1_ij
template <bool DBG>
__global__ __launch_bounds__(NTHR) void att_bwd_j1_kernel(const AttBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int dbg = DBG ? a.dbg : 0;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int T = a.T, M = a.M, D = a.D, Tp = pad32(T);
    int tile, split, b;
    decode_block((M + 16 * NW - 1) / (16 * NW), a.splits, a.B, tile, split, b);
    const int n = (tile * NW + wave) * 16 + r;  // modality row j
    const int rps = a.rows_per_split;

    char* pTd = smem;
    char* pDa = smem + PANEL_B;
    char* pDb = smem + 2 * PANEL_B;
    float* rt_all = reinterpret_cast<float*>(smem + 3 * PANEL_B);
    float* rmax_all = rt_all + rps;
    float* rinv_all = rt_all + 2 * rps;    // 1/rowsum, 0 beyond the slice
    float* dl1_all = rt_all + 3 * rps;
    float* sTd_all = rt_all + 4 * rps;
    float* sDa_all = rt_all + 5 * rps;
    float* sDb_all = rt_all + 6 * rps;
    float* red = rt_all + 7 * rps;

    const size_t szT = planes_sample_bytes(T), szM = planes_sample_bytes(M);
    side_t sideS, sideM, sideQ;
    float inS, inM, inQ;
    load_side_planes(sideS, inS, a.pMw + (size_t)b * szM, a.iMw + (size_t)b * pad32(M), n, M, g);
    load_side_planes(sideM, inM, a.pM + (size_t)b * szM, a.iM + (size_t)b * pad32(M), n, M, g);
    load_side_planes(sideQ, inQ, a.pQ + (size_t)b * szM, a.iQ + (size_t)b * pad32(M), n, M, g);
    const bool nin = n < M;
    const float cterm = nin ? a.cterm[(size_t)b * M + n] : 0.f;
    const bool mm = nin ? mask_live(a.mod_mask, a.mod_len, b, M, n) : false;
    const float mmf = mm ? 1.f : 0.f;
    const tr_off tr = make_tr_off(lane);

    const int row_begin = split * rps, row_end = (dbg & 16) ? row_begin : min(T, row_begin + rps);
    float im[3] = {0.f, 0.f, 0.f};
    for (int i = tid; i < rps; i += NTHR) {
        const int t = row_begin + i;
        const bool in = t < min(T, row_begin + rps);
        const size_t bt = (size_t)b * T + t;
        rt_all[i] = in ? a.rterm[bt] : 0.f;
        rmax_all[i] = in ? a.row_stat[bt * 2] : INFINITY;   // exp(x - inf) = 0 beyond the slice
        rinv_all[i] = in ? 1.0f / a.row_stat[bt * 2 + 1] : 0.f;
        dl1_all[i] = in ? a.delta1[bt] : 0.f;
        const float v0 = in ? a.iTd[(size_t)b * Tp + t] : 0.f, v1 = in ? a.iDa[(size_t)b * Tp + t] : 0.f, v2 = in ? a.iDb[(size_t)b * Tp + t] : 0.f;
        sTd_all[i] = v0; sDa_all[i] = v1; sDb_all[i] = v2;
        im[0] = fmaxf(im[0], v0); im[1] = fmaxf(im[1], v1); im[2] = fmaxf(im[2], v2);
    }
    wg_allmax(im, red, tid);
    const float cDa = cmap(im[1]), cDb = cmap(im[2]);
    // |dS1_ij| <= |dP1_ij| + |delta1_i| <= 2 D 2^28 (inv_da_i inv_mod_j + inv_db_i inv_q_j)   (row maxima < 2^14 inv)
    const float cS = cmap_bound(im[0], 1.3743895e11f /* 2^37 */ * (im[1] * inM + im[2] * inQ));

    acc_t dq, dmc, dmd;
    zero_acc(dq);
    zero_acc(dmc);
    zero_acc(dmd);
    float dc = 0.f;

    const char* td_b = a.pTd + (size_t)b * szT;
    const char* da_b = a.pDa + (size_t)b * szT;
    const char* db_b = a.pDb + (size_t)b * szT;
    for (int p0 = row_begin; p0 < row_end; p0 += PR) {
        __syncthreads();
        if (!(dbg & 1) || p0 == row_begin) {
            stage_panel(pTd, td_b, p0, tid);
            stage_panel(pDa, da_b, p0, tid);
            stage_panel(pDb, db_b, p0, tid);
        }
        __syncthreads();
        const int i0 = p0 - row_begin;
        f4 s[2], dpa[2], dpb[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) s[q] = dpa[q] = dpb[q] = f4{0.f, 0.f, 0.f, 0.f};
        if (!(dbg & 2)) {
            sprod2<false>(pTd, r, g, sideS, s);
            sprod2<false>(pDa, r, g, sideM, dpa);
            sprod2<false>(pDb, r, g, sideQ, dpb);
        }
        f4 wq[2], wc[2], wd[2];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int ml = i0 + mb * 16 + 4 * g + e;
                const float x = mm ? s[mb][e] * (sTd_all[ml] * inS) + rt_all[ml] + cterm : NEG;
                const float p = __expf(x - rmax_all[ml]) * rinv_all[ml];  // rinv = 0 beyond the slice
                const float dp = dpa[mb][e] * (sDa_all[ml] * inM) + dpb[mb][e] * (sDb_all[ml] * inQ);
                const float gd = p * (dp - dl1_all[ml]) * mmf;
                dc += gd;
                wq[mb][e] = p * (sDb_all[ml] * cDb);
                wc[mb][e] = p * (sDa_all[ml] * cDa);
                wd[mb][e] = gd * (sTd_all[ml] * cS);
            }
        half8 W0, W1;
        split_w(wq[0], wq[1], W0, W1);
        if (!(dbg & 4)) pvprod<1>(pDb, tr, W0, W1, dq);
        split_w(wc[0], wc[1], W0, W1);
        if (!(dbg & 4)) pvprod<1>(pDa, tr, W0, W1, dmc);
        split_w(wd[0], wd[1], W0, W1);
        if (!(dbg & 4)) pvprod<1>(pTd, tr, W0, W1, dmd);
    }
    dc = kg_allsum(dc);
    if (dbg & 8) return;
    {
        float* et = reinterpret_cast<float*>(smem);
        const int row0 = tile * NW * 16;
        const size_t prow0 = ((size_t)b * a.splits + split) * M + row0;
        park_store(et, dq, 1.0f / cDb, a.p_dq + prow0 * D, M - row0, D, tid);
        park_store(et, dmc, 1.0f / cDa, a.p_dmc + prow0 * D, M - row0, D, tid);
        park_store(et, dmd, 1.0f / cS, a.p_dmd1 + prow0 * D, M - row0, D, tid);
        if (nin && g == 0) a.p_dc1[prow0 + (n - row0)] = dc;
    }
}

// j-side sweep 2 (needs the complete dq = sum of the sweep-1 partials):
//   dS2 = P2 (dP2 - delta2_j) mask_i, dP2_ij = text_i . dq_j ; dmodd_j += sum_i dS2_ij text_d_i ; dc_j += sum_i dS2_ij
//   split 0 also publishes dq_j (as planes) and delta2_j = q_j . dq_j for the i-side pass
template <bool DBG>
__global__ __launch_bounds__(NTHR) void att_bwd_j2_kernel(const AttBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int dbg = DBG ? a.dbg : 0;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int T = a.T, M = a.M, D = a.D, Tp = pad32(T), Mp = pad32(M);
    int tile, split, b;
    decode_block((M + 16 * NW - 1) / (16 * NW), a.splits, a.B, tile, split, b);
    const int n = (tile * NW + wave) * 16 + r;
    const int rps = a.rows_per_split;

    const bool sep = a.pTd != a.pT;
    const int stage_b = (sep ? 2 : 1) * PANEL_B;
    const bool db = a.nstage_j2 == 2;
    float* rt_all = reinterpret_cast<float*>(smem + a.nstage_j2 * stage_b);
    int* code_all = reinterpret_cast<int*>(rt_all + rps);   // 0 beyond slice, 1 masked, 2 live
    float* sT_all = rt_all + 2 * rps;
    float* sTd_all = rt_all + 3 * rps;
    float* red = rt_all + 4 * rps;

    const size_t szT = planes_sample_bytes(T), szM = planes_sample_bytes(M);
    side_t sideS, sideDq;
    float inS, inDq;
    load_side_planes(sideS, inS, a.pMw + (size_t)b * szM, a.iMw + (size_t)b * Mp, n, M, g);
    float delta2;
    {
        float x[KT][8];
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int j = 0; j < 8; ++j) x[kt][j] = 0.f;
        for (int p = 0; p < a.splits; ++p) {
            float t[KT][8];
            load_row_regs(t, a.p_dq + ((size_t)b * a.splits + p) * M * D, n, M, D, g, nullptr);
#pragma unroll
            for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                for (int j = 0; j < 8; ++j) x[kt][j] += t[kt][j];
        }
        side_t sq;
        float inQ;
        load_side_planes(sq, inQ, a.pQ + (size_t)b * szM, a.iQ + (size_t)b * Mp, n, M, g);
        delta2 = kg_allsum(side_dot_regs(sq, inQ, x));
        side_from_regs(x, sideDq, inDq);
    }
    const bool nin = n < M;
    if (split == 0 && n < Mp) {   // rows M..Mp-1 are written as zeros (their loads were guarded)
        char* dst = a.pDq + (size_t)b * szM;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            char* d = dst + pl_off_att(n, 4 * kt + g);
            *reinterpret_cast<half8*>(d) = sideDq.h[kt][0];
            *reinterpret_cast<half8*>(d + 1024) = sideDq.h[kt][1];
        }
        if (g == 0) {
            a.iDq[(size_t)b * Mp + n] = inDq;
            if (nin) a.delta2[(size_t)b * M + n] = delta2;
        }
    }
    const float cterm = nin ? a.cterm[(size_t)b * M + n] : 0.f;
    const float cmax = nin ? a.col_stat[((size_t)b * M + n) * 2] : 0.f;
    const float cinv = nin ? 1.0f / a.col_stat[((size_t)b * M + n) * 2 + 1] : 0.f;
    const tr_off tr = make_tr_off(lane);

    const int row_begin = split * rps, row_end = (dbg & 16) ? row_begin : min(T, row_begin + rps);
    float im[2] = {0.f, 0.f};
    for (int i = tid; i < rps; i += NTHR) {
        const int t = row_begin + i;
        const bool in = t < min(T, row_begin + rps);
        rt_all[i] = in ? a.rterm[(size_t)b * T + t] : 0.f;
        code_all[i] = mask_code(in, a.text_mask, a.text_len, b, T, t);
        const float v0 = in ? a.iT[(size_t)b * Tp + t] : 0.f, v1 = in ? a.iTd[(size_t)b * Tp + t] : 0.f;
        sT_all[i] = v0; sTd_all[i] = v1;
        im[0] = fmaxf(im[0], v0); im[1] = fmaxf(im[1], v1);
    }
    wg_allmax(im, red, tid);
    // |dS2_ij| <= |dP2_ij| + |delta2_j| <= 2 D 2^28 inv_t_i inv_dq_j
    const float cS = cmap_bound(im[1], 1.3743895e11f * (im[0] * inDq));

    acc_t dmd;
    zero_acc(dmd);
    float dc = 0.f;
    const char* td_b = a.pTd + (size_t)b * szT;
    const char* t_b = a.pT + (size_t)b * szT;
    auto stage = [&](char* base, int p0) {
        if ((dbg & 1) && p0 != row_begin) return;
        stage_panel(base, t_b, p0, tid);
        if (sep) stage_panel(base + PANEL_B, td_b, p0, tid);
    };
    int cur = 0;
    if (db) {
        if (row_begin < row_end) stage(smem, row_begin);
        __syncthreads();
    }
    for (int p0 = row_begin; p0 < row_end; p0 += PR) {
        char* base = smem + (db ? cur * stage_b : 0);
        if (db) {
            if (p0 + PR < row_end) stage(smem + (cur ^ 1) * stage_b, p0 + PR);
        } else {
            __syncthreads();
            stage(base, p0);
            __syncthreads();
        }
        const char* pT = base;
        const char* pTd = sep ? base + PANEL_B : pT;
        const int i0 = p0 - row_begin;
        f4 s[2], dp[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) s[q] = dp[q] = f4{0.f, 0.f, 0.f, 0.f};
        if (!(dbg & 2)) {
            sprod2(pTd, r, g, sideS, s);
            sprod2(pT, r, g, sideDq, dp);
        }
        f4 wd[2];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int ml = i0 + mb * 16 + 4 * g + e;
                const int code = code_all[ml];
                const float x = code == 2 ? s[mb][e] * (sTd_all[ml] * inS) + rt_all[ml] + cterm : NEG;
                const float p = code ? __expf(x - cmax) * cinv : 0.f;
                const float gd = code == 2 ? p * (dp[mb][e] * (sT_all[ml] * inDq) - delta2) : 0.f;
                dc += gd;
                wd[mb][e] = gd * (sTd_all[ml] * cS);
            }
        half8 W0, W1;
        split_w(wd[0], wd[1], W0, W1);
        if (!(dbg & 4)) pvprod(pTd, tr, W0, W1, dmd);
        if (db) {
            __syncthreads();
            cur ^= 1;
        }
    }
    dc = kg_allsum(dc);
    if (dbg & 8) return;
    {
        float* et = reinterpret_cast<float*>(smem);
        const int row0 = tile * NW * 16;
        const size_t prow0 = ((size_t)b * a.splits + split) * M + row0;
        park_store(et, dmd, 1.0f / cS, a.p_dmd2 + prow0 * D, M - row0, D, tid);
        if (nin && g == 0) a.p_dc2[prow0 + (n - row0)] = dc;
    }
}

// j-side epilogue: one wave per JF_ROWS modality rows, lane = 4 features.  Sums the split partials, writes
//   d_mod_d_j = dc_j w_m + w_tm * dmodd_j ;  d_mod_j = dmodc_j (+ d_mod_d_j when folded)
// and accumulates d_w_m += sum_j dc_j mod_d[j,:]: registers over the wave's rows, LDS across the 4 waves, then ONE
// atomic per feature and workgroup with consecutive lanes on consecutive addresses.
template <int JF_ROWS>
__device__ __forceinline__ void att_bwd_jfin_body(const AttBwdArgs& a, int B, int block, float (*wred)[256]) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int chunk = block * 4 + wave;
    const int M = a.M, D = a.D, S = a.splits;
    const int rows = B * M;
    const int d = lane * 4;
    const bool din = d < D;
    const f4 wm = din ? *reinterpret_cast<const f4*>(a.w_m + d) : f4{0.f, 0.f, 0.f, 0.f};
    const f4 wtm = din ? *reinterpret_cast<const f4*>(a.w_tm + d) : f4{0.f, 0.f, 0.f, 0.f};
    float dc[JF_ROWS];
    f4 c[JF_ROWS], dd[JF_ROWS];
    size_t base[JF_ROWS];
#pragma unroll
    for (int rr = 0; rr < JF_ROWS; ++rr) {
        const int row = min(chunk * JF_ROWS + rr, rows - 1);
        const int b = row / M, n = row % M;
        base[rr] = (size_t)b * S * M + n;
        dc[rr] = 0.f;
        c[rr] = dd[rr] = f4{0.f, 0.f, 0.f, 0.f};
    }
    const int dsafe = din ? d : 0;   // lanes beyond D read column 0 and drop it: no branch around the loads, so that all
                                     // 3 * JF_ROWS vector loads of a split are in flight together
    for (int p = 0; p < S; ++p) {
        f4 vc[JF_ROWS], v1[JF_ROWS], v2[JF_ROWS];
        float s1[JF_ROWS], s2[JF_ROWS];
#pragma unroll
        for (int rr = 0; rr < JF_ROWS; ++rr) {
            const size_t prow = base[rr] + (size_t)p * M;
            s1[rr] = a.p_dc1[prow];
            s2[rr] = a.p_dc2[prow];
            vc[rr] = *reinterpret_cast<const f4*>(a.p_dmc + prow * D + dsafe);
            v1[rr] = *reinterpret_cast<const f4*>(a.p_dmd1 + prow * D + dsafe);
            v2[rr] = *reinterpret_cast<const f4*>(a.p_dmd2 + prow * D + dsafe);
        }
#pragma unroll
        for (int rr = 0; rr < JF_ROWS; ++rr) {
            dc[rr] += s1[rr] + s2[rr];
            c[rr] += vc[rr];
            dd[rr] += v1[rr];
            dd[rr] += v2[rr];
        }
    }
    f4 wacc = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int rr = 0; rr < JF_ROWS; ++rr) {
        const int row = chunk * JF_ROWS + rr;
        if (row < rows && din) {
            const f4 gd = wm * dc[rr] + wtm * dd[rr];
            if (a.fold) {
                *reinterpret_cast<f4*>(a.d_mod + (size_t)row * D + d) = c[rr] + gd;
            } else {
                *reinterpret_cast<f4*>(a.d_mod + (size_t)row * D + d) = c[rr];
                *reinterpret_cast<f4*>(a.d_mod_d + (size_t)row * D + d) = gd;
            }
            wacc += *reinterpret_cast<const f4*>(a.mod_d + (size_t)row * D + d) * dc[rr];
        }
    }
    *reinterpret_cast<f4*>(&wred[wave][d]) = wacc;
    __syncthreads();
    const int t = threadIdx.x;
    if (t < D) atomicAdd(a.d_w_m + t, (wred[0][t] + wred[1][t]) + (wred[2][t] + wred[3][t]));
}
__device__ __forceinline__ void att_bwd_jfin_dispatch(const AttBwdArgs& a, int block, float (*wred)[256]) {
    switch (a.jf_rows) {
        case 8: att_bwd_jfin_body<8>(a, a.B, block, wred); break;
        case 4: att_bwd_jfin_body<4>(a, a.B, block, wred); break;
        case 2: att_bwd_jfin_body<2>(a, a.B, block, wred); break;
        default: att_bwd_jfin_body<1>(a, a.B, block, wred); break;
    }
}
// stand-alone launch of the j-side epilogue (timing tools, MMB_ATT_JFIN_SEPARATE=1); by default its workgroups ride in the
// launch of att_bwd_i_kernel, which leaves 32 of the 256 CUs idle and does not depend on them
__global__ __launch_bounds__(256) void att_bwd_jfin_kernel(const AttBwdArgs a) {
    __shared__ float wred[4][256];
    att_bwd_jfin_dispatch(a, blockIdx.x, wred);
}

// i-side pass (lane side = text rows i, streams all modality rows j):
//   dS = P1 (dP1 - delta1_i) mask_j + P2 (dP2 - delta2_j) mask_i
//   d_text_i += sum_j P2_ij dq_j ; dX_i = sum_j dS_ij mod_d_j ; dr_i = sum_j dS_ij
//   d_text_d_i = dr_i w_t + w_tm * dX_i ; d_w_t += dr_i text_d_i ; d_w_tm += dX_i * text_d_i ; d_bias += dr_i
// SAME = no dropped copies (text_d == text, mod_d == mod): the similarity is then formed as text . (mod * w_tm) -- lane side
// = the text planes the dP2 product needs anyway, streamed side = the w_tm-folded modality planes -- so the pass carries
// three lane-side operands (168 registers) instead of four and reads one 11.5-MB plane set less.
template <bool DBG, bool SAME>
__global__ __launch_bounds__(NTHR) void att_bwd_i_kernel(const AttBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int dbg = DBG ? a.dbg : 0;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int T = a.T, M = a.M, D = a.D, Mp = pad32(M);
    if ((int)blockIdx.x >= a.i_blocks) {
        // passenger workgroups: the j-side epilogue (sums of the split partials -> d_mod, d_mod_d, d_w_m).  It depends on the
        // two j sweeps only, like this pass, whose T/64 * B workgroups (224 at the metric configuration) leave CUs idle.
        att_bwd_jfin_dispatch(a, blockIdx.x - a.i_blocks, reinterpret_cast<float(*)[256]>(smem));
        return;
    }
    int tile, split, b;
    decode_block((T + 16 * NW - 1) / (16 * NW), 1, a.B, tile, split, b);
    const int n = (tile * NW + wave) * 16 + r;  // text row i

    const bool sep = a.pMd != a.pM;
    char* pM = smem;
    char* pQ = smem + PANEL_B;
    char* pDq = smem + 2 * PANEL_B;
    char* pMd = sep ? smem + 3 * PANEL_B : pM;
    char* pSp = SAME ? smem + 3 * PANEL_B : pMd;      // streamed S operand: mod * w_tm (SAME) or the dropped copy mod_d
    float* ct_all = reinterpret_cast<float*>(smem + 4 * PANEL_B);   // per-row scalars of ALL modality rows
    float* cmax_all = ct_all + Mp;
    float* cinv_all = ct_all + 2 * Mp;     // 0 beyond M
    float* dl2_all = ct_all + 3 * Mp;
    float* mmf_all = ct_all + 4 * Mp;      // modality mask as float, -1 beyond M
    float* sM_all = ct_all + 5 * Mp;
    float* sMd_all = ct_all + 6 * Mp;
    float* sQ_all = ct_all + 7 * Mp;
    float* sDq_all = ct_all + 8 * Mp;
    float* sSp_all = ct_all + 9 * Mp;      // inverse row scales of the streamed S operand
    float* red = ct_all + 10 * Mp;

    const size_t szT = planes_sample_bytes(T), szM = planes_sample_bytes(M);
    side_t sideS_, sideDa, sideDb, sideT;
    float inS_ = 0.f, inDa, inDb, inT;
    if (!SAME) load_side_planes(sideS_, inS_, a.pTw + (size_t)b * szT, a.iTw + (size_t)b * pad32(T), n, T, g);
    load_side_planes(sideDa, inDa, a.pDa + (size_t)b * szT, a.iDa + (size_t)b * pad32(T), n, T, g);
    load_side_planes(sideDb, inDb, a.pDb + (size_t)b * szT, a.iDb + (size_t)b * pad32(T), n, T, g);
    load_side_planes(sideT, inT, a.pT + (size_t)b * szT, a.iT + (size_t)b * pad32(T), n, T, g);
    const side_t& sideS = SAME ? sideT : sideS_;
    const float inS = SAME ? inT : inS_;
    const bool nin = n < T;
    const float rterm = nin ? a.rterm[(size_t)b * T + n] : 0.f;
    const float rmax = nin ? a.row_stat[((size_t)b * T + n) * 2] : 0.f;
    const float rinv = nin ? 1.0f / a.row_stat[((size_t)b * T + n) * 2 + 1] : 0.f;
    const float dl1 = nin ? a.delta1[(size_t)b * T + n] : 0.f;
    const bool tm = nin ? mask_live(a.text_mask, a.text_len, b, T, n) : false;
    const float tmf = tm ? 1.f : 0.f;
    const tr_off tr = make_tr_off(lane);

    float im[4] = {0.f, 0.f, 0.f, 0.f};   // mod, mod_d, q, dq
    for (int j = tid; j < Mp; j += NTHR) {
        const bool in = j < M;
        const size_t bj = (size_t)b * M + j;
        ct_all[j] = in ? a.cterm[bj] : 0.f;
        cmax_all[j] = in ? a.col_stat[bj * 2] : 0.f;
        cinv_all[j] = in ? 1.0f / a.col_stat[bj * 2 + 1] : 0.f;
        dl2_all[j] = in ? a.delta2[bj] : 0.f;
        mmf_all[j] = in ? (mask_live(a.mod_mask, a.mod_len, b, M, j) ? 1.f : 0.f) : -1.f;
        const size_t pj = (size_t)b * Mp + j;
        const float v0 = in ? a.iM[pj] : 0.f, v1 = in ? a.iMd[pj] : 0.f, v2 = in ? a.iQ[pj] : 0.f, v3 = in ? a.iDq[pj] : 0.f;
        sM_all[j] = v0; sMd_all[j] = v1; sQ_all[j] = v2; sDq_all[j] = v3;
        sSp_all[j] = SAME ? (in ? a.iMw[pj] : 0.f) : v1;
        im[0] = fmaxf(im[0], v0); im[1] = fmaxf(im[1], v1); im[2] = fmaxf(im[2], v2); im[3] = fmaxf(im[3], v3);
    }
    wg_allmax(im, red, tid);
    const float cDq = cmap(im[3]);
    const float cS = cmap_bound(im[1], 1.3743895e11f * (inDa * im[0] + inDb * im[2] + im[3] * inT));

    acc_t dtx, dX;
    zero_acc(dtx);
    zero_acc(dX);
    float dr = 0.f;
    const char* m_b = a.pM + (size_t)b * szM;
    const char* md_b = a.pMd + (size_t)b * szM;
    const char* q_b = a.pQ + (size_t)b * szM;
    const char* dq_b = a.pDq + (size_t)b * szM;
    const char* sp_b = a.pMw + (size_t)b * szM;
    const int Mloop = (dbg & 16) ? 0 : M;
    for (int p0 = 0; p0 < Mloop; p0 += PR) {
        __syncthreads();
        if (!(dbg & 1) || p0 == 0) {
            stage_panel(pM, m_b, p0, tid);
            stage_panel(pQ, q_b, p0, tid);
            stage_panel(pDq, dq_b, p0, tid);
            if (SAME) stage_panel(pSp, sp_b, p0, tid);
            else if (sep) stage_panel(pMd, md_b, p0, tid);
        }
        __syncthreads();
        f4 s[2], dpa[2], dpb[2], dp2[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) s[q] = dpa[q] = dpb[q] = dp2[q] = f4{0.f, 0.f, 0.f, 0.f};
        if (!(dbg & 2)) {
            sprod2<false>(pSp, r, g, sideS, s);
            sprod2<false>(pM, r, g, sideDa, dpa);
            sprod2<false>(pQ, r, g, sideDb, dpb);
            sprod2<false>(pDq, r, g, sideT, dp2);
        }
        f4 wt[2], wx[2];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int ml = p0 + mb * 16 + 4 * g + e;
                const float mf = mmf_all[ml];
                const float x = s[mb][e] * (sSp_all[ml] * inS) + rterm + ct_all[ml];
                const float P1 = mf >= 0.f ? __expf((mf > 0.f ? x : NEG) - rmax) * rinv : 0.f;
                const float P2 = mf >= 0.f ? __expf((tm ? x : NEG) - cmax_all[ml]) * cinv_all[ml] : 0.f;
                const float dp1 = dpa[mb][e] * (sM_all[ml] * inDa) + dpb[mb][e] * (sQ_all[ml] * inDb);
                const float g1 = mf > 0.f ? P1 * (dp1 - dl1) : 0.f;
                const float g2 = P2 * (dp2[mb][e] * (sDq_all[ml] * inT) - dl2_all[ml]) * tmf;
                dr += g1 + g2;
                wt[mb][e] = P2 * (sDq_all[ml] * cDq);
                wx[mb][e] = (g1 + g2) * (sMd_all[ml] * cS);
            }
        half8 W0, W1;
        split_w(wt[0], wt[1], W0, W1);
        if (!(dbg & 4)) pvprod<1>(pDq, tr, W0, W1, dtx);
        split_w(wx[0], wx[1], W0, W1);
        if (!(dbg & 4)) pvprod<1>(pMd, tr, W0, W1, dX);
    }
    dr = kg_allsum(dr);
    if (dbg & 8) return;
    const float sdtx = 1.0f / cDq, sdX = 1.0f / cS;
    // ---- epilogue.  The accumulator tiles hold 16 rows x 64-B pieces per instruction; the workgroup parks dX and the
    // P2.dq sum in LDS (the panels are dead) and then works on whole rows -- one text row per wave-instruction, lane = 16-B
    // chunk -- so that text_d, the d_text read-modify-write and d_text_d are fully coalesced, and the parameter-gradient
    // sums over rows (d_w_t, d_w_tm) are plain per-lane accumulations over the wave's 16 rows (no cross-lane reduction).
    float* eX = reinterpret_cast<float*>(smem);                 // [64][LDP]  dX
    float* eT = eX + 16 * NW * LDP;                             // [64][LDP]  sum_j P2 dq
    float* drs = eT + 16 * NW * LDP;                            // [64]       dr
    float* part = drs + 16 * NW;                                // [NW][2][256] per-wave partial sums of d_w_t, d_w_tm
    __syncthreads();
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
        *reinterpret_cast<f4*>(eX + (wave * 16 + r) * LDP + 16 * dt + 4 * g) = dX[dt] * sdX;
        *reinterpret_cast<f4*>(eT + (wave * 16 + r) * LDP + 16 * dt + 4 * g) = dtx[dt] * sdtx;
    }
    if (g == 0) drs[wave * 16 + r] = nin ? dr : 0.f;
    __syncthreads();
    const int row0 = tile * NW * 16, d4 = 4 * lane;
    const bool cin = d4 < D;
    const f4 wt4 = cin ? *reinterpret_cast<const f4*>(a.w_t + d4) : f4{0.f, 0.f, 0.f, 0.f};
    const f4 wtm = cin ? *reinterpret_cast<const f4*>(a.w_tm + d4) : f4{0.f, 0.f, 0.f, 0.f};
    f4 pt = f4{0.f, 0.f, 0.f, 0.f}, ptm = pt;
    f4 tdv[16], prev[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {   // all loads of the wave's 16 rows in flight together
        const int gn = row0 + wave + NW * k;
        const bool ok = gn < T && cin;
        const size_t o = ((size_t)b * T + min(gn, T - 1)) * D + d4;
        tdv[k] = ok ? *reinterpret_cast<const f4*>(a.text_d + o) : f4{0.f, 0.f, 0.f, 0.f};
        prev[k] = ok ? *reinterpret_cast<const f4*>(a.d_text + o) : f4{0.f, 0.f, 0.f, 0.f};   // g0 + g2*a + g3*b from the prologue
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int rr = wave + NW * k, gn = row0 + rr;
        if (gn < T && cin) {
            const f4 dXv = *reinterpret_cast<const f4*>(eX + rr * LDP + d4);
            const f4 dtv = *reinterpret_cast<const f4*>(eT + rr * LDP + d4);
            const float drr = drs[rr];
            const f4 gd = wt4 * drr + wtm * dXv;
            const size_t o = ((size_t)b * T + gn) * D + d4;
            if (a.fold) {
                *reinterpret_cast<f4*>(a.d_text + o) = prev[k] + dtv + gd;
            } else {
                *reinterpret_cast<f4*>(a.d_text + o) = prev[k] + dtv;
                *reinterpret_cast<f4*>(a.d_text_d + o) = gd;
            }
            pt += tdv[k] * drr;
            ptm += tdv[k] * dXv;
        }
    }
    *reinterpret_cast<f4*>(part + (wave * 2 + 0) * 256 + d4) = pt;
    *reinterpret_cast<f4*>(part + (wave * 2 + 1) * 256 + d4) = ptm;
    __syncthreads();
    {
        const int which = tid >> 7 ? 1 : 0, i = tid & 127;   // 2 x 128 threads x 2 features cover d_w_t | d_w_tm (D <= 208)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int d = i + 128 * h;
            if (d < D) {
                float acc = 0.f;
#pragma unroll
                for (int w = 0; w < NW; ++w) acc += part[(w * 2 + which) * 256 + d];
                atomicAdd((which ? a.d_w_tm : a.d_w_t) + d, acc);
            }
        }
    }
    if (wave == 0) {
        const float sb = wave_allsum(drs[lane]);
        if (lane == 0) atomicAdd(a.d_bias, sb);
    }
}

// ------------------------------------------------------------------------------------------ host side
// How many ways to split the streamed side (R rows, PR-row panels) of a sweep whose lane side has N rows per sample:
// the workgroups (4 waves = 64 lane-side rows each) run in rounds of `slots` (256 CUs x workgroups that fit a CU),
// so the cost is rounds x (panels per split + fixed per-workgroup work) plus the traffic of the per-split partials.
static int pick_splits(int B, int N, int R, int slots = 256) {
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
// tuning aid (tools/att_bench.py): MMB_ATT_FSPLIT / MMB_ATT_BSPLIT force the number of splits of the column pass / of
// the backward j sweeps (read at every call)
static int forced_splits(const char* name, int R) {
    const char* e = getenv(name);
    const int v = e ? atoi(e) : 0;
    return v > 0 ? (v < (R + PR - 1) / PR ? v : (R + PR - 1) / PR) : 0;
}
static int rows_per_split(int R, int splits) {
    int rp = (R + splits - 1) / splits;
    return (rp + PR - 1) / PR * PR;
}

static size_t align256(size_t x) { return (x + 255) / 256 * 256; }

// saved-for-backward buffer of the fused path: planes + inverse row scales of text, mod, q (and of the dropped copies)
struct SavedLayout {
    size_t pT, pTd, pM, pMd, pQ, pTw, pMw, iT, iTd, iM, iMd, iQ, iTw, iMw, total;
};
static SavedLayout saved_layout(int B, int T, int M, int drop) {
    SavedLayout L{};
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t at = o; o += align256(bytes); return at; };
    const size_t szT = planes_sample_bytes(T) * B, szM = planes_sample_bytes(M) * B;
    const size_t nT = (size_t)B * pad32(T) * sizeof(float), nM = (size_t)B * pad32(M) * sizeof(float);
    L.pT = take(szT);
    L.pTd = drop ? take(szT) : L.pT;
    L.pM = take(szM);
    L.pMd = drop ? take(szM) : L.pM;
    L.pQ = take(szM);
    L.pTw = take(szT);
    L.pMw = take(szM);
    L.iT = take(nT);
    L.iTd = drop ? take(nT) : L.iT;
    L.iM = take(nM);
    L.iMd = drop ? take(nM) : L.iM;
    L.iQ = take(nM);
    L.iTw = take(nT);
    L.iMw = take(nM);
    L.total = o;
    return L;
}

struct BwdWs {
    size_t pDa, pDb, pDq, iDa, iDb, iDq, delta1, delta2, p_dq, p_dmc, p_dmd1, p_dmd2, p_dc1, p_dc2, total;   // bytes
    int splits;
};
static BwdWs bwd_layout(int B, int T, int M, int D) {
    BwdWs w{};
    w.splits = pick_splits(B, M, T);
    if (int f = forced_splits("MMB_ATT_BSPLIT", T)) w.splits = f;
    const size_t S = w.splits;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t at = o; o += align256(bytes); return at; };
    const size_t szT = planes_sample_bytes(T) * B, szM = planes_sample_bytes(M) * B;
    w.pDa = take(szT);
    w.pDb = take(szT);
    w.pDq = take(szM);
    w.iDa = take((size_t)B * pad32(T) * 4);
    w.iDb = take((size_t)B * pad32(T) * 4);
    w.iDq = take((size_t)B * pad32(M) * 4);
    w.delta1 = take((size_t)B * T * 4);
    w.delta2 = take((size_t)B * M * 4);
    w.p_dq = take(S * B * M * D * 4);
    w.p_dmc = take(S * B * M * D * 4);
    w.p_dmd1 = take(S * B * M * D * 4);
    w.p_dmd2 = take(S * B * M * D * 4);
    w.p_dc1 = take(S * B * M * 4);
    w.p_dc2 = take(S * B * M * 4);
    w.total = o;
    return w;
}
static int fwd_splits(int B, int T, int M) {
    if (int f = forced_splits("MMB_ATT_FSPLIT", T)) return f;
    return pick_splits(B, M, T, 512);   // two workgroups per CU
}
static size_t fwd_ws_bytes(int B, int T, int M, int D) {
    const size_t S = fwd_splits(B, T, M);
    return align256(S * B * M * D * 4) + align256(S * B * M * 2 * 4);
}

template <typename K>
static int allow_lds(K kernel, size_t bytes) {
    MMB_REQUIRE(bytes <= 160 * 1024, "bidaf: %zu bytes of LDS needed, 160 KiB available (sequence too long for one split)", bytes);
    MMB_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return MMB_OK;
}

}  // namespace mmb

using namespace mmb;

static int check_att_dims(int B, int T, int M, int D) {
    MMB_REQUIRE(B >= 1 && T >= 1 && M >= 1, "bidaf: bad sizes B=%d T=%d M=%d", B, T, M);
    MMB_REQUIRE(D >= 4 && D % 4 == 0 && D <= MMB_ATT_GENERAL_MAX_D, "bidaf: D=%d must be a multiple of 4 and <= %d", D,
                MMB_ATT_GENERAL_MAX_D);
    return MMB_OK;
}

extern "C" void mmb_set_att_debug(int mask) { mmb::g_att_dbg = mask; }

extern "C" size_t mmb_bidaf_saved_bytes(int B, int T, int M, int D, int has_drop) {
    if (B < 1 || T < 1 || M < 1 || D < 4) return 0;
    if (D > MMB_ATT_MAX_D) return (size_t)B * M * D * sizeof(float);   // general path: q (B,M,D) fp32
    return saved_layout(B, T, M, has_drop).total;
}

extern "C" size_t mmb_bidaf_fwd_workspace_bytes(int B, int T, int M, int D) {
    if (B < 1 || T < 1 || M < 1 || D < 4) return 0;
    if (D > MMB_ATT_MAX_D) return bidaf_big_fwd_ws_floats(B, T, M, D) * sizeof(float);
    return fwd_ws_bytes(B, T, M, D);
}

extern "C" int mmb_bidaf_fwd(const float* text, const float* mod, const uint8_t* text_mask, const uint8_t* mod_mask,
                             const int32_t* text_len, const int32_t* mod_len,
                             const float* text_d, const float* mod_d, const float* w_t, const float* w_m,
                             const float* w_tm, const float* bias, float* out, float* bsave, float* rterm,
                             float* cterm, float* row_stat, float* col_stat, void* saved, size_t saved_bytes,
                             float* workspace, size_t workspace_bytes,
                             int B, int T, int M, int D, int device, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (int rc = check_att_dims(B, T, M, D)) return rc;
    MMB_REQUIRE(text && mod && w_t && w_m && w_tm && bias && out && bsave && rterm && cterm && row_stat && col_stat && saved,
                "mmb_bidaf_fwd: null pointer");
    MMB_REQUIRE((text_mask || text_len) && (mod_mask || mod_len), "mmb_bidaf_fwd: a mask or a length vector is needed for each side");
    const bool drop = text_d != nullptr;
    MMB_REQUIRE(drop == (mod_d != nullptr), "mmb_bidaf_fwd: text_d and mod_d must both be given or both be NULL");
    MMB_REQUIRE(saved_bytes >= mmb_bidaf_saved_bytes(B, T, M, D, drop), "mmb_bidaf_fwd: saved buffer too small (%zu < %zu)",
                saved_bytes, mmb_bidaf_saved_bytes(B, T, M, D, drop));
    MMB_REQUIRE(workspace && workspace_bytes >= mmb_bidaf_fwd_workspace_bytes(B, T, M, D),
                "mmb_bidaf_fwd: needs a workspace of mmb_bidaf_fwd_workspace_bytes() = %zu bytes", mmb_bidaf_fwd_workspace_bytes(B, T, M, D));
    MMB_HIP(hipSetDevice(device));
    if (!text_d) text_d = text;
    if (!mod_d) mod_d = mod;

    if (D > MMB_ATT_MAX_D) {   // general-size path (bidaf_big.hip): similarity matrix materialised in the workspace
        MMB_REQUIRE(text_mask && mod_mask, "mmb_bidaf_fwd: the general-width path (D > %d) takes u8 masks", MMB_ATT_MAX_D);
        {
            ProfScope ps_(MMB_K_ATT_RANK1, stream);
            hipLaunchKernelGGL(att_rank1_kernel, dim3((B * T + B * M + 3) / 4), dim3(256), 0, stream, text_d, mod_d, w_t, w_m, bias, rterm, cterm,
                               B * T, B * M, D);
        }
        MMB_HIP(hipGetLastError());
        return bidaf_big_fwd(text, mod, text_mask, mod_mask, text_d, mod_d, w_tm, out, static_cast<float*>(saved), bsave, rterm, cterm,
                             row_stat, col_stat, workspace, B, T, M, D, stream);
    }
    const SavedLayout L = saved_layout(B, T, M, drop);
    char* sv = static_cast<char*>(saved);
    auto fp = [&](size_t off) { return reinterpret_cast<float*>(sv + off); };
    ProfScope ps_all_(MMB_K_ATT_FWD, stream);   // the whole fused forward (bench.py's roofline figure)

    // ---- split passes: planes + inverse row scales of text / mod (and the dropped copies), rank-1 terms
    {
        PrepArgs p{};
        p.D = D; p.B = B;
        int k = 0;
        p.t[k++] = SplitSrc{text_d, sv + L.pTd, fp(L.iTd), w_t, bias, rterm, nullptr, T};
        p.t[k++] = SplitSrc{mod_d, sv + L.pMd, fp(L.iMd), w_m, nullptr, cterm, nullptr, M};
        p.t[k++] = SplitSrc{text_d, sv + L.pTw, fp(L.iTw), nullptr, nullptr, nullptr, w_tm, T};   // lane-side S operands: w_tm folded in
        p.t[k++] = SplitSrc{mod_d, sv + L.pMw, fp(L.iMw), nullptr, nullptr, nullptr, w_tm, M};
        if (drop) {
            p.t[k++] = SplitSrc{text, sv + L.pT, fp(L.iT), nullptr, nullptr, nullptr, nullptr, T};
            p.t[k++] = SplitSrc{mod, sv + L.pM, fp(L.iM), nullptr, nullptr, nullptr, nullptr, M};
        }
        p.n = k;
        const long rows = (long)B * pad32(T > M ? T : M);
        ProfScope ps_(MMB_K_ATT_RANK1, stream);
        hipLaunchKernelGGL(att_prep_kernel, dim3((unsigned)((rows + 3) / 4), k), dim3(256), 0, stream, p);
        MMB_HIP(hipGetLastError());
    }
    // ---- column pass: lane side = modality rows, streams text; partials merged (and q split into planes) by the combine
    const int splits = fwd_splits(B, T, M);
    float* part_o = workspace;
    float* part_stat = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + align256((size_t)splits * B * M * D * 4));
    {
        AttFwdArgs a{};
        a.side_p = sv + L.pMw; a.side_i = fp(L.iMw);
        a.mS = sv + L.pTd; a.iS = fp(L.iTd); a.mV0 = sv + L.pT; a.iV0 = fp(L.iT); a.mV1 = nullptr; a.iV1 = nullptr;
        a.m_mask = text_len ? nullptr : text_mask; a.m_len = text_len; a.m_term = rterm; a.n_term = cterm;
        a.part_o = part_o; a.part_stat = part_stat;
        a.N = M; a.R = T; a.D = D; a.B = B; a.splits = splits; a.rows_per_split = rows_per_split(T, splits);
        a.dbg = att_dbg();
        const size_t arr = ((size_t)5 * a.rows_per_split + 16) * sizeof(float), stage_b = (size_t)(drop ? 2 : 1) * PANEL_B;
        a.nstage = (2 * stage_b + arr <= 80 * 1024 && a.rows_per_split > PR) ? 2 : 1;   // keep two workgroups per CU
        size_t lds = a.nstage * stage_b + arr;
        if (lds < (size_t)16 * NW * LDP * sizeof(float)) lds = (size_t)16 * NW * LDP * sizeof(float);   // epilogue staging tile
        auto kern = a.dbg ? att_col_kernel<true> : att_col_kernel<false>;
        if (int rc = allow_lds(kern, lds)) return rc;
        {
            ProfScope ps_(MMB_K_ATT_COL, stream);
            hipLaunchKernelGGL(kern, dim3(((M + 63) / 64) * splits * B), dim3(NTHR), lds, stream, a);
        }
        MMB_HIP(hipGetLastError());
        const long rows = (long)B * pad32(M);
        {
            ProfScope ps_(MMB_K_ATT_COMBINE, stream);
            hipLaunchKernelGGL(att_combine_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, part_o, part_stat, sv + L.pQ,
                               fp(L.iQ), col_stat, B, M, D, splits);
        }
        MMB_HIP(hipGetLastError());
    }
    // ---- row pass: lane side = text rows, streams [mod | q]
    {
        AttFwdArgs a{};
        a.side_p = sv + L.pTw; a.side_i = fp(L.iTw);
        a.mS = sv + L.pMd; a.iS = fp(L.iMd); a.mV0 = sv + L.pM; a.iV0 = fp(L.iM); a.mV1 = sv + L.pQ; a.iV1 = fp(L.iQ);
        a.m_mask = mod_len ? nullptr : mod_mask; a.m_len = mod_len; a.m_term = cterm; a.n_term = rterm; a.stat = row_stat;
        a.text = text; a.out = out; a.bsave = bsave;
        a.N = T; a.R = M; a.D = D; a.B = B; a.splits = 1; a.rows_per_split = rows_per_split(M, 1);
        a.dbg = att_dbg();
        const size_t arr = ((size_t)5 * a.rows_per_split + 16) * sizeof(float), stage_b = (size_t)(drop ? 3 : 2) * PANEL_B;
        a.nstage = (2 * stage_b + arr <= 160 * 1024 && a.rows_per_split > PR) ? 2 : 1;
        size_t lds = a.nstage * stage_b + arr;
        const size_t epi = (size_t)16 * NW * LDP * sizeof(float);
        if (lds < epi) lds = epi;
        static const bool row8 = [] { const char* e = getenv("MMB_ATT_ROW8"); return !e || atoi(e) != 0; }();
        if (row8 && lds < 2 * epi) lds = 2 * epi;   // the two roles' epilogue staging tiles
        auto kern = row8 ? (a.dbg ? att_row8_kernel<true> : att_row8_kernel<false>) : (a.dbg ? att_row_kernel<true> : att_row_kernel<false>);
        if (int rc = allow_lds(kern, lds)) return rc;
        {
            ProfScope ps_(MMB_K_ATT_ROW, stream);
            hipLaunchKernelGGL(kern, dim3(((T + 63) / 64) * B), dim3(row8 ? 2 * NTHR : NTHR), lds, stream, a);
        }
        MMB_HIP(hipGetLastError());
    }
    return MMB_OK;
}

extern "C" size_t mmb_bidaf_bwd_workspace_bytes(int B, int T, int M, int D) {
    if (B < 1 || T < 1 || M < 1 || D < 4) return 0;
    if (D > MMB_ATT_MAX_D) return bidaf_big_bwd_ws_floats(B, T, M, D) * sizeof(float);
    return bwd_layout(B, T, M, D).total;
}

extern "C" int mmb_bidaf_bwd(const float* d_out, const float* out, const float* text, const float* mod,
                             const uint8_t* text_mask, const uint8_t* mod_mask, const int32_t* text_len, const int32_t* mod_len,
                             const float* text_d, const float* mod_d,
                             const float* w_t, const float* w_m, const float* w_tm, const void* saved, const float* bsave,
                             const float* rterm, const float* cterm, const float* row_stat, const float* col_stat,
                             float* d_text, float* d_mod, float* d_text_d, float* d_mod_d, float* d_w_t, float* d_w_m,
                             float* d_w_tm, float* d_bias, float* workspace, size_t workspace_bytes, int B, int T, int M,
                             int D, int device, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (int rc = check_att_dims(B, T, M, D)) return rc;
    MMB_REQUIRE(d_out && out && text && mod && w_t && w_m && w_tm && saved && bsave && rterm && cterm &&
                    row_stat && col_stat && d_text && d_mod && d_w_t && d_w_m && d_w_tm && d_bias && workspace,
                "mmb_bidaf_bwd: null pointer");
    MMB_REQUIRE((text_mask || text_len) && (mod_mask || mod_len), "mmb_bidaf_bwd: a mask or a length vector is needed for each side");
    const bool drop_t = text_d != nullptr, drop_m = mod_d != nullptr;
    MMB_REQUIRE(drop_t == drop_m, "mmb_bidaf_bwd: text_d and mod_d must both be given or both be NULL");
    MMB_REQUIRE(drop_t ? (d_text_d && d_mod_d) : (!d_text_d && !d_mod_d),
                "mmb_bidaf_bwd: d_text_d/d_mod_d must be given exactly when text_d/mod_d are");
    MMB_REQUIRE(workspace_bytes >= mmb_bidaf_bwd_workspace_bytes(B, T, M, D), "mmb_bidaf_bwd: workspace too small (%zu < %zu)",
                workspace_bytes, mmb_bidaf_bwd_workspace_bytes(B, T, M, D));
    MMB_HIP(hipSetDevice(device));
    if (D > MMB_ATT_MAX_D) {
        MMB_REQUIRE(text_mask && mod_mask, "mmb_bidaf_bwd: the general-width path (D > %d) takes u8 masks", MMB_ATT_MAX_D);
        return bidaf_big_bwd(d_out, out, text, mod, text_mask, mod_mask, text_d, mod_d, w_t, w_m, w_tm, static_cast<const float*>(saved), bsave,
                             rterm, cterm, row_stat, col_stat, d_text, d_mod, d_text_d, d_mod_d, d_w_t, d_w_m, d_w_tm, d_bias, workspace,
                             B, T, M, D, stream);
    }
    const SavedLayout S = saved_layout(B, T, M, drop_t);
    const BwdWs L = bwd_layout(B, T, M, D);
    ProfScope ps_all_(MMB_K_ATT_BWD, stream);   // the whole fused backward
    const char* sv = static_cast<const char*>(saved);
    char* ws = reinterpret_cast<char*>(workspace);
    auto sf = [&](size_t off) { return reinterpret_cast<const float*>(sv + off); };
    auto wf = [&](size_t off) { return reinterpret_cast<float*>(ws + off); };

    AttBwdArgs a{};
    a.text = text; a.mod = mod; a.text_d = drop_t ? text_d : text; a.mod_d = drop_m ? mod_d : mod;
    a.text_mask = text_len ? nullptr : text_mask; a.mod_mask = mod_len ? nullptr : mod_mask;
    a.text_len = text_len; a.mod_len = mod_len;
    a.w_t = w_t; a.w_m = w_m; a.w_tm = w_tm;
    a.rterm = rterm; a.cterm = cterm; a.row_stat = row_stat; a.col_stat = col_stat;
    a.pT = sv + S.pT; a.pTd = sv + S.pTd; a.pM = sv + S.pM; a.pMd = sv + S.pMd; a.pQ = sv + S.pQ; a.pTw = sv + S.pTw; a.pMw = sv + S.pMw;
    a.iT = sf(S.iT); a.iTd = sf(S.iTd); a.iM = sf(S.iM); a.iMd = sf(S.iMd); a.iQ = sf(S.iQ); a.iTw = sf(S.iTw); a.iMw = sf(S.iMw);
    a.pDa = ws + L.pDa; a.pDb = ws + L.pDb; a.pDq = ws + L.pDq;
    a.iDa = wf(L.iDa); a.iDb = wf(L.iDb); a.iDq = wf(L.iDq);
    a.delta1 = wf(L.delta1); a.delta2 = wf(L.delta2);
    a.p_dq = wf(L.p_dq); a.p_dmc = wf(L.p_dmc); a.p_dmd1 = wf(L.p_dmd1);
    a.p_dmd2 = wf(L.p_dmd2); a.p_dc1 = wf(L.p_dc1); a.p_dc2 = wf(L.p_dc2);
    a.d_mod = d_mod; a.d_mod_d = d_mod_d; a.d_text = d_text; a.d_text_d = d_text_d;
    a.d_w_t = d_w_t; a.d_w_m = d_w_m; a.d_w_tm = d_w_tm; a.d_bias = d_bias;
    a.B = B; a.T = T; a.M = M; a.D = D; a.fold = drop_t ? 0 : 1;
    a.splits = L.splits;
    a.rows_per_split = rows_per_split(T, a.splits);
    a.dbg = att_dbg();

    {
        const long rows = (long)B * pad32(T);
        ProfScope ps_(MMB_K_ATT_BWD_PRE, stream);
        hipLaunchKernelGGL(att_bwd_pre_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, d_out, out, text, bsave,
                           ws + L.pDa, wf(L.iDa), ws + L.pDb, wf(L.iDb), wf(L.delta1), d_text, d_w_t, d_w_m, d_w_tm, d_bias, B, T, D);
    }
    MMB_HIP(hipGetLastError());
    const int tiles_m = (M + 63) / 64, tiles_t = (T + 63) / 64;
    {
        const size_t lds = (size_t)3 * PANEL_B + ((size_t)7 * a.rows_per_split + 16) * sizeof(float);
        auto kern = a.dbg ? att_bwd_j1_kernel<true> : att_bwd_j1_kernel<false>;
        if (int rc = allow_lds(kern, lds)) return rc;
        ProfScope ps_(MMB_K_ATT_BWD_J1, stream);
        hipLaunchKernelGGL(kern, dim3(tiles_m * a.splits * B), dim3(NTHR), lds, stream, a);
    }
    MMB_HIP(hipGetLastError());
    {
        const size_t arr = ((size_t)4 * a.rows_per_split + 16) * sizeof(float), stage_b = (size_t)(drop_t ? 2 : 1) * PANEL_B;
        a.nstage_j2 = (2 * stage_b + arr <= 160 * 1024 && a.rows_per_split > PR) ? 2 : 1;
        size_t lds = a.nstage_j2 * stage_b + arr;
        if (lds < (size_t)16 * NW * LDP * sizeof(float)) lds = (size_t)16 * NW * LDP * sizeof(float);   // epilogue staging tile
        auto kern = a.dbg ? att_bwd_j2_kernel<true> : att_bwd_j2_kernel<false>;
        if (int rc = allow_lds(kern, lds)) return rc;
        ProfScope ps_(MMB_K_ATT_BWD_J2, stream);
        hipLaunchKernelGGL(kern, dim3(tiles_m * a.splits * B), dim3(NTHR), lds, stream, a);
    }
    MMB_HIP(hipGetLastError());
    // j-side epilogue: rows per wave such that there are enough waves to fill the chip and few enough workgroups that the
    // d_w_m atomics stay cheap.  Its workgroups ride in the i-side launch (same block size, needs only the j sweeps) unless
    // the timing tools ask for their own launch.
    const int rows = B * M;
    a.jf_rows = rows >= 8192 ? 8 : rows >= 4096 ? 4 : rows >= 2048 ? 2 : 1;
    const int jf_blocks = ((rows + a.jf_rows - 1) / a.jf_rows + 3) / 4;
    static const bool jf_separate = [] { const char* e = getenv("MMB_ATT_JFIN_SEPARATE"); return e && atoi(e) != 0; }();
    const bool separate = jf_separate || a.dbg != 0;
    if (separate) {
        ProfScope ps_(MMB_K_ATT_BWD_JFIN, stream);
        hipLaunchKernelGGL(att_bwd_jfin_kernel, dim3(jf_blocks), dim3(256), 0, stream, a);
        MMB_HIP(hipGetLastError());
    }
    {
        size_t lds = (size_t)4 * PANEL_B + ((size_t)10 * pad32(M) + 16) * sizeof(float);
        const size_t epi = ((size_t)2 * 16 * NW * LDP + 16 * NW + NW * 2 * 256) * sizeof(float);   // parked dX, P2.dq tiles + dr + partial sums
        if (lds < epi) lds = epi;
        const bool same = a.fold && a.pMd == a.pM && a.pTd == a.pT;
        auto kern = a.dbg ? (same ? att_bwd_i_kernel<true, true> : att_bwd_i_kernel<true, false>)
                          : (same ? att_bwd_i_kernel<false, true> : att_bwd_i_kernel<false, false>);
        if (int rc = allow_lds(kern, lds)) return rc;
        a.i_blocks = tiles_t * B;
        ProfScope ps_(MMB_K_ATT_BWD_I, stream);
        hipLaunchKernelGGL(kern, dim3(a.i_blocks + (separate ? 0 : jf_blocks)), dim3(NTHR), lds, stream, a);
    }
    MMB_HIP(hipGetLastError());
    return MMB_OK;
}
