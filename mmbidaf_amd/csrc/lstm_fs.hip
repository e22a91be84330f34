// Fused-step LSTM recurrence for hidden sizes above the register-resident limit (H > MMB_LSTM_MAX_H, e.g. BASELINE
// cfg5's H = 512): ONE kernel per time step -- recurrent product on the 16-bit matrix cores + the whole cell update --
// instead of lstm_big.hip's exact-f32 skinny GEMM + cell kernel (two launches, ~26 us per step pair at cfg5).
// The stream order is still the only barrier between steps: no cooperative launch, no spin wait, nothing that can hang.
//
//   forward step s   pre (B x 4H) = h_{s-1} (B x H) . W_hh^T,  then gates, c, h  (reference: nn.LSTM as called at
//                    layers/encoding.py:79-81,96 on a packed batch; same semantics and saved tensors as lstm.hip)
//   BPTT step s      dh (B x H) = d_a_{s-1} (B x 4H) . W_hh,   then d_a, running dc
//
// Arithmetic: the error-compensated two-term fp16 split of bidaf.hip / planes.hip (x s = h0 + h1, three MFMA cross
// products, fp32 accumulate: fp32-level accuracy).  W_hh is split ONCE per layer call into planes whose ROWS are in
// unit-major order (row 4u + gate), so the 16x16 accumulator tile of v_mfma_f32_16x16x32_f16 -- lane (n, g) holds rows
// 4g..4g+3 of column n -- gives every lane the FOUR GATES of one (unit, sample) pair: the cell update is lane-local, no
// LDS exchange, no barrier in the kernel at all.  The recurrent operand travels between steps as planes too:
//   h      : |h| < 1, fixed scale 2^13, each lane stores its own two halfs into the next step's planes (ping-pong)
//   d_a    : scale 2^4 / max(previous step's max |d_a|, max |d_y|) tracked per (problem, direction) by atomicMax -- 2^10 of
//            head-room for growth between consecutive steps, 2^-29 of the maximum as absolute precision floor; the very
//            first step uses the true bound max |d_y| + max |d_hn|.
// Operand fragments are read straight from global memory (L2-resident: W planes 4 MB per chain at H = 512, the
// recurrent operand 128 / 512 KB), 16 B per lane, exactly the fragment shape -- no LDS at all.
//
// STATUS: opt-in (MMB_LSTM_FS=1), parity-tested at H = 136 / 144 / 256 / 512, and SLOWER than lstm_big.hip's two launches
// per step at cfg5 (B=64, H=512): 119 vs 92 ms per region step (r02; 134 ms before the fragment loads were batched by hand).  Why: with W_hh not resident, every workgroup
// re-reads its 128-KB W slice AND the whole recurrent operand of its chain every step -- 768 KB per workgroup and step
// forward (147 MB chip-wide), 4x that in the BPTT where K = 4H -- and one CU takes in ~100 GB/s from L2: ~8 us per
// forward step, more backward.  Staging through LDS would cut the forward to 256 KB per workgroup (~4 us); the BPTT
// operand (512 KB per chain and step, needed by every workgroup of the chain) does not fit.  Only a persistent kernel
// with W_hh resident in LDS avoids the reload -- and needs a per-step barrier across the 32 workgroups of a chain.
#include "common.h"

namespace mmb {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));

constexpr float FS_HSCALE = 8192.0f;   // 2^13: |h| < 1

// planes of an (R x K) operand: [row block of 16][k tile of 32][plane 0|1][16 rows x 64 B], no swizzle (never in LDS)
__host__ __device__ __forceinline__ size_t fs_planes_bytes(int rows, int nkt) { return (size_t)((rows + 15) / 16) * nkt * 2048; }
__device__ __forceinline__ size_t fs_off(int row, int oct, int nkt) {
    return ((size_t)(row >> 4) * nkt + (oct >> 2)) * 2048 + (row & 15) * 64 + (oct & 3) * 16;
}
__device__ __forceinline__ float fs_pow2_scale(float amax, int target_exp) {   // power of two s: s * amax in [2^(t-1), 2^t)
    const unsigned u = __float_as_uint(amax);
    const int e = (int)((u >> 23) & 0xFF) - 127;
    if (!(amax > 0.0f) || e > 100 || e < -100) return 1.0f;
    return __uint_as_float((unsigned)(target_exp - 1 - e + 127) << 23);
}
__device__ __forceinline__ f4 fs_mfma(const half8 a, const half8 b, const f4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

// ---- W_hh planes.  mode 0 (forward): plane row 4u+gate holds W_hh[gate*H+u][0..H)          (rows 4H, K = H)
//                    mode 1 (BPTT):    plane row u holds W_hh[gate*H+u'][u] at k = 4u'+gate     (rows H,  K = 4H)
// one wave per plane row: row maximum -> power-of-two scale (max to [2^13, 2^14)) -> two fp16 terms; padding = zeros
__global__ __launch_bounds__(256) void lstm_fs_wprep_kernel(const float* __restrict__ w_hh, char* __restrict__ planes,
                                                            float* __restrict__ inv, int H, int mode, int rows_p, int nkt) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows_p) return;
    const int rows = mode ? H : 4 * H, K = mode ? 4 * H : H;
    auto elem = [&](int k) -> float {
        if (row >= rows || k >= K) return 0.f;
        if (mode == 0) return w_hh[(size_t)((row & 3) * H + (row >> 2)) * H + k];
        return w_hh[(size_t)((k & 3) * H + (k >> 2)) * H + row];
    };
    float amax = 0.f;
    for (int k = lane; k < K; k += 64) amax = fmaxf(amax, fabsf(elem(k)));
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o));
    const float s = fs_pow2_scale(amax, 14);
    if (lane == 0) inv[row] = amax > 0.f ? 1.0f / s : 0.f;
    for (int oct = lane; oct < nkt * 4; oct += 64) {
        half8 h0, h1;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float v = elem(8 * oct + j) * s;
            const _Float16 a = (_Float16)v;
            h0[j] = a;
            h1[j] = (_Float16)(v - (float)a);
        }
        char* d = planes + fs_off(row, oct, nkt);
        *reinterpret_cast<half8*>(d) = h0;
        *reinterpret_cast<half8*>(d + 1024) = h1;
    }
}

// ------------------------------------------------------------------------------------------ forward step
struct FsFwdProb {
    const float* gx;        // (B,T,2,H,4)
    const int* len;
    float* y; float* gates; float* cs; float* h_n; float* c_n;
    const int* hn_pos;      // (B) or null: h_n as (B,2,H) rows hn_pos[b]
    const char* wp[2];      // W_hh planes (mode 0) per direction
    const float* winv[2];
    char* hp[2][2];         // h planes [direction][step parity], rows = samples padded to 64, zero at entry
    int B, T, H;
};
struct FsFwdArgs { FsFwdProb p[MMB_MAX_GROUP]; int n, nkt; };

// grid (unit slices of 16, chains = 2 * problems, sample blocks of 64); 8 waves: wave = (m tile of 4 units, half of the
// sample block); no LDS, no barrier
__global__ __launch_bounds__(512) void lstm_fs_fwd_kernel(const FsFwdArgs args, const int s) {
    const FsFwdProb& P = args.p[blockIdx.y >> 1];
    const int dir = blockIdx.y & 1;
    const int H = P.H, T = P.T, nkt = args.nkt;
    if (s >= T) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, g = lane >> 4, mt = wave & 3, nh = wave >> 2;
    const int b0 = 64 * blockIdx.z + 32 * nh;
    if (b0 >= P.B) return;
    const int rowA = 64 * blockIdx.x + 16 * mt + r;
    const char* A = P.wp[dir] + fs_off(rowA, g, nkt);
    const char* hprev = P.hp[dir][(s + 1) & 1];
    const char* Bq[2] = {hprev + fs_off(b0 + r, g, nkt), hprev + fs_off(b0 + 16 + r, g, nkt)};
    f4 c[2] = {f4{0.f, 0.f, 0.f, 0.f}, f4{0.f, 0.f, 0.f, 0.f}};
    if (s > 0) {   // h_{-1} = 0
        // 4 k tiles per trip, all 24 fragment loads issued before the first MFMA (hipcc does not unroll this loop by
        // itself, and 6 loads in flight per wave leave the kernel waiting on L2 latency)
        const half8 z8 = {};
        for (int kt0 = 0; kt0 < nkt; kt0 += 4) {
            half8 a0[4], a1[4], h0[4][2], h1[4][2];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const size_t o = (size_t)min(kt0 + q, nkt - 1) * 2048;
                a0[q] = *reinterpret_cast<const half8*>(A + o);
                a1[q] = *reinterpret_cast<const half8*>(A + o + 1024);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    h0[q][j] = *reinterpret_cast<const half8*>(Bq[j] + o);
                    h1[q][j] = *reinterpret_cast<const half8*>(Bq[j] + o + 1024);
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (kt0 + q >= nkt) { a0[q] = z8; a1[q] = z8; }
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    c[j] = fs_mfma(a0[q], h1[q][j], c[j]);
                    c[j] = fs_mfma(a1[q], h0[q][j], c[j]);
                    c[j] = fs_mfma(a0[q], h0[q][j], c[j]);
                }
            }
        }
    }
    // lane (n = r, g) holds rows 4g..4g+3 of the tile = gates i,f,g,o of unit u for sample b
    const int u = 16 * blockIdx.x + 4 * mt + g;
    if (u >= H) return;
    const f4 wi = *reinterpret_cast<const f4*>(P.winv[dir] + 4 * u);
    char* hnext = P.hp[dir][s & 1];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int b = b0 + 16 * j + r;
        if (b >= P.B) continue;
        const int len = min(max(P.len[b], 0), T);
        if (s >= len) continue;
        const int t = dir ? len - 1 - s : s;
        const size_t row = (size_t)b * T + t;
        const f4 pre = c[j] * wi * (1.0f / FS_HSCALE);
        const f4 gx = *reinterpret_cast<const f4*>(P.gx + (row * 2 + dir) * 4 * H + (size_t)u * 4);
        float c_prev = 0.f;
        if (s > 0) c_prev = P.cs[((size_t)b * T + (dir ? t + 1 : t - 1)) * 2 * H + dir * H + u];
        const float gi = sigmoidf_(pre.x + gx.x), gf = sigmoidf_(pre.y + gx.y);
        const float gg = tanhf_(pre.z + gx.z), go = sigmoidf_(pre.w + gx.w);
        const float cc = fmaf(gf, c_prev, gi * gg);
        const float h = go * tanhf_(cc);
        *reinterpret_cast<f4*>(P.gates + (row * 2 + dir) * 4 * H + (size_t)u * 4) = f4{gi, gf, gg, go};
        P.cs[row * 2 * H + dir * H + u] = cc;
        P.y[row * 2 * H + dir * H + u] = h;
        {   // this lane's element of the next step's operand planes
            const float hv = h * FS_HSCALE;
            const _Float16 h0 = (_Float16)hv;
            const _Float16 h1 = (_Float16)(hv - (float)h0);
            char* d = hnext + fs_off(b, u >> 3, nkt) + (u & 7) * 2;
            *reinterpret_cast<_Float16*>(d) = h0;
            *reinterpret_cast<_Float16*>(d + 1024) = h1;
        }
        if (s == len - 1) {
            const size_t st = ((size_t)dir * P.B + b) * H + u;
            P.h_n[P.hn_pos ? ((size_t)P.hn_pos[b] * 2 + dir) * H + u : st] = h;
            P.c_n[st] = cc;
        }
    }
}

// ------------------------------------------------------------------------------------------ BPTT step
struct FsBwdProb {
    const float* d_y; const float* d_hn; const float* gates; const float* cs;
    const int* hn_pos; const int* len;
    float* d_a;             // (B,T,2,4H) torch gate order
    float* dc;              // (2,B,H) running cell-state gradient
    const char* wtp[2];     // W_hh planes (mode 1) per direction
    const float* wtinv[2];
    char* ap[2][2];         // d_a planes [direction][step parity], rows = samples padded to 64, K = 4H unit-major; zero at entry
    float* amax[2];         // per direction: [T + 1] running maxima; amax[dir][s] = max |d_a| of step s (atomicMax, zero at entry)
    const float* bound;     // [2]: max |d_y|, max |d_hn| over the whole problem
    int B, T, H;
};
struct FsBwdArgs { FsBwdProb p[MMB_MAX_GROUP]; int n, nkt4; };

// scale of the d_a planes WRITTEN at step s (and read at step s+1): every lane derives the same power of two
__device__ __forceinline__ float fs_da_scale(const FsBwdProb& P, int dir, int s) {
    const float y = P.bound[0];
    const float ref = s == 0 ? y + P.bound[1] : fmaxf(P.amax[dir][s - 1], y);
    return fs_pow2_scale(ref, 5);      // reference magnitude to [2^4, 2^5): 2^10 of head-room below the 2^15 clamp
}

// grid (unit slices of 32, chains, sample blocks of 64); 8 waves: wave = (m tile of 16 units, n tile of 16 samples), full K
__global__ __launch_bounds__(512) void lstm_fs_bwd_kernel(const FsBwdArgs args, const int s) {
    const FsBwdProb& P = args.p[blockIdx.y >> 1];
    const int dir = blockIdx.y & 1;
    const int H = P.H, T = P.T, nkt = args.nkt4;
    if (s >= T) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, g = lane >> 4, mt = wave & 1, nt = wave >> 1;
    const int b = 64 * blockIdx.z + 16 * nt + r;
    if (64 * blockIdx.z + 16 * nt >= P.B) return;
    f4 c = f4{0.f, 0.f, 0.f, 0.f};
    if (s > 0) {
        const char* A = P.wtp[dir] + fs_off(32 * blockIdx.x + 16 * mt + r, g, nkt);
        const char* Bq = P.ap[dir][(s + 1) & 1] + fs_off(64 * blockIdx.z + 16 * nt + r, g, nkt);
        const half8 z8 = {};
        for (int kt0 = 0; kt0 < nkt; kt0 += 8) {   // 8 k tiles per trip: 32 fragment loads in flight
            half8 a0[8], a1[8], d0[8], d1[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const size_t o = (size_t)min(kt0 + q, nkt - 1) * 2048;
                a0[q] = *reinterpret_cast<const half8*>(A + o);
                a1[q] = *reinterpret_cast<const half8*>(A + o + 1024);
                d0[q] = *reinterpret_cast<const half8*>(Bq + o);
                d1[q] = *reinterpret_cast<const half8*>(Bq + o + 1024);
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                if (kt0 + q >= nkt) { a0[q] = z8; a1[q] = z8; }
                c = fs_mfma(a0[q], d1[q], c);
                c = fs_mfma(a1[q], d0[q], c);
                c = fs_mfma(a0[q], d0[q], c);
            }
        }
    }
    const float inv_prev = s > 0 ? 1.0f / fs_da_scale(P, dir, s - 1) : 0.f;
    const float sc = fs_da_scale(P, dir, s);
    char* anext = P.ap[dir][s & 1];
    float lmax = 0.f;
    const int len = b < P.B ? min(max(P.len[b], 0), T) : 0;
    if (b < P.B && s < len) {
        // BPTT visits the forward processing order backwards: forward direction t = len-1-s, reverse direction t = s
        const int t = dir ? s : len - 1 - s;
        const size_t row = (size_t)b * T + t;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int u = 32 * blockIdx.x + 16 * mt + 4 * g + e;
            if (u >= H) continue;
            const size_t st = ((size_t)dir * P.B + b) * H + u;
            float dh = c[e] * P.wtinv[dir][u] * inv_prev + P.d_y[row * 2 * H + dir * H + u];
            float dcs = 0.f;
            if (s == 0) { if (P.d_hn) dh += P.d_hn[P.hn_pos ? ((size_t)P.hn_pos[b] * 2 + dir) * H + u : st]; }
            else dcs = P.dc[st];
            const f4 g4 = *reinterpret_cast<const f4*>(P.gates + (row * 2 + dir) * 4 * H + (size_t)u * 4);
            const float c_t = P.cs[row * 2 * H + dir * H + u];
            const bool has_prev = dir ? (t + 1 < len) : (t > 0);
            const float c_prev = has_prev ? P.cs[((size_t)b * T + (dir ? t + 1 : t - 1)) * 2 * H + dir * H + u] : 0.f;
            const float gi = g4.x, gf = g4.y, gg = g4.z, go = g4.w;
            const float tc = tanhf_(c_t);
            const float dc_t = fmaf(dh * go, 1.0f - tc * tc, dcs);
            const float da0 = dc_t * gg * gi * (1.0f - gi), da1 = dc_t * c_prev * gf * (1.0f - gf);
            const float da2 = dc_t * gi * (1.0f - gg * gg), da3 = dh * tc * go * (1.0f - go);
            float* da = P.d_a + row * 8 * H + (size_t)dir * 4 * H + u;
            da[0] = da0;
            da[(size_t)H] = da1;
            da[(size_t)2 * H] = da2;
            da[(size_t)3 * H] = da3;
            P.dc[st] = dc_t * gf;
            lmax = fmaxf(lmax, fmaxf(fmaxf(fabsf(da0), fabsf(da1)), fmaxf(fabsf(da2), fabsf(da3))));
            // the four gates of unit u are 4 consecutive k of the next step's operand: one 8-B store per plane
            const float v[4] = {da0 * sc, da1 * sc, da2 * sc, da3 * sc};
            half4 h0, h1;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float x = fminf(fmaxf(v[q], -60000.0f), 60000.0f);
                const _Float16 a = (_Float16)x;
                h0[q] = a;
                h1[q] = (_Float16)(x - (float)a);
            }
            char* d = anext + fs_off(b, u >> 1, nkt) + (u & 1) * 8;
            *reinterpret_cast<half4*>(d) = h0;
            *reinterpret_cast<half4*>(d + 1024) = h1;
        }
    }
    // (a sample whose chain has ended keeps stale but finite planes: whatever the later steps compute for it is ignored)
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) lmax = fmaxf(lmax, __shfl_xor(lmax, o));
    if (lane == 0 && lmax > 0.f) atomicMax(reinterpret_cast<unsigned*>(&P.amax[dir][s]), __float_as_uint(lmax));
}

// out[0] = max(out[0], max |p[i]|)  (out pre-zeroed)
__global__ __launch_bounds__(256) void lstm_fs_absmax_kernel(const float* __restrict__ p, long n, float* out) {
    float m = 0.f;
    if (p)
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) m = fmaxf(m, fabsf(p[i]));
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(reinterpret_cast<unsigned*>(out), __float_as_uint(m));
}

// ------------------------------------------------------------------------------------------ host side
static size_t fs_rup(size_t x) { return (x + 255) / 256 * 256; }
static int fs_pad(int x, int a) { return (x + a - 1) / a * a; }

struct FsFwdWs { size_t wp[2], winv[2], hp[2][2], total, zero_from; int nkt; };
static FsFwdWs fs_fwd_layout(int B, int H) {
    FsFwdWs w{};
    w.nkt = fs_pad(H, 32) / 32;
    const int rows_p = fs_pad(4 * H, 64), Bp = fs_pad(B, 64);
    size_t o = 0;
    for (int d = 0; d < 2; ++d) { w.wp[d] = o; o += fs_rup(fs_planes_bytes(rows_p, w.nkt)); }
    for (int d = 0; d < 2; ++d) { w.winv[d] = o; o += fs_rup((size_t)rows_p * 4); }
    w.zero_from = o;
    for (int d = 0; d < 2; ++d)
        for (int q = 0; q < 2; ++q) { w.hp[d][q] = o; o += fs_rup(fs_planes_bytes(Bp, w.nkt)); }
    w.total = o;
    return w;
}
struct FsBwdWs { size_t wtp[2], wtinv[2], bound, amax[2], dc, ap[2][2], total, zero_from; int nkt4; };
static FsBwdWs fs_bwd_layout(int B, int T, int H) {
    FsBwdWs w{};
    w.nkt4 = fs_pad(4 * H, 32) / 32;
    const int rows_p = fs_pad(H, 32), Bp = fs_pad(B, 64);
    size_t o = 0;
    for (int d = 0; d < 2; ++d) { w.wtp[d] = o; o += fs_rup(fs_planes_bytes(rows_p, w.nkt4)); }
    for (int d = 0; d < 2; ++d) { w.wtinv[d] = o; o += fs_rup((size_t)rows_p * 4); }
    w.zero_from = o;
    w.bound = o; o += 256;
    for (int d = 0; d < 2; ++d) { w.amax[d] = o; o += fs_rup((size_t)(T + 1) * 4); }
    w.dc = o; o += fs_rup((size_t)2 * B * H * 4);
    for (int d = 0; d < 2; ++d)
        for (int q = 0; q < 2; ++q) { w.ap[d][q] = o; o += fs_rup(fs_planes_bytes(Bp, w.nkt4)); }
    w.total = o;
    return w;
}
size_t lstm_fs_fwd_ws_bytes(int B, int H) { return fs_fwd_layout(B, H).total; }
size_t lstm_fs_bwd_ws_bytes(int B, int T, int H) { return fs_bwd_layout(B, T, H).total; }

// prep (W_hh planes, zeroed recurrent operand) + the time loop; lstm_big.hip does the packed-sequence post-processing
int lstm_fs_fwd(const mmb_lstm_fwd_desc* d, int n, char* const* ws, hipStream_t stream) {
    FsFwdArgs a{};
    a.n = n;
    const int H = d[0].H;
    int maxT = 0, maxB = 0;
    for (int i = 0; i < n; ++i) {
        const mmb_lstm_fwd_desc& p = d[i];
        const FsFwdWs L = fs_fwd_layout(p.B, H);
        a.nkt = L.nkt;
        MMB_HIP(hipMemsetAsync(ws[i] + L.zero_from, 0, L.total - L.zero_from, stream));   // h planes: h_{-1} = 0, padding rows 0
        FsFwdProb& q = a.p[i];
        q.gx = p.gx; q.len = p.lengths; q.y = p.y; q.gates = p.gates; q.cs = p.cs; q.h_n = p.h_n; q.c_n = p.c_n; q.hn_pos = p.hn_pos;
        q.B = p.B; q.T = p.T; q.H = H;
        const int rows_p = fs_pad(4 * H, 64);
        for (int dir = 0; dir < 2; ++dir) {
            q.wp[dir] = ws[i] + L.wp[dir];
            q.winv[dir] = reinterpret_cast<const float*>(ws[i] + L.winv[dir]);
            q.hp[dir][0] = ws[i] + L.hp[dir][0];
            q.hp[dir][1] = ws[i] + L.hp[dir][1];
            hipLaunchKernelGGL(lstm_fs_wprep_kernel, dim3((rows_p + 3) / 4), dim3(256), 0, stream, p.w_hh[dir], ws[i] + L.wp[dir],
                               reinterpret_cast<float*>(ws[i] + L.winv[dir]), H, 0, rows_p, L.nkt);
        }
        maxT = max(maxT, p.T);
        maxB = max(maxB, p.B);
    }
    MMB_HIP(hipGetLastError());
    const dim3 grid((H + 15) / 16, 2 * n, (maxB + 63) / 64);
    {
        ProfScope ps_(MMB_K_LSTM_REC_FWD, stream);
        for (int s = 0; s < maxT; ++s) hipLaunchKernelGGL(lstm_fs_fwd_kernel, grid, dim3(512), 0, stream, a, s);
    }
    MMB_HIP(hipGetLastError());
    return MMB_OK;
}

int lstm_fs_bwd(const mmb_lstm_bwd_desc* d, int n, char* const* ws, hipStream_t stream) {
    FsBwdArgs a{};
    a.n = n;
    const int H = d[0].H;
    int maxT = 0, maxB = 0;
    for (int i = 0; i < n; ++i) {
        const mmb_lstm_bwd_desc& p = d[i];
        const FsBwdWs L = fs_bwd_layout(p.B, p.T, H);
        a.nkt4 = L.nkt4;
        MMB_HIP(hipMemsetAsync(ws[i] + L.zero_from, 0, L.total - L.zero_from, stream));   // bounds, maxima, dc, d_a planes
        FsBwdProb& q = a.p[i];
        q.d_y = p.d_y; q.d_hn = p.d_hn; q.hn_pos = p.hn_pos; q.gates = p.gates; q.cs = p.cs; q.len = p.lengths;
        q.d_a = p.d_a; q.dc = reinterpret_cast<float*>(ws[i] + L.dc);
        float* bound = reinterpret_cast<float*>(ws[i] + L.bound);
        q.bound = bound;
        q.B = p.B; q.T = p.T; q.H = H;
        hipLaunchKernelGGL(lstm_fs_absmax_kernel, dim3(256), dim3(256), 0, stream, p.d_y, (long)p.B * p.T * 2 * H, bound);
        hipLaunchKernelGGL(lstm_fs_absmax_kernel, dim3(16), dim3(256), 0, stream, p.d_hn, (long)2 * p.B * H, bound + 1);
        const int rows_p = fs_pad(H, 32);
        for (int dir = 0; dir < 2; ++dir) {
            q.wtp[dir] = ws[i] + L.wtp[dir];
            q.wtinv[dir] = reinterpret_cast<const float*>(ws[i] + L.wtinv[dir]);
            q.ap[dir][0] = ws[i] + L.ap[dir][0];
            q.ap[dir][1] = ws[i] + L.ap[dir][1];
            q.amax[dir] = reinterpret_cast<float*>(ws[i] + L.amax[dir]);
            hipLaunchKernelGGL(lstm_fs_wprep_kernel, dim3((rows_p + 3) / 4), dim3(256), 0, stream, p.w_hh[dir], ws[i] + L.wtp[dir],
                               reinterpret_cast<float*>(ws[i] + L.wtinv[dir]), H, 1, rows_p, L.nkt4);
        }
        maxT = max(maxT, p.T);
        maxB = max(maxB, p.B);
    }
    MMB_HIP(hipGetLastError());
    const dim3 grid((H + 31) / 32, 2 * n, (maxB + 63) / 64);
    {
        ProfScope ps_(MMB_K_LSTM_REC_BWD, stream);
        for (int s = 0; s < maxT; ++s) hipLaunchKernelGGL(lstm_fs_bwd_kernel, grid, dim3(512), 0, stream, a, s);
    }
    MMB_HIP(hipGetLastError());
    return MMB_OK;
}

}  // namespace mmb
