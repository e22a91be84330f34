// General-size BiDAF attention (D = 2H > MMB_ATT_MAX_D, e.g. BASELINE cfg5's D = 1024): the fused kernels of bidaf.hip
// keep a row's whole feature vector in registers, which stops at D = 208.  Beyond that the similarity matrix is
// MATERIALISED per sample in a caller-provided workspace (B*T*M floats: 26 MB at cfg5) and every product runs on the
// batched exact-f32 MFMA GEMM; small kernels do the two masked softmaxes, the reductions and the element-wise glue.
// Same arithmetic as the reference (layers/attention.py:37-98: S = text.w_t + mod.w_m + (text*w_tm).mod^T + bias,
// masked_softmax over each axis with the -1e30 blend, a = s1.mod, b = s1.(s2^T.text), out = [text, a, text*a, text*b])
// and the same saved tensors / gradient conventions as the fused path, so the Python side does not change.
#include "common.h"

namespace mmb {

constexpr float BIG_NEG = -1e30f;

static int bgemm(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc, int ta, int tb,
                 int accumulate, int batch, long sA, long sB, long sC, hipStream_t stream) {
    GemmArgs g{};
    g.A = A; g.B = B; g.C = C; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.ta = ta; g.tb = tb; g.accumulate = accumulate; g.periodB = 1;
    g.batch = batch; g.sA = sA; g.sB = sB; g.sC = sC;
    if (batch == 1) g.batch = 0;
    return gemm_launch(g, stream);
}

// dst[row, d] = src[row, d] * w[d]
__global__ __launch_bounds__(256) void big_scale_kernel(const float* __restrict__ src, const float* __restrict__ w,
                                                        float* __restrict__ dst, long rows, int D) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;   // over rows * D/4
    if (i >= rows * (D / 4)) return;
    const int d4 = i % (D / 4);
    const f4 v = reinterpret_cast<const f4*>(src)[i] * reinterpret_cast<const f4*>(w)[d4];
    reinterpret_cast<f4*>(dst)[i] = v;
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// column softmax over i (text axis) with the text mask: P2[b,i,j], col_stat[b,j] = {max, sum}.  Workgroup = 64 columns x
// 4 waves that take every 4th row (256-B coalesced rows, four independent loads in flight per thread); each thread keeps
// an online (max, sum), the four partial pairs meet in LDS, a second pass writes P2.
__global__ __launch_bounds__(256) void big_colsoft_kernel(const float* __restrict__ S, const float* __restrict__ rterm,
                                                          const float* __restrict__ cterm, const uint8_t* __restrict__ tmask,
                                                          float* __restrict__ P2, float* __restrict__ col_stat, int T, int M) {
    __shared__ float pm[4][64], ps[4][64];
    const int b = blockIdx.y, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int j = blockIdx.x * 64 + lane;
    const bool on = j < M;
    const float* Sb = S + (size_t)b * T * M + (on ? j : 0);
    const float* r = rterm + (size_t)b * T;
    const uint8_t* mk = tmask + (size_t)b * T;
    const float c = on ? cterm[(size_t)b * M + j] : 0.f;
    float mx = -INFINITY, sum = 0.f;
    for (int i0 = w; i0 < T; i0 += 16) {
        float x[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = i0 + 4 * q;
            x[q] = i < T ? (mk[i] ? Sb[(size_t)i * M] + r[i] + c : BIG_NEG) : -INFINITY;
        }
        const float m4 = fmaxf(fmaxf(x[0], x[1]), fmaxf(x[2], x[3]));
        if (m4 > mx) { sum *= expf(mx - m4); mx = m4; }     // (mx = -inf at first: expf(-inf) = 0, sum is 0 anyway)
#pragma unroll
        for (int q = 0; q < 4; ++q) sum += expf(x[q] - mx);    // rows beyond T: expf(-inf) = 0
    }
    pm[w][lane] = mx; ps[w][lane] = sum;
    __syncthreads();
    mx = fmaxf(fmaxf(pm[0][lane], pm[1][lane]), fmaxf(pm[2][lane], pm[3][lane]));
    sum = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) sum += ps[q][lane] * expf(pm[q][lane] - mx);   // a wave without rows: 0 * expf(-inf) = 0
    if (!on) return;
    const float inv = 1.0f / sum;
    float* Pb = P2 + (size_t)b * T * M + j;
    for (int i0 = w; i0 < T; i0 += 16) {
        float x[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = i0 + 4 * q;
            x[q] = i < T ? (mk[i] ? Sb[(size_t)i * M] + r[i] + c : BIG_NEG) : 0.f;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (i0 + 4 * q < T) Pb[(size_t)(i0 + 4 * q) * M] = expf(x[q] - mx) * inv;
    }
    if (w == 0) {
        col_stat[((size_t)b * M + j) * 2] = mx;
        col_stat[((size_t)b * M + j) * 2 + 1] = sum;
    }
}

// row softmax over j (modality axis) with the modality mask, IN PLACE over S: P1[b,i,j], row_stat[b,i] = {max, sum}.
// One wave per row.
__global__ __launch_bounds__(256) void big_rowsoft_kernel(float* __restrict__ S, const float* __restrict__ rterm,
                                                          const float* __restrict__ cterm, const uint8_t* __restrict__ mmask,
                                                          float* __restrict__ row_stat, long BT, int T, int M) {
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= BT) return;
    const int b = row / T;
    float* Sr = S + (size_t)row * M;
    const float r = rterm[row];
    const float* c = cterm + (size_t)b * M;
    const uint8_t* mk = mmask + (size_t)b * M;
    float mx = -INFINITY;
    for (int j = lane; j < M; j += 64) mx = fmaxf(mx, mk[j] ? Sr[j] + r + c[j] : BIG_NEG);
    mx = wave_max(mx);
    float sum = 0.f;
    for (int j = lane; j < M; j += 64) sum += expf((mk[j] ? Sr[j] + r + c[j] : BIG_NEG) - mx);
    sum = wave_sum(sum);
    const float inv = 1.0f / sum;
    for (int j = lane; j < M; j += 64) Sr[j] = expf((mk[j] ? Sr[j] + r + c[j] : BIG_NEG) - mx) * inv;
    if (lane == 0) { row_stat[row * 2] = mx; row_stat[row * 2 + 1] = sum; }
}

// out[:, 0:D] = text, out[:, 2D:3D] = text*a, out[:, 3D:4D] = text*b, with a = out[:, D:2D] (written by the GEMM)
__global__ __launch_bounds__(256) void big_assemble_kernel(const float* __restrict__ text, const float* __restrict__ bsave,
                                                           float* __restrict__ out, long rows, int D) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * (D / 4)) return;
    const long row = i / (D / 4);
    const int d = (i % (D / 4)) * 4;
    const f4 t = *reinterpret_cast<const f4*>(text + row * D + d);
    const f4 b = *reinterpret_cast<const f4*>(bsave + row * D + d);
    float* o = out + row * 4 * D + d;
    const f4 a = *reinterpret_cast<const f4*>(o + D);
    *reinterpret_cast<f4*>(o) = t;
    *reinterpret_cast<f4*>(o + 2 * D) = t * a;
    *reinterpret_cast<f4*>(o + 3 * D) = t * b;
}

// backward glue 1: da = g1 + g2*text, db = g3*text, d_text = g0 + g2*a + g3*b
__global__ __launch_bounds__(256) void big_bwd_pre_kernel(const float* __restrict__ d_out, const float* __restrict__ out,
                                                          const float* __restrict__ text, const float* __restrict__ bsave,
                                                          float* __restrict__ da, float* __restrict__ db, float* __restrict__ d_text,
                                                          long rows, int D) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * (D / 4)) return;
    const long row = i / (D / 4);
    const int d = (i % (D / 4)) * 4;
    const float* g = d_out + row * 4 * D + d;
    const f4 g0 = *reinterpret_cast<const f4*>(g), g1 = *reinterpret_cast<const f4*>(g + D);
    const f4 g2 = *reinterpret_cast<const f4*>(g + 2 * D), g3 = *reinterpret_cast<const f4*>(g + 3 * D);
    const f4 t = *reinterpret_cast<const f4*>(text + row * D + d);
    const f4 a = *reinterpret_cast<const f4*>(out + row * 4 * D + D + d);
    const f4 b = *reinterpret_cast<const f4*>(bsave + row * D + d);
    *reinterpret_cast<f4*>(da + row * D + d) = g1 + g2 * t;
    *reinterpret_cast<f4*>(db + row * D + d) = g3 * t;
    *reinterpret_cast<f4*>(d_text + row * D + d) = g0 + g2 * a + g3 * b;
}

// P1 (in place over S) and P2 from S, the rank-1 terms, the masks and the saved softmax statistics
__global__ __launch_bounds__(256) void big_recompute_kernel(float* __restrict__ S, float* __restrict__ P2,
                                                            const float* __restrict__ rterm, const float* __restrict__ cterm,
                                                            const uint8_t* __restrict__ tmask, const uint8_t* __restrict__ mmask,
                                                            const float* __restrict__ row_stat, const float* __restrict__ col_stat,
                                                            long total, int T, int M) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const long row = i / M;           // b*T + t
    const int j = i % M;
    const long b = row / T;
    const float x = S[i] + rterm[row] + cterm[b * M + j];
    const float x1 = mmask[b * M + j] ? x : BIG_NEG, x2 = tmask[row] ? x : BIG_NEG;
    S[i] = expf(x1 - row_stat[row * 2]) / row_stat[row * 2 + 1];
    P2[i] = expf(x2 - col_stat[(b * M + j) * 2]) / col_stat[(b * M + j) * 2 + 1];
}

// rowdot[row] = sum_j X[row,j] * (Y ? Y[row,j] : 1)      (one wave per row)
__global__ __launch_bounds__(256) void big_rowdot_kernel(const float* __restrict__ X, const float* __restrict__ Y,
                                                         float* __restrict__ dst, long rows, int M) {
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    float acc = 0.f;
    for (int j = lane; j < M; j += 64) acc += X[(size_t)row * M + j] * (Y ? Y[(size_t)row * M + j] : 1.0f);
    acc = wave_sum(acc);
    if (lane == 0) dst[row] = acc;
}

// coldot[b,j] = sum_i X[b,i,j] * (Y ? Y[b,i,j] : 1)       (workgroup = 64 columns x 4 waves that take every 4th row)
__global__ __launch_bounds__(256) void big_coldot_kernel(const float* __restrict__ X, const float* __restrict__ Y,
                                                         float* __restrict__ dst, int T, int M) {
    __shared__ float pa[4][64];
    const int b = blockIdx.y, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int j = blockIdx.x * 64 + lane;
    const bool on = j < M;
    const size_t base = (size_t)b * T * M + (on ? j : 0);
    float acc = 0.f;
    for (int i0 = w; i0 < T; i0 += 16) {
        float x[4], y[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = i0 + 4 * q;
            x[q] = i < T ? X[base + (size_t)i * M] : 0.f;
            y[q] = (Y && i < T) ? Y[base + (size_t)i * M] : 1.0f;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) acc = fmaf(x[q], y[q], acc);
    }
    pa[w][lane] = acc;
    __syncthreads();
    if (w == 0 && on) dst[(size_t)b * M + j] = (pa[0][lane] + pa[1][lane]) + (pa[2][lane] + pa[3][lane]);
}

// dS = mmask_j * P1 * (dP1 - rowdot_i) + tmask_i * P2 * (dP2 - coldot_j), written over dP1
__global__ __launch_bounds__(256) void big_ds_kernel(const float* __restrict__ P1, const float* __restrict__ P2,
                                                     float* __restrict__ dP1, const float* __restrict__ dP2,
                                                     const float* __restrict__ rowdot, const float* __restrict__ coldot,
                                                     const uint8_t* __restrict__ tmask, const uint8_t* __restrict__ mmask,
                                                     long total, int T, int M) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const long row = i / M;
    const int j = i % M;
    const long b = row / T;
    float v = 0.f;
    if (mmask[b * M + j]) v += P1[i] * (dP1[i] - rowdot[row]);
    if (tmask[row]) v += P2[i] * (dP2[i] - coldot[b * M + j]);
    dP1[i] = v;
}

// dst[row, d] (= | +=) G[row, d] * w_tm[d] + s[row] * w[d]
__global__ __launch_bounds__(256) void big_post_kernel(const float* __restrict__ G, const float* __restrict__ w_tm,
                                                       const float* __restrict__ s, const float* __restrict__ w,
                                                       float* __restrict__ dst, int accumulate, long rows, int D) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * (D / 4)) return;
    const long row = i / (D / 4);
    const int d4 = i % (D / 4);
    f4 v = reinterpret_cast<const f4*>(G)[i] * reinterpret_cast<const f4*>(w_tm)[d4] + reinterpret_cast<const f4*>(w)[d4] * s[row];
    if (accumulate) v += reinterpret_cast<const f4*>(dst)[i];
    reinterpret_cast<f4*>(dst)[i] = v;
}

// out[d] += sum over a chunk of rows of X[row,d] * (Y ? Y[row,d] : 1) * (s ? s[row] : 1)   (out pre-zeroed)
__global__ __launch_bounds__(256) void big_colreduce_kernel(const float* __restrict__ X, const float* __restrict__ Y,
                                                            const float* __restrict__ s, float* __restrict__ out, long rows, int D) {
    const int d = blockIdx.x * 256 + threadIdx.x;
    if (d >= D) return;
    const long chunk = (rows + gridDim.y - 1) / gridDim.y;
    const long r0 = (long)blockIdx.y * chunk, r1 = min(rows, r0 + chunk);
    float acc = 0.f;
#pragma unroll 4
    for (long r = r0; r < r1; ++r) acc += X[(size_t)r * D + d] * (Y ? Y[(size_t)r * D + d] : 1.0f) * (s ? s[r] : 1.0f);
    atomicAdd(&out[d], acc);
}

// out[0] = sum of n values (single workgroup)
__global__ __launch_bounds__(256) void big_total_kernel(const float* __restrict__ x, float* __restrict__ out, long n) {
    __shared__ float part[4];
    float acc = 0.f;
    for (long i = threadIdx.x; i < n; i += 256) acc += x[i];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = part[0] + part[1] + part[2] + part[3];
}

// ------------------------------------------------------------------------------------------ host side
static size_t rup64(size_t x) { return (x + 63) / 64 * 64; }
struct BigFwdWs { size_t tw, S, P2, total; };
static BigFwdWs big_fwd_layout(int B, int T, int M, int D) {
    BigFwdWs w{};
    size_t o = 0;
    w.tw = o; o += rup64((size_t)B * T * D);
    w.S = o;  o += rup64((size_t)B * T * M);
    w.P2 = o; o += rup64((size_t)B * T * M);
    w.total = o;
    return w;
}
struct BigBwdWs { size_t tw, S, P2, dP1, dP2, da, db, dq, G, Hm, rowv, colv, dr, dc, total; };
static BigBwdWs big_bwd_layout(int B, int T, int M, int D) {
    BigBwdWs w{};
    size_t o = 0;
    const size_t btd = rup64((size_t)B * T * D), btm = rup64((size_t)B * T * M), bmd = rup64((size_t)B * M * D);
    w.tw = o; o += btd;
    w.S = o; o += btm;
    w.P2 = o; o += btm;
    w.dP1 = o; o += btm;
    w.dP2 = o; o += btm;
    w.da = o; o += btd;
    w.db = o; o += btd;
    w.dq = o; o += bmd;
    w.G = o; o += btd;
    w.Hm = o; o += bmd;
    w.rowv = o; o += rup64((size_t)B * T);
    w.colv = o; o += rup64((size_t)B * M);
    w.dr = o; o += rup64((size_t)B * T);
    w.dc = o; o += rup64((size_t)B * M);
    w.total = o;
    return w;
}
size_t bidaf_big_fwd_ws_floats(int B, int T, int M, int D) { return big_fwd_layout(B, T, M, D).total; }
size_t bidaf_big_bwd_ws_floats(int B, int T, int M, int D) { return big_bwd_layout(B, T, M, D).total; }

#define BIG_EW(kernel, n, ...)                                                                   \
    do {                                                                                         \
        hipLaunchKernelGGL(kernel, dim3(((n) + 255) / 256), dim3(256), 0, stream, __VA_ARGS__);  \
        MMB_HIP(hipGetLastError());                                                              \
    } while (0)

// S (B,T,M) = (text_d * w_tm) . mod_d^T, into ws
static int big_similarity(const float* text_d, const float* mod_d, const float* w_tm, float* tw, float* S, int B, int T, int M,
                          int D, hipStream_t stream) {
    BIG_EW(big_scale_kernel, (long)B * T * (D / 4), text_d, w_tm, tw, (long)B * T, D);
    return bgemm(tw, mod_d, S, T, M, D, D, D, M, 0, 1, 0, B, (long)T * D, (long)M * D, (long)T * M, stream);
}

int bidaf_big_fwd(const float* text, const float* mod, const uint8_t* text_mask, const uint8_t* mod_mask, const float* text_d,
                  const float* mod_d, const float* w_tm, float* out, float* q, float* bsave, const float* rterm,
                  const float* cterm, float* row_stat, float* col_stat, float* ws, int B, int T, int M, int D, hipStream_t stream) {
    const BigFwdWs L = big_fwd_layout(B, T, M, D);
    float* tw = ws + L.tw;
    float* S = ws + L.S;
    float* P2 = ws + L.P2;
    if (int rc = big_similarity(text_d, mod_d, w_tm, tw, S, B, T, M, D, stream)) return rc;
    {
        ProfScope ps_(MMB_K_ATT_COL, stream);
        hipLaunchKernelGGL(big_colsoft_kernel, dim3((M + 63) / 64, B), dim3(256), 0, stream, S, rterm, cterm, text_mask, P2, col_stat, T, M);
        MMB_HIP(hipGetLastError());
    }
    {
        ProfScope ps_(MMB_K_ATT_ROW, stream);
        hipLaunchKernelGGL(big_rowsoft_kernel, dim3(((long)B * T + 3) / 4), dim3(256), 0, stream, S, rterm, cterm, mod_mask, row_stat,
                           (long)B * T, T, M);
        MMB_HIP(hipGetLastError());
    }
    const long sTD = (long)T * D, sMD = (long)M * D, sTM = (long)T * M;
    // q = P2^T . text;  a = P1 . mod (straight into out[:, D:2D]);  b = P1 . q
    if (int rc = bgemm(P2, text, q, M, D, T, M, D, D, 1, 0, 0, B, sTM, sTD, sMD, stream)) return rc;
    if (int rc = bgemm(S, mod, out + D, T, D, M, M, D, 4 * D, 0, 0, 0, B, sTM, sMD, 4 * sTD, stream)) return rc;
    if (int rc = bgemm(S, q, bsave, T, D, M, M, D, D, 0, 0, 0, B, sTM, sMD, sTD, stream)) return rc;
    BIG_EW(big_assemble_kernel, (long)B * T * (D / 4), text, bsave, out, (long)B * T, D);
    return MMB_OK;
}

int bidaf_big_bwd(const float* d_out, const float* out, const float* text, const float* mod, const uint8_t* text_mask,
                  const uint8_t* mod_mask, const float* text_d, const float* mod_d, const float* w_t, const float* w_m,
                  const float* w_tm, const float* q, const float* bsave, const float* rterm, const float* cterm,
                  const float* row_stat, const float* col_stat, float* d_text, float* d_mod, float* d_text_d, float* d_mod_d,
                  float* d_w_t, float* d_w_m, float* d_w_tm, float* d_bias, float* ws, int B, int T, int M, int D, hipStream_t stream) {
    const BigBwdWs L = big_bwd_layout(B, T, M, D);
    float* tw = ws + L.tw; float* P1 = ws + L.S; float* P2 = ws + L.P2; float* dP1 = ws + L.dP1; float* dP2 = ws + L.dP2;
    float* da = ws + L.da; float* db = ws + L.db; float* dq = ws + L.dq; float* G = ws + L.G; float* Hm = ws + L.Hm;
    float* rowv = ws + L.rowv; float* colv = ws + L.colv; float* dr = ws + L.dr; float* dc = ws + L.dc;
    const bool drop = text_d != nullptr;
    const float* td = drop ? text_d : text;
    const float* md = drop ? mod_d : mod;
    const long BT = (long)B * T, BM = (long)B * M, total = BT * M;
    const long sTD = (long)T * D, sMD = (long)M * D, sTM = (long)T * M;

    BIG_EW(big_bwd_pre_kernel, BT * (D / 4), d_out, out, text, bsave, da, db, d_text, BT, D);
    if (int rc = big_similarity(td, md, w_tm, tw, P1, B, T, M, D, stream)) return rc;
    BIG_EW(big_recompute_kernel, total, P1, P2, rterm, cterm, text_mask, mod_mask, row_stat, col_stat, total, T, M);
    // dP1 = da . mod^T + db . q^T
    if (int rc = bgemm(da, mod, dP1, T, M, D, D, D, M, 0, 1, 0, B, sTD, sMD, sTM, stream)) return rc;
    if (int rc = bgemm(db, q, dP1, T, M, D, D, D, M, 0, 1, 1, B, sTD, sMD, sTM, stream)) return rc;
    // d_mod (clean path) = P1^T . da;  dq = P1^T . db
    if (int rc = bgemm(P1, da, d_mod, M, D, T, M, D, D, 1, 0, 0, B, sTM, sTD, sMD, stream)) return rc;
    if (int rc = bgemm(P1, db, dq, M, D, T, M, D, D, 1, 0, 0, B, sTM, sTD, sMD, stream)) return rc;
    // dP2 = text . dq^T;  d_text += P2 . dq
    if (int rc = bgemm(text, dq, dP2, T, M, D, D, D, M, 0, 1, 0, B, sTD, sMD, sTM, stream)) return rc;
    if (int rc = bgemm(P2, dq, d_text, T, D, M, M, D, D, 0, 0, 1, B, sTM, sMD, sTD, stream)) return rc;
    // softmax backward of both axes -> dS (over dP1)
    hipLaunchKernelGGL(big_rowdot_kernel, dim3((BT + 3) / 4), dim3(256), 0, stream, P1, dP1, rowv, BT, M);
    hipLaunchKernelGGL(big_coldot_kernel, dim3((M + 63) / 64, B), dim3(256), 0, stream, P2, dP2, colv, T, M);
    MMB_HIP(hipGetLastError());
    BIG_EW(big_ds_kernel, total, P1, P2, dP1, dP2, rowv, colv, text_mask, mod_mask, total, T, M);
    float* dS = dP1;
    hipLaunchKernelGGL(big_rowdot_kernel, dim3((BT + 3) / 4), dim3(256), 0, stream, dS, static_cast<const float*>(nullptr), dr, BT, M);
    hipLaunchKernelGGL(big_coldot_kernel, dim3((M + 63) / 64, B), dim3(256), 0, stream, dS, static_cast<const float*>(nullptr), dc, T, M);
    MMB_HIP(hipGetLastError());
    // G = dS . mod_d (T x D), Hm = dS^T . text_d (M x D): the bilinear term without w_tm
    if (int rc = bgemm(dS, md, G, T, D, M, M, D, D, 0, 0, 0, B, sTM, sMD, sTD, stream)) return rc;
    if (int rc = bgemm(dS, td, Hm, M, D, T, M, D, D, 1, 0, 0, B, sTM, sTD, sMD, stream)) return rc;
    // similarity-path input gradients: into the dropped copies' gradients, or folded into d_text / d_mod
    BIG_EW(big_post_kernel, BT * (D / 4), G, w_tm, dr, w_t, drop ? d_text_d : d_text, drop ? 0 : 1, BT, D);
    BIG_EW(big_post_kernel, BM * (D / 4), Hm, w_tm, dc, w_m, drop ? d_mod_d : d_mod, drop ? 0 : 1, BM, D);
    // parameter gradients
    MMB_HIP(hipMemsetAsync(d_w_t, 0, sizeof(float) * D, stream));
    MMB_HIP(hipMemsetAsync(d_w_m, 0, sizeof(float) * D, stream));
    MMB_HIP(hipMemsetAsync(d_w_tm, 0, sizeof(float) * D, stream));
    const float* none = nullptr;
    hipLaunchKernelGGL(big_colreduce_kernel, dim3((D + 255) / 256, 256), dim3(256), 0, stream, td, G, none, d_w_tm, BT, D);
    hipLaunchKernelGGL(big_colreduce_kernel, dim3((D + 255) / 256, 256), dim3(256), 0, stream, td, none, static_cast<const float*>(dr), d_w_t, BT, D);
    hipLaunchKernelGGL(big_colreduce_kernel, dim3((D + 255) / 256, 256), dim3(256), 0, stream, md, none, static_cast<const float*>(dc), d_w_m, BM, D);
    hipLaunchKernelGGL(big_total_kernel, dim3(1), dim3(256), 0, stream, dr, d_bias, BT);
    MMB_HIP(hipGetLastError());
    return MMB_OK;
}

}  // namespace mmb
