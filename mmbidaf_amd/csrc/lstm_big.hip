// General-size LSTM recurrence (H > MMB_LSTM_MAX_H, e.g. BASELINE cfg5's H = 512): W_hh (4H x H fp32, 4 MiB at H = 512)
// no longer fits one workgroup's registers, so the time loop runs as ONE KERNEL LAUNCH PER STEP over all
// (encoder, direction, sample, unit) of the group -- no grid-wide spin barrier, the stream order is the barrier.
// Same semantics and the same saved-tensor layouts as the register-resident path (lstm.hip), so the hoisted input
// projection and the gradient GEMMs are shared:  reference layers/encoding.py:79-81,93-99 (packed nn.LSTM).
//
//   forward step s   sample b (if s < len[b]) handles t = s (forward direction) or len[b]-1-s (reverse);
//                    h_{prev} is read from y[b, t -+ 1], c_{prev} from cs[b, t -+ 1]; a workgroup owns 32 units x 8
//                    samples, stages the 8 h vectors in LDS and streams W_hh^T (pre-transposed once per call to
//                    [k][u][gate], so a thread's four gate weights are one coalesced float4).
//   BPTT step s      visits the forward order backwards; dh = d_y[t] + d_a[step before] . W_hh (rows of W_hh are
//                    already contiguous in the output unit), then the gate gradients, d_a[b,t] and the running dc.
#include "common.h"

namespace mmb {

constexpr int BIG_UT = 32, BIG_ST = 8;   // units x samples per workgroup (128 threads: 32 units x 4 sample pairs)

struct BigFwdProb {
    const float* gx;        // (B,T,2,H,4)
    const float* whhT[2];   // (H, H, 4): [k][u][gate]
    const int* len;
    float* y; float* gates; float* cs; float* h_n; float* c_n;
    int B, T, H;
};
struct BigFwdArgs { BigFwdProb p[MMB_MAX_GROUP]; int n; };

// W_hh (4H,H) [g*H+u][k]  ->  [k][u][g]
__global__ __launch_bounds__(256) void lstm_big_transpose_kernel(const float* __restrict__ w, float* __restrict__ wt, int H) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;   // over (k, u)
    if (idx >= (long)H * H) return;
    const int k = idx / H, u = idx % H;
    f4 v;
#pragma unroll
    for (int g = 0; g < 4; ++g) v[g] = w[((size_t)g * H + u) * H + k];
    *reinterpret_cast<f4*>(wt + idx * 4) = v;
}

__global__ __launch_bounds__(128) void lstm_big_fwd_step_kernel(const BigFwdArgs args, const int s) {
    extern __shared__ __attribute__((aligned(16))) float hs[];   // [H][BIG_ST]
    const BigFwdProb& P = args.p[blockIdx.z >> 1];
    const int dir = blockIdx.z & 1;
    const int H = P.H, T = P.T;
    const int u0 = blockIdx.x * BIG_UT, b0 = blockIdx.y * BIG_ST;
    if (u0 >= H || b0 >= P.B || s >= T) return;
    const int tid = threadIdx.x, ul = tid & 31, sg = tid >> 5;

    // stage h_{prev} of the 8 samples (zeros for a sample's first step and for finished / absent samples)
    for (int i = tid; i < BIG_ST * H; i += 128) {
        const int sl = i / H, k = i - sl * H, b = b0 + sl;
        float v = 0.f;
        if (b < P.B) {
            const int len = min(max(P.len[b], 0), T);
            if (s > 0 && s < len) {
                const int tp = dir ? len - s : s - 1;   // the step this direction processed before
                v = P.y[((size_t)b * T + tp) * 2 * H + dir * H + k];
            }
        }
        hs[k * BIG_ST + sl] = v;
    }
    __syncthreads();
    const int u = min(u0 + ul, H - 1);
    const float* wt = P.whhT[dir] + (size_t)u * 4;
    f4 acc0 = f4{0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
#pragma unroll 4
    for (int k = 0; k < H; ++k) {
        const f4 w = *reinterpret_cast<const f4*>(wt + (size_t)k * 4 * H);
        const f2 hv = *reinterpret_cast<const f2*>(&hs[k * BIG_ST + 2 * sg]);
        acc0 += w * hv.x;
        acc1 += w * hv.y;
    }
    if (u0 + ul >= H) return;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int b = b0 + 2 * sg + e;
        if (b >= P.B) continue;
        const int len = min(max(P.len[b], 0), T);
        if (s >= len) continue;
        const int t = dir ? len - 1 - s : s;
        const size_t row = (size_t)b * T + t;
        const f4 pre = (e ? acc1 : acc0) + *reinterpret_cast<const f4*>(P.gx + (row * 2 + dir) * 4 * H + (size_t)u * 4);
        float c_prev = 0.f;
        if (s > 0) c_prev = P.cs[((size_t)b * T + (dir ? t + 1 : t - 1)) * 2 * H + dir * H + u];
        const float gi = sigmoidf_(pre.x), gf = sigmoidf_(pre.y), gg = tanhf_(pre.z), go = sigmoidf_(pre.w);
        const float c = fmaf(gf, c_prev, gi * gg);
        const float h = go * tanhf_(c);
        *reinterpret_cast<f4*>(P.gates + (row * 2 + dir) * 4 * H + (size_t)u * 4) = f4{gi, gf, gg, go};
        P.cs[row * 2 * H + dir * H + u] = c;
        P.y[row * 2 * H + dir * H + u] = h;
        if (s == len - 1) {
            P.h_n[((size_t)dir * P.B + b) * H + u] = h;
            P.c_n[((size_t)dir * P.B + b) * H + u] = c;
        }
    }
}

// rows t >= len[b] of a (B,T,W) tensor := 0   (y: pad_packed_sequence zeros, encoding.py:99;  d_a: dead steps)
__global__ __launch_bounds__(256) void lstm_big_zero_tail_kernel(float* __restrict__ p, const int* __restrict__ lens, int B, int T, int W) {
    const int b = blockIdx.y;
    const int len = min(max(lens[b], 0), T);
    const long n = (long)(T - len) * W;
    float* base = p + ((size_t)b * T + len) * W;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) base[i] = 0.f;
}

// samples with len == 0 never reach a last step: their h_n / c_n are zero
__global__ __launch_bounds__(256) void lstm_big_empty_state_kernel(float* __restrict__ h_n, float* __restrict__ c_n,
                                                                   const int* __restrict__ lens, int B, int H) {
    const int idx = blockIdx.x * 256 + threadIdx.x;   // over (dir, b, u)
    if (idx >= 2 * B * H) return;
    const int b = (idx / H) % B;
    if (lens[b] <= 0) { h_n[idx] = 0.f; c_n[idx] = 0.f; }
}

struct BigBwdProb {
    const float* d_y; const float* d_hn; const float* gates; const float* cs;
    const float* w_hh[2];   // (4H,H)
    const int* len;
    float* d_a;             // (B,T,2,4H) torch gate order
    float* dc;              // (2,B,H) running cell-state gradient
    int B, T, H;
};
struct BigBwdArgs { BigBwdProb p[MMB_MAX_GROUP]; int n; };

__global__ __launch_bounds__(128) void lstm_big_bwd_step_kernel(const BigBwdArgs args, const int s) {
    extern __shared__ __attribute__((aligned(16))) float das[];   // [4H][BIG_ST]
    const BigBwdProb& P = args.p[blockIdx.z >> 1];
    const int dir = blockIdx.z & 1;
    const int H = P.H, T = P.T;
    const int u0 = blockIdx.x * BIG_UT, b0 = blockIdx.y * BIG_ST;
    if (u0 >= H || b0 >= P.B || s >= T) return;
    const int tid = threadIdx.x, ul = tid & 31, sg = tid >> 5;

    // BPTT step s of sample b handles t = len-1-s (forward direction) or s (reverse); the step before it handled
    // t+1 (forward) / t-1 (reverse): stage that step's d_a (4H values per sample)
    for (int i = tid; i < BIG_ST * 4 * H; i += 128) {
        const int sl = i / (4 * H), r = i - sl * 4 * H, b = b0 + sl;
        float v = 0.f;
        if (b < P.B) {
            const int len = min(max(P.len[b], 0), T);
            if (s > 0 && s < len) {
                const int tn = dir ? s - 1 : len - s;
                v = P.d_a[((size_t)b * T + tn) * 8 * H + (size_t)dir * 4 * H + r];
            }
        }
        das[r * BIG_ST + sl] = v;
    }
    __syncthreads();
    const int u = min(u0 + ul, H - 1);
    const float* w = P.w_hh[dir] + u;
    float acc0 = 0.f, acc1 = 0.f;
#pragma unroll 4
    for (int r = 0; r < 4 * H; ++r) {
        const float wv = w[(size_t)r * H];
        const f2 dv = *reinterpret_cast<const f2*>(&das[r * BIG_ST + 2 * sg]);
        acc0 = fmaf(wv, dv.x, acc0);
        acc1 = fmaf(wv, dv.y, acc1);
    }
    if (u0 + ul >= H) return;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int b = b0 + 2 * sg + e;
        if (b >= P.B) continue;
        const int len = min(max(P.len[b], 0), T);
        if (s >= len) continue;
        const int t = dir ? s : len - 1 - s;
        const size_t row = (size_t)b * T + t;
        const size_t st = ((size_t)dir * P.B + b) * H + u;
        float dh = (e ? acc1 : acc0) + P.d_y[row * 2 * H + dir * H + u];
        float dc = 0.f;
        if (s == 0) { if (P.d_hn) dh += P.d_hn[st]; }
        else dc = P.dc[st];
        const f4 g4 = *reinterpret_cast<const f4*>(P.gates + (row * 2 + dir) * 4 * H + (size_t)u * 4);
        const float c_t = P.cs[row * 2 * H + dir * H + u];
        // c of the step the FORWARD recurrence ran before t: t-1 (forward direction), t+1 (reverse); zero at its start
        const bool has_prev = dir ? (t + 1 < len) : (t > 0);
        const float c_prev = has_prev ? P.cs[((size_t)b * T + (dir ? t + 1 : t - 1)) * 2 * H + dir * H + u] : 0.f;
        const float gi = g4.x, gf = g4.y, gg = g4.z, go = g4.w;
        const float tc = tanhf_(c_t);
        const float dc_t = fmaf(dh * go, 1.0f - tc * tc, dc);
        float* da = P.d_a + row * 8 * H + (size_t)dir * 4 * H + u;
        da[0] = dc_t * gg * gi * (1.0f - gi);
        da[(size_t)H] = dc_t * c_prev * gf * (1.0f - gf);
        da[(size_t)2 * H] = dc_t * gi * (1.0f - gg * gg);
        da[(size_t)3 * H] = dh * tc * go * (1.0f - go);
        P.dc[st] = dc_t * gf;
    }
}

// d_b (2,4H) = column sums of d_a (B*T, 8H): one thread per column, rows split over blockIdx.y with atomics
__global__ __launch_bounds__(256) void lstm_big_colsum_kernel(const float* __restrict__ d_a, float* __restrict__ d_b, long rows, int cols) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= cols) return;
    const long chunk = (rows + gridDim.y - 1) / gridDim.y;
    const long r0 = (long)blockIdx.y * chunk, r1 = min(rows, r0 + chunk);
    float acc = 0.f;
    for (long r = r0; r < r1; ++r) acc += d_a[(size_t)r * cols + c];
    atomicAdd(&d_b[c], acc);
}

// ------------------------------------------------------------------------------------------ host side
size_t lstm_big_fwd_ws_bytes(int H) { return (size_t)2 * 4 * H * H * sizeof(float); }          // W_hh^T, both directions
size_t lstm_big_bwd_ws_bytes(int B, int H) { return (size_t)2 * B * H * sizeof(float); }     // running dc

int lstm_big_fwd(const mmb_lstm_fwd_desc* d, int n, char* const* big_ws, hipStream_t stream) {
    BigFwdArgs a{};
    a.n = n;
    const int H = d[0].H;
    int maxT = 0, maxB = 0;
    for (int i = 0; i < n; ++i) {
        const mmb_lstm_fwd_desc& p = d[i];
        float* wt = reinterpret_cast<float*>(big_ws[i]);
        for (int dir = 0; dir < 2; ++dir) {
            float* dst = wt + (size_t)dir * 4 * H * H;
            hipLaunchKernelGGL(lstm_big_transpose_kernel, dim3(((long)H * H + 255) / 256), dim3(256), 0, stream, p.w_hh[dir], dst, H);
            a.p[i].whhT[dir] = dst;
        }
        BigFwdProb& q = a.p[i];
        q.gx = p.gx; q.len = p.lengths; q.y = p.y; q.gates = p.gates; q.cs = p.cs; q.h_n = p.h_n; q.c_n = p.c_n;
        q.B = p.B; q.T = p.T; q.H = H;
        maxT = max(maxT, p.T);
        maxB = max(maxB, p.B);
    }
    MMB_HIP(hipGetLastError());
    const size_t lds = (size_t)BIG_ST * H * sizeof(float);
    MMB_REQUIRE(lds <= 160 * 1024, "H=%d too large for the general recurrence", H);
    static bool attr = false;
    if (!attr) {
        MMB_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(lstm_big_fwd_step_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr = true;
    }
    const dim3 grid((H + BIG_UT - 1) / BIG_UT, (maxB + BIG_ST - 1) / BIG_ST, 2 * n);
    {
        ProfScope ps_(MMB_K_LSTM_REC_FWD, stream);
        for (int s = 0; s < maxT; ++s) hipLaunchKernelGGL(lstm_big_fwd_step_kernel, grid, dim3(128), lds, stream, a, s);
    }
    MMB_HIP(hipGetLastError());
    for (int i = 0; i < n; ++i) {
        const mmb_lstm_fwd_desc& p = d[i];
        hipLaunchKernelGGL(lstm_big_zero_tail_kernel, dim3(64, p.B), dim3(256), 0, stream, p.y, p.lengths, p.B, p.T, 2 * H);
        hipLaunchKernelGGL(lstm_big_empty_state_kernel, dim3((2 * p.B * H + 255) / 256), dim3(256), 0, stream, p.h_n, p.c_n, p.lengths, p.B, H);
    }
    MMB_HIP(hipGetLastError());
    return MMB_OK;
}

int lstm_big_bwd(const mmb_lstm_bwd_desc* d, int n, char* const* big_ws, hipStream_t stream) {
    BigBwdArgs a{};
    a.n = n;
    const int H = d[0].H;
    int maxT = 0, maxB = 0;
    for (int i = 0; i < n; ++i) {
        const mmb_lstm_bwd_desc& p = d[i];
        BigBwdProb& q = a.p[i];
        q.d_y = p.d_y; q.d_hn = p.d_hn; q.gates = p.gates; q.cs = p.cs; q.w_hh[0] = p.w_hh[0]; q.w_hh[1] = p.w_hh[1];
        q.len = p.lengths; q.d_a = p.d_a; q.dc = reinterpret_cast<float*>(big_ws[i]);
        q.B = p.B; q.T = p.T; q.H = H;
        maxT = max(maxT, p.T);
        maxB = max(maxB, p.B);
    }
    const size_t lds = (size_t)BIG_ST * 4 * H * sizeof(float);
    MMB_REQUIRE(lds <= 160 * 1024, "H=%d too large for the general recurrence", H);
    static bool attr = false;
    if (!attr) {
        MMB_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(lstm_big_bwd_step_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr = true;
    }
    const dim3 grid((H + BIG_UT - 1) / BIG_UT, (maxB + BIG_ST - 1) / BIG_ST, 2 * n);
    {
        ProfScope ps_(MMB_K_LSTM_REC_BWD, stream);
        for (int s = 0; s < maxT; ++s) hipLaunchKernelGGL(lstm_big_bwd_step_kernel, grid, dim3(128), lds, stream, a, s);
    }
    MMB_HIP(hipGetLastError());
    for (int i = 0; i < n; ++i) {
        const mmb_lstm_bwd_desc& p = d[i];
        hipLaunchKernelGGL(lstm_big_zero_tail_kernel, dim3(64, p.B), dim3(256), 0, stream, p.d_a, p.lengths, p.B, p.T, 8 * H);
        MMB_HIP(hipMemsetAsync(p.d_b, 0, sizeof(float) * 8 * H, stream));
        hipLaunchKernelGGL(lstm_big_colsum_kernel, dim3((8 * H + 255) / 256, 64), dim3(256), 0, stream, p.d_a, p.d_b, (long)p.B * p.T, 8 * H);
    }
    MMB_HIP(hipGetLastError());
    return MMB_OK;
}

}  // namespace mmb
