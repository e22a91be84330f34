// Bidirectional LSTM layer for gfx950: hoisted input projection (MFMA GEMM) + a register-resident
// recurrence kernel, forward and BPTT.  Replaces torch.nn.LSTM on a packed batch as called by
// the reference RNNEncoder (layers/encoding.py:79-81,96).
//
// Recurrence design (H <= 128, fp32): the time loop is a chain of T dependent (1 x H).(H x 4H)
// products, so it is latency-bound, and fp32 MFMA has no rate advantage over the vector ALU on
// gfx950.  One workgroup therefore owns ONE (encoder, direction, sample) chain and keeps the whole
// W_hh (4H x H, 160 KB at H = 100) in its VGPRs for all T steps:
//   thread (u, kq), u = hidden unit, kq = lane-in-quad, holds W_hh[g*H+u][kq*KQ .. kq*KQ+KQ) for the
//   four gates g = i,f,g,o  (4*KQ floats, KQ = H/4 rounded up).
// Per step a thread reads its quarter of h_{t-1} from LDS (broadcast b128 reads), does 4*KQ FMAs,
// and a transposing quad reduction (DPP) leaves gate kq of unit u in lane kq.  Each lane applies
// its own activation, the quad exchanges the four gates by DPP broadcast and updates (c, h)
// redundantly; h_t goes to a double-buffered LDS vector: ONE barrier per time step.
// Up to MMB_MAX_GROUP independent encoders x 2 directions x B samples are co-scheduled in a single
// launch (models.py:97,102,113 are independent; so are models.py:134,135).
//
// Packed-sequence semantics (pack_padded_sequence / pad_packed_sequence, encoding.py:93,99): sample
// b runs exactly len[b] steps, the reverse direction starts at t = len-1, y is 0 for t >= len, and
// h_n is the state after the sample's own last step.
#include "common.h"

namespace mmb {

constexpr int PF = 4;  // software prefetch distance (time steps) for the streamed per-step operands

struct RecFwdProb {
    const float* gx;       // (B,T,2,H,4)
    const float* w_hh[2];  // (4H,H)
    const int* len;        // (B)
    float* y;              // (B,T,2H)
    float* gates;          // (B,T,2,H,4)
    float* cs;             // (B,T,2,H)
    float* h_n;            // (2,B,H)
    float* c_n;            // (2,B,H)
    int B, T, H, wg_begin;
};
struct RecFwdArgs {
    RecFwdProb p[MMB_MAX_GROUP];
    int n;
};

template <int KQ>
__global__ __launch_bounds__(512) void lstm_rec_fwd_kernel(const RecFwdArgs args) {
    constexpr int KQP = (KQ + 3) & ~3;
    __shared__ __attribute__((aligned(16))) float hbuf[2][4][KQP];

    int pi = 0;
    for (int i = 1; i < args.n; ++i)
        if ((int)blockIdx.x >= args.p[i].wg_begin) pi = i;
    const RecFwdProb& P = args.p[pi];
    const int local = blockIdx.x - P.wg_begin;
    const int dir = local / P.B, b = local % P.B;
    const int H = P.H, T = P.T;
    const int len = min(max(P.len[b], 0), T);
    const int tid = threadIdx.x, u = tid >> 2, kq = tid & 3;
    const bool live = u < H;
    const int uu = live ? u : 0;

    // ---- W_hh slice into registers
    float w[4][KQ];
    {
        const float* W = P.w_hh[dir];
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int kk = 0; kk < KQ; ++kk) {
                const int k = kq * KQ + kk;
                w[g][kk] = (live && k < H) ? W[(size_t)(g * H + uu) * H + k] : 0.f;
            }
    }
    for (int i = tid; i < 2 * 4 * KQP; i += blockDim.x) (&hbuf[0][0][0])[i] = 0.f;

    const size_t row0 = (size_t)b * T;  // row (b,t) = row0 + t
    const float* gxp = P.gx + (size_t)dir * 4 * H + (size_t)uu * 4 + kq;
    float* gatesp = P.gates + (size_t)dir * 4 * H + (size_t)uu * 4 + kq;
    float* csp = P.cs + (size_t)dir * H + uu;
    float* yp = P.y + (size_t)dir * H + uu;
    const int rev = dir;
    auto tof = [&](int s) { return rev ? (len - 1 - s) : s; };

    float gxr[PF];
#pragma unroll
    for (int j = 0; j < PF; ++j) gxr[j] = (live && j < len) ? gxp[(row0 + tof(j)) * 8 * H] : 0.f;

    float c = 0.f, h = 0.f;
    const bool is_tanh = kq == 2;
    __syncthreads();

    int cur = 0;
    for (int s0 = 0; s0 < len; s0 += PF) {
#pragma unroll
        for (int j = 0; j < PF; ++j) {
            const int s = s0 + j;
            if (s < len) {  // block-uniform
                const int t = tof(s);
                const float gx = gxr[j];
                if (live && s + PF < len) gxr[j] = gxp[(row0 + tof(s + PF)) * 8 * H];
                // matvec: my quarter of h against my 4 gate rows
                float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
                const float* hq = &hbuf[cur][kq][0];
#pragma unroll
                for (int k4 = 0; k4 < KQP; k4 += 4) {
                    const f4 hv = *reinterpret_cast<const f4*>(hq + k4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (k4 + e < KQ) {
                            a0 = fmaf(w[0][k4 + e], hv[e], a0);
                            a1 = fmaf(w[1][k4 + e], hv[e], a1);
                            a2 = fmaf(w[2][k4 + e], hv[e], a2);
                            a3 = fmaf(w[3][k4 + e], hv[e], a3);
                        }
                    }
                }
                // transposing quad reduction: lane kq ends with the full sum of gate kq
                const bool b0 = kq & 1, b1 = kq & 2;
                const float keepA = b0 ? a1 : a0, sendA = b0 ? a0 : a1;
                const float keepB = b0 ? a3 : a2, sendB = b0 ? a2 : a3;
                const float rA = keepA + quad_xor1(sendA);  // gate b0      over lanes {l, l^1}
                const float rB = keepB + quad_xor1(sendB);  // gate 2 + b0
                const float keep = b1 ? rB : rA, send = b1 ? rA : rB;
                const float pre = keep + quad_xor2(send) + gx;
                // activation of my gate: sigmoid for i,f,o; tanh for g
                const float sg = fast_rcp(1.0f + __expf(is_tanh ? -2.0f * pre : -pre));
                const float act = is_tanh ? 2.0f * sg - 1.0f : sg;
                const float gi = quad_bcast<0>(act), gf = quad_bcast<1>(act);
                const float gg = quad_bcast<2>(act), go = quad_bcast<3>(act);
                c = fmaf(gf, c, gi * gg);
                h = go * tanhf_(c);
                if (live) {
                    const size_t row = row0 + t;
                    gatesp[row * 8 * H] = act;
                    if (kq == 0) {
                        hbuf[cur ^ 1][u / KQ][u % KQ] = h;
                    } else if (kq == 1) {
                        yp[row * 2 * H] = h;
                    } else if (kq == 2) {
                        csp[row * 2 * H] = c;
                    }
                }
                cur ^= 1;
                __syncthreads();
            }
        }
    }
    if (live) {
        if (kq == 0) P.h_n[((size_t)dir * P.B + b) * H + u] = h;
        if (kq == 1) P.c_n[((size_t)dir * P.B + b) * H + u] = c;
    }
    // zero the padded tail of y (pad_packed_sequence, encoding.py:99)
    for (int i = tid; i < (T - len) * H; i += blockDim.x) {
        const int t = len + i / H, k = i % H;
        P.y[(row0 + t) * 2 * H + dir * H + k] = 0.f;
    }
}

// ------------------------------------------------------------------------------------------ BPTT
struct RecBwdProb {
    const float* d_y;      // (B,T,2H)
    const float* d_hn;     // (2,B,H) or null
    const float* gates;    // (B,T,2,H,4)
    const float* cs;       // (B,T,2,H)
    const float* w_hh[2];  // (4H,H)
    const int* len;
    float* d_a;            // (B,T,2,4H) torch gate order
    float* d_b;            // (2,4H) accumulated with atomics (pre-zeroed)
    int B, T, H, wg_begin;
};
struct RecBwdArgs {
    RecBwdProb p[MMB_MAX_GROUP];
    int n;
};

// thread (u, kq) holds column u of gate block kq of W_hh: wT[kk] = W_hh[kq*H + kk][u]
template <int KQ>
__global__ __launch_bounds__(512) void lstm_rec_bwd_kernel(const RecBwdArgs args) {
    constexpr int HP = 4 * KQ;  // >= H, multiple of 4
    __shared__ __attribute__((aligned(16))) float dabuf[2][4][HP];

    int pi = 0;
    for (int i = 1; i < args.n; ++i)
        if ((int)blockIdx.x >= args.p[i].wg_begin) pi = i;
    const RecBwdProb& P = args.p[pi];
    const int local = blockIdx.x - P.wg_begin;
    const int dir = local / P.B, b = local % P.B;
    const int H = P.H, T = P.T;
    const int len = min(max(P.len[b], 0), T);
    const int tid = threadIdx.x, u = tid >> 2, kq = tid & 3;
    const bool live = u < H;
    const int uu = live ? u : 0;

    float wT[HP];
    {
        const float* W = P.w_hh[dir];
#pragma unroll
        for (int kk = 0; kk < HP; ++kk) wT[kk] = (live && kk < H) ? W[(size_t)(kq * H + kk) * H + uu] : 0.f;
    }
    for (int i = tid; i < 2 * 4 * HP; i += blockDim.x) (&dabuf[0][0][0])[i] = 0.f;

    const size_t row0 = (size_t)b * T;
    const int rev = dir;
    // BPTT visits the forward processing order backwards: fwd dir t = len-1..0, reverse dir t = 0..len-1
    auto tof = [&](int s) { return rev ? s : (len - 1 - s); };
    const float* gatesp = P.gates + (size_t)dir * 4 * H + (size_t)uu * 4;
    const float* csp = P.cs + (size_t)dir * H + uu;
    const float* dyp = P.d_y + (size_t)dir * H + uu;
    float* dap = P.d_a + (size_t)dir * 4 * H + (size_t)kq * H + uu;

    f4 gr[PF];
    float cpr[PF], dyr[PF];  // c_{prev(t)} and d_y[t]
    auto load_step = [&](int s, f4& g4, float& cp, float& dyv) {
        const int t = tof(s);
        g4 = *reinterpret_cast<const f4*>(gatesp + (row0 + t) * 8 * H);
        const int tp = rev ? t + 1 : t - 1;  // step processed before t by the forward recurrence
        cp = (tp >= 0 && tp < len) ? csp[(row0 + tp) * 2 * H] : 0.f;
        dyv = dyp[(row0 + t) * 2 * H];
    };
#pragma unroll
    for (int j = 0; j < PF; ++j) {
        gr[j] = f4{0.f, 0.f, 0.f, 0.f};
        cpr[j] = 0.f;
        dyr[j] = 0.f;
        if (live && j < len) load_step(j, gr[j], cpr[j], dyr[j]);
    }
    float c_t = (live && len > 0) ? csp[(row0 + tof(0)) * 2 * H] : 0.f;
    float dh = (live && P.d_hn) ? P.d_hn[((size_t)dir * P.B + b) * H + u] : 0.f;
    float dc = 0.f;
    float db_acc = 0.f;
    __syncthreads();

    int cur = 0;
    for (int s0 = 0; s0 < len; s0 += PF) {
#pragma unroll
        for (int j = 0; j < PF; ++j) {
            const int s = s0 + j;
            if (s < len) {
                const int t = tof(s);
                const f4 g4 = gr[j];
                const float c_prev = cpr[j], dyv = dyr[j];
                if (live && s + PF < len) load_step(s + PF, gr[j], cpr[j], dyr[j]);
                // recurrent part: dh += d_a(prev BPTT step) . W_hh  (my gate block, then quad all-reduce)
                if (s > 0) {
                    float a0 = 0.f, a1 = 0.f;
                    const float* dq = &dabuf[cur][kq][0];
#pragma unroll
                    for (int k4 = 0; k4 < HP; k4 += 4) {
                        const f4 dv = *reinterpret_cast<const f4*>(dq + k4);
                        a0 = fmaf(wT[k4 + 0], dv.x, a0);
                        a1 = fmaf(wT[k4 + 1], dv.y, a1);
                        a0 = fmaf(wT[k4 + 2], dv.z, a0);
                        a1 = fmaf(wT[k4 + 3], dv.w, a1);
                    }
                    float a = a0 + a1;
                    a += quad_xor1(a);
                    a += quad_xor2(a);
                    dh = a;
                }
                const float gi = g4.x, gf = g4.y, gg = g4.z, go = g4.w;
                const float tc = tanhf_(c_t);
                const float dh_t = dh + dyv;
                const float d_o = dh_t * tc;
                const float dc_t = fmaf(dh_t * go, 1.0f - tc * tc, dc);
                float da;
                if (kq == 0)
                    da = dc_t * gg * gi * (1.0f - gi);
                else if (kq == 1)
                    da = dc_t * c_prev * gf * (1.0f - gf);
                else if (kq == 2)
                    da = dc_t * gi * (1.0f - gg * gg);
                else
                    da = d_o * go * (1.0f - go);
                dc = dc_t * gf;
                c_t = c_prev;
                if (live) {
                    dabuf[cur ^ 1][kq][u] = da;
                    dap[(row0 + t) * 8 * H] = da;
                    db_acc += da;
                }
                cur ^= 1;
                __syncthreads();
            }
        }
    }
    if (live) atomicAdd(&P.d_b[(size_t)dir * 4 * H + kq * H + u], db_acc);
    // dead steps contribute nothing: zero their d_a rows for the weight-gradient GEMMs
    for (int i = tid; i < (T - len) * 4 * H; i += blockDim.x) {
        const int t = len + i / (4 * H), k = i % (4 * H);
        P.d_a[(row0 + t) * 8 * H + (size_t)dir * 4 * H + k] = 0.f;
    }
}

template <typename ArgsT, typename K>
static int launch_rec(K kernel, const ArgsT& a, int total_wgs, int H, hipStream_t stream, int kid) {
    const int threads = ((4 * H + 63) / 64) * 64;
    ProfScope ps_(kid, stream);
    hipLaunchKernelGGL(kernel, dim3(total_wgs), dim3(threads), 0, stream, a);
    MMB_HIP(hipGetLastError());
    return MMB_OK;
}

static int kq_for(int H) {
    if (H <= 32) return 8;
    if (H <= 64) return 16;
    if (H <= 100) return 25;
    return 32;
}

}  // namespace mmb

using namespace mmb;

extern "C" int mmb_bilstm_layer_fwd(const mmb_lstm_fwd_desc* d, int n, int device, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    MMB_REQUIRE(d && n >= 1 && n <= MMB_MAX_GROUP, "mmb_bilstm_layer_fwd: n=%d out of range", n);
    MMB_HIP(hipSetDevice(device));
    RecFwdArgs ra{};
    ra.n = n;
    int wg = 0;
    const int H = d[0].H;
    for (int i = 0; i < n; ++i) {
        const mmb_lstm_fwd_desc& p = d[i];
        MMB_REQUIRE(p.H == H, "grouped LSTM problems must share H (%d vs %d)", p.H, H);
        MMB_REQUIRE(p.H >= 1 && p.H <= MMB_LSTM_MAX_H, "H=%d unsupported (max %d)", p.H, MMB_LSTM_MAX_H);
        MMB_REQUIRE(p.B >= 1 && p.T >= 1 && p.I >= 1, "bad LSTM sizes B=%d T=%d I=%d", p.B, p.T, p.I);
        MMB_REQUIRE(p.x && p.lengths && p.y && p.h_n && p.c_n && p.gx && p.gates && p.cs, "null pointer in desc %d", i);
        // input projection, both directions: Gx[:, dir] = x . W_ih[dir]^T + b_ih[dir] + b_hh[dir]
        for (int dir = 0; dir < 2; ++dir) {
            MMB_REQUIRE(p.w_ih[dir] && p.w_hh[dir] && p.b_ih[dir] && p.b_hh[dir], "null weight in desc %d", i);
            GemmArgs g{};
            g.A = p.x; g.B = p.w_ih[dir]; g.C = p.gx + (size_t)dir * 4 * H;
            g.bias = p.b_ih[dir]; g.bias2 = p.b_hh[dir];
            g.M = p.B * p.T; g.N = 4 * H; g.K = p.I;
            g.lda = p.I; g.ldb = p.I; g.ldc = 8 * H;
            g.ta = 0; g.tb = 1; g.accumulate = 0; g.gate_H = H; g.shiftB = 0; g.periodB = 1;
            const int rc = gemm_launch(g, stream);
            if (rc) return rc;
        }
        RecFwdProb& q = ra.p[i];
        q.gx = p.gx; q.w_hh[0] = p.w_hh[0]; q.w_hh[1] = p.w_hh[1]; q.len = p.lengths;
        q.y = p.y; q.gates = p.gates; q.cs = p.cs; q.h_n = p.h_n; q.c_n = p.c_n;
        q.B = p.B; q.T = p.T; q.H = p.H; q.wg_begin = wg;
        wg += 2 * p.B;
    }
    switch (kq_for(H)) {
        case 8: return launch_rec(lstm_rec_fwd_kernel<8>, ra, wg, H, stream, MMB_K_LSTM_REC_FWD);
        case 16: return launch_rec(lstm_rec_fwd_kernel<16>, ra, wg, H, stream, MMB_K_LSTM_REC_FWD);
        case 25: return launch_rec(lstm_rec_fwd_kernel<25>, ra, wg, H, stream, MMB_K_LSTM_REC_FWD);
        default: return launch_rec(lstm_rec_fwd_kernel<32>, ra, wg, H, stream, MMB_K_LSTM_REC_FWD);
    }
}

extern "C" int mmb_bilstm_layer_bwd(const mmb_lstm_bwd_desc* d, int n, int device, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    MMB_REQUIRE(d && n >= 1 && n <= MMB_MAX_GROUP, "mmb_bilstm_layer_bwd: n=%d out of range", n);
    MMB_HIP(hipSetDevice(device));
    RecBwdArgs ra{};
    ra.n = n;
    int wg = 0;
    const int H = d[0].H;
    for (int i = 0; i < n; ++i) {
        const mmb_lstm_bwd_desc& p = d[i];
        MMB_REQUIRE(p.H == H, "grouped LSTM problems must share H (%d vs %d)", p.H, H);
        MMB_REQUIRE(p.H >= 1 && p.H <= MMB_LSTM_MAX_H, "H=%d unsupported (max %d)", p.H, MMB_LSTM_MAX_H);
        MMB_REQUIRE(p.d_y && p.x && p.y && p.lengths && p.gates && p.cs && p.d_w_ih && p.d_w_hh && p.d_b && p.d_a,
                    "null pointer in bwd desc %d", i);
        MMB_HIP(hipMemsetAsync(p.d_b, 0, sizeof(float) * 8 * H, stream));
        RecBwdProb& q = ra.p[i];
        q.d_y = p.d_y; q.d_hn = p.d_hn; q.gates = p.gates; q.cs = p.cs;
        q.w_hh[0] = p.w_hh[0]; q.w_hh[1] = p.w_hh[1]; q.len = p.lengths;
        q.d_a = p.d_a; q.d_b = p.d_b; q.B = p.B; q.T = p.T; q.H = p.H; q.wg_begin = wg;
        wg += 2 * p.B;
    }
    int rc;
    switch (kq_for(H)) {
        case 8: rc = launch_rec(lstm_rec_bwd_kernel<8>, ra, wg, H, stream, MMB_K_LSTM_REC_BWD); break;
        case 16: rc = launch_rec(lstm_rec_bwd_kernel<16>, ra, wg, H, stream, MMB_K_LSTM_REC_BWD); break;
        case 25: rc = launch_rec(lstm_rec_bwd_kernel<25>, ra, wg, H, stream, MMB_K_LSTM_REC_BWD); break;
        default: rc = launch_rec(lstm_rec_bwd_kernel<32>, ra, wg, H, stream, MMB_K_LSTM_REC_BWD); break;
    }
    if (rc) return rc;
    for (int i = 0; i < n; ++i) {
        const mmb_lstm_bwd_desc& p = d[i];
        const int BT = p.B * p.T;
        // d_w_ih (2,4H,I) = d_a^T (8H x BT) . x (BT x I)
        {
            GemmArgs g{};
            g.A = p.d_a; g.B = p.x; g.C = p.d_w_ih;
            g.M = 8 * H; g.N = p.I; g.K = BT; g.lda = 8 * H; g.ldb = p.I; g.ldc = p.I;
            g.ta = 1; g.tb = 0; g.periodB = 1;
            rc = gemm_launch(g, stream);
            if (rc) return rc;
        }
        // d_w_hh[dir] (4H,H) = d_a[dir]^T . h_prev[dir];  h_prev = y shifted by one step inside each sample
        for (int dir = 0; dir < 2; ++dir) {
            GemmArgs g{};
            g.A = p.d_a + (size_t)dir * 4 * H; g.B = p.y + (size_t)dir * H; g.C = p.d_w_hh + (size_t)dir * 4 * H * H;
            g.M = 4 * H; g.N = H; g.K = BT; g.lda = 8 * H; g.ldb = 2 * H; g.ldc = H;
            g.ta = 1; g.tb = 0; g.shiftB = dir ? +1 : -1; g.periodB = p.T;
            rc = gemm_launch(g, stream);
            if (rc) return rc;
        }
        // d_x (BT,I) = sum_dir d_a[dir] (BT x 4H) . W_ih[dir] (4H x I)
        if (p.d_x) {
            for (int dir = 0; dir < 2; ++dir) {
                GemmArgs g{};
                g.A = p.d_a + (size_t)dir * 4 * H; g.B = p.w_ih[dir]; g.C = p.d_x;
                g.M = BT; g.N = p.I; g.K = 4 * H; g.lda = 8 * H; g.ldb = p.I; g.ldc = p.I;
                g.ta = 0; g.tb = 0; g.accumulate = dir; g.periodB = 1;
                rc = gemm_launch(g, stream);
                if (rc) return rc;
            }
        }
    }
    return MMB_OK;
}
