// deepipr_hip.hip -- hand-written CDNA4 (gfx950, MI355X) kernels of the DeepIPR passport layer and
// the C ABI declared in include/deepipr_hip.h.  No torch types, no allocation, no synchronisation:
// every entry point only enqueues kernels on the caller's HIP stream.
//
// Hot-path map (reference kamwoh/DeepIPR, paths relative to /root/reference):
//   passport conv -> pool -> gamma,beta   models/layers/passportconv2d.py:142-175   k_pooled_patch_mean, k_gamma_beta
//   gamma*xhat + beta, ReLU               models/layers/passportconv2d.py:220-222   k_affine_fwd*
//   backward of both                      (stock autograd in the reference)         k_affine_bwd*, k_passport_bwd_finish
//   hinge sign loss on gamma              models/losses/sign_loss.py:18-54          sign_loss_block / k_sign_loss_*
//
// Design notes (DESIGN.md has the roofline arithmetic):
//   * Everything here is HBM/L2- or latency-bound byte work on fp32 NCHW tensors.  The passport conv
//     followed by the global mean is linear in the key, so gamma = W_mat . pooled_im2col(key): a
//     GEMV that streams W exactly once.  Reshaping it into a GEMM to reach MFMA would multiply the
//     flops by L = Ho*Wo without removing a single byte of W traffic, so no MFMA is used.
//   * 64-wide wavefronts: cross-lane reductions are __shfl_xor trees over 64 lanes, per-workgroup
//     combines go through LDS, cross-workgroup combines are fixed-order partial sums finished in the
//     prologue of the NEXT kernel (no float atomics, no in-launch grid sync): bit-reproducible.
//   * 16 B per lane (float4) global accesses wherever the plane size allows; grids are sized to
//     >= 4 workgroups per CU (256 CUs) and capped at 2048 with grid-stride loops.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <mutex>
#include <vector>

#include "../../include/deepipr_hip.h"

namespace {

constexpr int kThreads = 256;           // 4 wavefronts of 64
constexpr int kWave = 64;
constexpr int kMaxGrid = 2048;          // 8 workgroups per CU on 256 CUs
constexpr int kDkeySplit = 16;
constexpr int kRowPairMinCo = 256;     // W rows are processed two per workgroup once Co >= 512

thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(DEEPIPR_ELAUNCH, "%s: %s", what, hipGetErrorString(e));
    return DEEPIPR_OK;
}

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// ---------------------------------------------------------------------------------------------
// Opt-in in-situ kernel timing (deepipr_profile_*): when enabled, every launch is bracketed by two
// hipEvents recorded on the launch stream; durations are read back later with deepipr_profile_read.
// Off by default (one relaxed bool load per launch); must stay off during hipGraph capture.
struct ProfState {
    std::mutex mu;
    bool on = false;
    unsigned long long scopes = 0;
    std::vector<hipEvent_t> pool;
    struct Pending { int k; hipEvent_t a, b; };
    std::vector<Pending> pending;
    double total_ms[DEEPIPR_PROFILE_KERNELS] = {};
    long long launches[DEEPIPR_PROFILE_KERNELS] = {};
};
ProfState g_prof;

struct ProfScope {
    int k;
    hipStream_t st;
    hipEvent_t a = nullptr, b = nullptr;
    ProfScope(int kernel, hipStream_t stream) : k(kernel), st(stream) {
        if (!g_prof.on) return;
        std::lock_guard<std::mutex> lk(g_prof.mu);
        auto take = [&]() {
            hipEvent_t e;
            if (!g_prof.pool.empty()) { e = g_prof.pool.back(); g_prof.pool.pop_back(); }
            else if (hipEventCreate(&e) != hipSuccess) e = nullptr;
            return e;
        };
        a = take();
        b = take();
        if (a && b) (void)hipEventRecord(a, st);
    }
    ~ProfScope() {
        if (!a || !b) return;
        (void)hipEventRecord(b, st);
        std::lock_guard<std::mutex> lk(g_prof.mu);
        g_prof.pending.push_back({k, a, b});
        // every 4th bracket is followed by an EMPTY one on the same stream: its elapsed time is the
        // event-pair overhead at this point of the run, which callers subtract (DEEPIPR_K_NULL_BRACKET)
        if ((++g_prof.scopes & 3) == 0) {
            hipEvent_t n0 = nullptr, n1 = nullptr;
            auto take = [&]() {
                hipEvent_t e;
                if (!g_prof.pool.empty()) { e = g_prof.pool.back(); g_prof.pool.pop_back(); }
                else if (hipEventCreate(&e) != hipSuccess) e = nullptr;
                return e;
            };
            n0 = take();
            n1 = take();
            if (n0 && n1) {
                (void)hipEventRecord(n0, st);
                (void)hipEventRecord(n1, st);
                g_prof.pending.push_back({DEEPIPR_K_NULL_BRACKET, n0, n1});
            }
        }
    }
};

// ---------------------------------------------------------------------------------------------
// Exact unsigned division by a launch-time constant (n < 2^31): q = (n * M) >> S.
struct FastDiv {
    unsigned long long M;
    unsigned S;
    unsigned d;
};

FastDiv make_fastdiv(unsigned d) {
    FastDiv f;
    f.d = d;
    unsigned l = 0;
    while ((1ull << l) < d) ++l;
    f.S = 32 + l;
    f.M = ((1ull << f.S) + d - 1) / d;
    return f;
}

__device__ __forceinline__ unsigned fdiv(unsigned n, const FastDiv &f) {
    return static_cast<unsigned>((static_cast<unsigned long long>(n) * f.M) >> f.S);
}

// ---------------------------------------------------------------------------------------------
// 64-lane butterfly sums (fixed order -> deterministic).
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, kWave);
    return v;
}

// Sum over the whole 256-thread workgroup; result valid in every thread.  `red` holds >= 4 doubles.
__device__ __forceinline__ double block_sum(double v, double *red) {
    v = wave_sum(v);
    const int wave = threadIdx.x >> 6;
    __syncthreads();                      // protect `red` from the previous use
    if ((threadIdx.x & 63) == 0) red[wave] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// y = relu?(g*x + b) with the reference's two roundings (aten::mul then aten::add).
template <bool RELU>
__device__ __forceinline__ float affine1(float x, float g, float b) {
    float y = __fadd_rn(__fmul_rn(g, x), b);
    return RELU ? fmaxf(y, 0.0f) : y;
}

// ============================================================================================
// Pooled passport patches:  m[key][k] = mean_{b,oh,ow} key[b, ci, oh*st + r - pad, ow*st + q - pad]
// One wavefront per k; lanes stride over the B*Ho*Wo patch positions; f64 accumulation.
// ============================================================================================
__global__ __launch_bounds__(kThreads) void k_pooled_patch_mean(
    const float *__restrict__ keys, int B, int Ci, int H, int W, int kh, int kw, int stride, int pad,
    int Ho, int Wo, double *__restrict__ m_out) {
    const int K = Ci * kh * kw;
    const int k = blockIdx.x * (kThreads / kWave) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (k >= K) return;                               // whole wave exits together
    const float *key = keys + static_cast<size_t>(blockIdx.y) * B * Ci * H * W;
    const int ci = k / (kh * kw);
    const int r = (k / kw) % kh;
    const int q = k % kw;
    const int L = Ho * Wo;
    double acc = 0.0;
    for (int i = lane; i < B * L; i += kWave) {
        const int b = i / L;
        const int l = i - b * L;
        const int oh = l / Wo, ow = l - oh * Wo;
        const int ih = oh * stride + r - pad, iw = ow * stride + q - pad;
        if (ih >= 0 && ih < H && iw >= 0 && iw < W)
            acc += static_cast<double>(key[((static_cast<size_t>(b) * Ci + ci) * H + ih) * W + iw]);
    }
    acc = wave_sum(acc);
    if (lane == 0) m_out[static_cast<size_t>(blockIdx.y) * K + k] = acc / static_cast<double>(B * L);
}

// ============================================================================================
// gamma/beta GEMV: RPW output-channel rows of W[Co][K] per workgroup; W is streamed once with
// 16 B/lane loads, the two pooled f64 vectors come from L2 and are reused for the RPW rows (they are
// 4x the bytes of a row, so RPW=2 halves the L2->CU traffic); f64 FMA accumulation.
// ============================================================================================
template <bool VEC, int RPW>
__global__ __launch_bounds__(kThreads) void k_gamma_beta(
    const float *__restrict__ W, const double *__restrict__ s, int Co, int K,
    float *__restrict__ gamma, float *__restrict__ beta) {
    __shared__ double red[8 * RPW];
    const int co0 = blockIdx.x * RPW;
    const double *ss = s, *sb = s + K;
    double as[RPW], ab[RPW];
    const float *row[RPW];
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        as[r] = 0.0;
        ab[r] = 0.0;
        row[r] = W + static_cast<size_t>(min(co0 + r, Co - 1)) * K;     // clamp: tail rows recompute the last row
    }
    if (VEC) {
        const double2 *ss2 = reinterpret_cast<const double2 *>(ss);
        const double2 *sb2 = reinterpret_cast<const double2 *>(sb);
        for (int q = threadIdx.x; q < K / 4; q += kThreads) {
            float4 w[RPW];
#pragma unroll
            for (int r = 0; r < RPW; ++r) w[r] = reinterpret_cast<const float4 *>(row[r])[q];
            const double2 s0 = ss2[2 * q], s1 = ss2[2 * q + 1];
            const double2 b0 = sb2[2 * q], b1 = sb2[2 * q + 1];
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                as[r] = fma(static_cast<double>(w[r].x), s0.x, as[r]);
                as[r] = fma(static_cast<double>(w[r].y), s0.y, as[r]);
                as[r] = fma(static_cast<double>(w[r].z), s1.x, as[r]);
                as[r] = fma(static_cast<double>(w[r].w), s1.y, as[r]);
                ab[r] = fma(static_cast<double>(w[r].x), b0.x, ab[r]);
                ab[r] = fma(static_cast<double>(w[r].y), b0.y, ab[r]);
                ab[r] = fma(static_cast<double>(w[r].z), b1.x, ab[r]);
                ab[r] = fma(static_cast<double>(w[r].w), b1.y, ab[r]);
            }
        }
    } else {
        for (int k = threadIdx.x; k < K; k += kThreads) {
            const double vs = ss[k], vb = sb[k];
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                const double w = static_cast<double>(row[r][k]);
                as[r] = fma(w, vs, as[r]);
                ab[r] = fma(w, vb, ab[r]);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        as[r] = block_sum(as[r], red + 8 * r);
        ab[r] = block_sum(ab[r], red + 8 * r + 4);
        if (threadIdx.x == 0 && co0 + r < Co) {
            gamma[co0 + r] = static_cast<float>(as[r]);
            beta[co0 + r] = static_cast<float>(ab[r]);
        }
    }
}

// Rank-2 update rows: dW[co0+r, :] = dg[r] * m_scale + db[r] * m_bias (pooled means rounded to f32),
// the pooled vectors loaded once for the RPW rows.
template <bool VEC, int RPW>
__device__ __forceinline__ void write_dw_rows(float *__restrict__ dW, const double *__restrict__ s, int Co,
                                              int K, int co0, const float *dg, const float *db) {
    const double *ss = s, *sb = s + K;
    if (VEC) {
        const double2 *ss2 = reinterpret_cast<const double2 *>(ss);
        const double2 *sb2 = reinterpret_cast<const double2 *>(sb);
        for (int q = threadIdx.x; q < K / 4; q += kThreads) {
            const double2 s0 = ss2[2 * q], s1 = ss2[2 * q + 1];
            const double2 b0 = sb2[2 * q], b1 = sb2[2 * q + 1];
            const float4 ms = make_float4(static_cast<float>(s0.x), static_cast<float>(s0.y),
                                          static_cast<float>(s1.x), static_cast<float>(s1.y));
            const float4 mb = make_float4(static_cast<float>(b0.x), static_cast<float>(b0.y),
                                          static_cast<float>(b1.x), static_cast<float>(b1.y));
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                if (co0 + r >= Co) break;
                float4 o;
                o.x = fmaf(dg[r], ms.x, db[r] * mb.x);
                o.y = fmaf(dg[r], ms.y, db[r] * mb.y);
                o.z = fmaf(dg[r], ms.z, db[r] * mb.z);
                o.w = fmaf(dg[r], ms.w, db[r] * mb.w);
                reinterpret_cast<float4 *>(dW + static_cast<size_t>(co0 + r) * K)[q] = o;
            }
        }
    } else {
        for (int k = threadIdx.x; k < K; k += kThreads) {
            const float ms = static_cast<float>(ss[k]), mb = static_cast<float>(sb[k]);
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                if (co0 + r >= Co) break;
                dW[static_cast<size_t>(co0 + r) * K + k] = fmaf(dg[r], ms, db[r] * mb);
            }
        }
    }
}

template <bool VEC, int RPW>
__global__ __launch_bounds__(kThreads) void k_gamma_beta_bwd(
    const float *__restrict__ dgamma, const float *__restrict__ dbeta, const double *__restrict__ s,
    int Co, int K, float *__restrict__ dW) {
    const int co0 = blockIdx.x * RPW;
    float dg[RPW], db[RPW];
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const int co = min(co0 + r, Co - 1);
        dg[r] = dgamma[co];
        db[r] = dbeta[co];
    }
    write_dw_rows<VEC, RPW>(dW, s, Co, K, co0, dg, db);
}

// ============================================================================================
// d/dkey:  u[j][k] = sum_co d[j][co] * W[co,k]  (split over co, f64 partials), then gathered back
// onto the key's pixels.
// ============================================================================================
__global__ __launch_bounds__(kThreads) void k_dkey_colsum(
    const float *__restrict__ dgamma, const float *__restrict__ dbeta, const float *__restrict__ W,
    int Co, int K, double *__restrict__ part /* [split][2][K] */) {
    const int k = blockIdx.x * kThreads + threadIdx.x;
    if (k >= K) return;
    const int per = (Co + kDkeySplit - 1) / kDkeySplit;
    const int c0 = blockIdx.y * per, c1 = min(Co, c0 + per);
    double ag = 0.0, ab = 0.0;
    for (int co = c0; co < c1; ++co) {
        const double w = static_cast<double>(W[static_cast<size_t>(co) * K + k]);
        ag = fma(static_cast<double>(dgamma[co]), w, ag);
        ab = fma(static_cast<double>(dbeta[co]), w, ab);
    }
    part[(static_cast<size_t>(blockIdx.y) * 2 + 0) * K + k] = ag;
    part[(static_cast<size_t>(blockIdx.y) * 2 + 1) * K + k] = ab;
}

__global__ __launch_bounds__(kThreads) void k_dkey_gather(
    const double *__restrict__ part, int K, int B, int Ci, int H, int W, int kh, int kw, int stride,
    int pad, int Ho, int Wo, double inv_n, float *__restrict__ dkeys /* [2][B][Ci][H][W] */) {
    const int per_key = B * Ci * H * W;
    const int idx = blockIdx.x * kThreads + threadIdx.x;
    if (idx >= 2 * per_key) return;
    const int j = idx / per_key;
    int rem = idx - j * per_key;
    const int iw = rem % W;
    rem /= W;
    const int ih = rem % H;
    rem /= H;
    const int ci = rem % Ci;
    double acc = 0.0;
    for (int r = 0; r < kh; ++r) {
        const int th = ih + pad - r;
        if (th < 0 || th % stride != 0 || th / stride >= Ho) continue;
        for (int q = 0; q < kw; ++q) {
            const int tw = iw + pad - q;
            if (tw < 0 || tw % stride != 0 || tw / stride >= Wo) continue;
            const int k = (ci * kh + r) * kw + q;
            double u = 0.0;
            for (int sp = 0; sp < kDkeySplit; ++sp) u += part[(static_cast<size_t>(sp) * 2 + j) * K + k];
            acc += u;
        }
    }
    dkeys[idx] = static_cast<float>(acc * inv_n);
}

// ============================================================================================
// Hinge sign loss on gamma, computed by ONE workgroup (C <= a few thousand floats).
// ============================================================================================
__device__ __forceinline__ void sign_loss_block(const float *__restrict__ gamma,
                                                const float *__restrict__ b, float alpha, float margin,
                                                float l2, int C, float *__restrict__ loss,
                                                float *__restrict__ acc, int8_t *__restrict__ bits,
                                                double *red /* >= 12 doubles of LDS */) {
    double hinge = 0.0, sq = 0.0, match = 0.0;
    for (int c = threadIdx.x; c < C; c += kThreads) {
        const float g = gamma[c], bb = b[c];
        // alpha * relu(-b*g + margin), same operation order as models/losses/sign_loss.py:27
        const float z = __fadd_rn(__fmul_rn(-bb, g), margin);
        hinge += static_cast<double>(__fmul_rn(alpha, fmaxf(z, 0.0f)));
        sq += static_cast<double>(__fmul_rn(g, g));
        const int sg = (g > 0.0f) - (g < 0.0f);
        const int sb = (bb > 0.0f) - (bb < 0.0f);
        match += (sg == sb) ? 1.0 : 0.0;
        if (bits) bits[c] = static_cast<int8_t>(sg);
    }
    hinge = block_sum(hinge, red);
    sq = block_sum(sq, red + 4);
    match = block_sum(match, red + 8);
    if (threadIdx.x == 0) {
        if (loss) *loss = static_cast<float>(hinge + static_cast<double>(l2) * sq);
        if (acc) *acc = static_cast<float>(match / static_cast<double>(C));
    }
}

__device__ __forceinline__ float sign_loss_grad1(float g, float bb, float alpha, float margin, float l2) {
    const float z = __fadd_rn(__fmul_rn(-bb, g), margin);
    const float h = (z > 0.0f) ? -alpha * bb : 0.0f;
    return h + 2.0f * l2 * g;
}

__global__ __launch_bounds__(kThreads) void k_sign_loss_fwd(
    const float *__restrict__ gamma, const float *__restrict__ b, float alpha, float margin, float l2,
    int C, float *__restrict__ loss, float *__restrict__ acc, int8_t *__restrict__ bits) {
    __shared__ double red[12];
    sign_loss_block(gamma, b, alpha, margin, l2, C, loss, acc, bits, red);
}

__global__ __launch_bounds__(kThreads) void k_sign_loss_bwd(
    const float *__restrict__ dloss, const float *__restrict__ gamma, const float *__restrict__ b,
    float alpha, float margin, float l2, int C, float *__restrict__ dgamma) {
    const int c = blockIdx.x * kThreads + threadIdx.x;
    if (c < C) dgamma[c] = dloss[0] * sign_loss_grad1(gamma[c], b[c], alpha, margin, l2);
}

// ============================================================================================
// Passport affine forward.  Flat grid-stride over float4s; plane index by exact fast division.
// An optional extra workgroup (blockIdx.x == gridDim.x-1 when `with_sign`) computes the sign loss
// so that a passport layer's forward after gamma/beta is a single launch.
// ============================================================================================
struct SignArgs {
    const float *b;
    float alpha, margin, l2;
    float *loss, *acc;
    int8_t *bits;
};

template <bool RELU>
__global__ __launch_bounds__(kThreads) void k_affine_fwd_v4(
    const float4 *__restrict__ x, const float *__restrict__ gamma, const float *__restrict__ beta,
    float4 *__restrict__ y, unsigned n4, FastDiv p4div, FastDiv cdiv, unsigned C, int with_sign,
    SignArgs sa) {
    __shared__ double red[12];
    unsigned nblk = gridDim.x;
    if (with_sign) {
        nblk -= 1;
        if (blockIdx.x == nblk) {         // the extra workgroup: sign loss only
            sign_loss_block(gamma, sa.b, sa.alpha, sa.margin, sa.l2, static_cast<int>(C), sa.loss,
                            sa.acc, sa.bits, red);
            return;
        }
    }
    const unsigned step = nblk * kThreads;
    unsigned q = blockIdx.x * kThreads + threadIdx.x;
    // two independent float4s per trip: both loads are in flight before either is consumed
    for (; q + step < n4; q += 2 * step) {
        const unsigned q1 = q + step;
        const float4 v0 = x[q], v1 = x[q1];
        const unsigned p0 = fdiv(q, p4div), p1 = fdiv(q1, p4div);
        const unsigned c0 = p0 - fdiv(p0, cdiv) * C, c1 = p1 - fdiv(p1, cdiv) * C;
        const float g0 = gamma[c0], t0 = beta[c0], g1 = gamma[c1], t1 = beta[c1];
        y[q] = make_float4(affine1<RELU>(v0.x, g0, t0), affine1<RELU>(v0.y, g0, t0),
                           affine1<RELU>(v0.z, g0, t0), affine1<RELU>(v0.w, g0, t0));
        y[q1] = make_float4(affine1<RELU>(v1.x, g1, t1), affine1<RELU>(v1.y, g1, t1),
                            affine1<RELU>(v1.z, g1, t1), affine1<RELU>(v1.w, g1, t1));
    }
    if (q < n4) {
        const unsigned plane = fdiv(q, p4div);
        const unsigned c = plane - fdiv(plane, cdiv) * C;
        const float g = gamma[c], bt = beta[c];
        const float4 v = x[q];
        y[q] = make_float4(affine1<RELU>(v.x, g, bt), affine1<RELU>(v.y, g, bt), affine1<RELU>(v.z, g, bt),
                           affine1<RELU>(v.w, g, bt));
    }
}

// Planes whose size is not a multiple of 4 floats (7x7 maps): one element per lane.
template <bool RELU>
__global__ __launch_bounds__(kThreads) void k_affine_fwd_s(
    const float *__restrict__ x, const float *__restrict__ gamma, const float *__restrict__ beta,
    float *__restrict__ y, unsigned n, FastDiv pdiv, FastDiv cdiv, unsigned C, int with_sign,
    SignArgs sa) {
    __shared__ double red[12];
    unsigned nblk = gridDim.x;
    if (with_sign) {
        nblk -= 1;
        if (blockIdx.x == nblk) {
            sign_loss_block(gamma, sa.b, sa.alpha, sa.margin, sa.l2, static_cast<int>(C), sa.loss,
                            sa.acc, sa.bits, red);
            return;
        }
    }
    const unsigned step = nblk * kThreads;
    for (unsigned i = blockIdx.x * kThreads + threadIdx.x; i < n; i += step) {
        const unsigned plane = fdiv(i, pdiv);
        const unsigned c = plane - fdiv(plane, cdiv) * C;
        y[i] = affine1<RELU>(x[i], gamma[c], beta[c]);
    }
}

// ============================================================================================
// Passport affine backward, one pass over dy and xhat.
//
// The tensor is [N][C][P].  A workgroup owns a tile of CT consecutive channels, i.e. for every image
// n one contiguous "row" of CT*P floats, and a slice of the batch (blockIdx.y = split).  Thread t owns
// one unit (VEC floats) at a fixed offset of the row, so its channel(s) never change while it walks
// over the images: per-thread register accumulation, then a fixed-order LDS + wave-shuffle combine
// per channel.  Partial sums per (split, channel) go to the f64 workspace and are finished by the
// consumer kernel (k_reduce_partials or k_passport_bwd_finish).
// ============================================================================================
struct BwdPlan {
    int VEC;        // 4 or 1 floats per unit
    int large;      // plane has more than 256 units: one channel per workgroup, loop over the plane
    int CT;         // channels per tile
    int row_u;      // units per row = CT*P/VEC (small) or P/VEC (large)
    int npi;        // images processed per iteration (small only)
    int tiles;      // channel tiles
    int iters;      // iterations over the batch in total
    int ips;        // iterations per split
    int NS;         // number of batch splits
};

BwdPlan plan_bwd(int N, int C, int P, bool can_vec) {
    BwdPlan p;
    p.VEC = (can_vec && P % 4 == 0) ? 4 : 1;
    const int pu = P / p.VEC;
    p.large = pu > kThreads;
    if (!p.large) {
        int ct = (64 + P - 1) / P;                      // rows of >= 256 B
        if (ct < 1) ct = 1;
        if (ct > C) ct = C;
        while (ct > 1 && ct * pu > kThreads) --ct;
        p.CT = ct;
        p.row_u = ct * pu;
        p.npi = kThreads / p.row_u;
        p.tiles = (C + ct - 1) / ct;
        p.iters = (N + p.npi - 1) / p.npi;
    } else {
        p.CT = 1;
        p.row_u = pu;
        p.npi = 1;
        p.tiles = C;
        p.iters = N;
    }
    int ns = (kMaxGrid / 2 + p.tiles - 1) / p.tiles;     // aim at >= 1024 workgroups
    if (ns < 1) ns = 1;
    if (ns > p.iters) ns = p.iters;
    p.ips = (p.iters + ns - 1) / ns;
    p.NS = (p.iters + p.ips - 1) / p.ips;
    return p;
}

template <int VEC>
struct Unit;
template <>
struct Unit<4> {
    using T = float4;
    static __device__ __forceinline__ void get(const T &v, float *a) { a[0] = v.x; a[1] = v.y; a[2] = v.z; a[3] = v.w; }
    static __device__ __forceinline__ T make(const float *a) { return make_float4(a[0], a[1], a[2], a[3]); }
};
template <>
struct Unit<1> {
    using T = float;
    static __device__ __forceinline__ void get(const T &v, float *a) { a[0] = v; }
    static __device__ __forceinline__ T make(const float *a) { return a[0]; }
};

template <int VEC, bool RELU>
__global__ __launch_bounds__(kThreads) void k_affine_bwd_small(
    const float *__restrict__ dy, const float *__restrict__ xh, const float *__restrict__ gamma,
    const float *__restrict__ beta, float *__restrict__ dx, double *__restrict__ part /* [NS][2][C] */,
    int N, int C, int P, BwdPlan pl) {
    using U = typename Unit<VEC>::T;
    __shared__ float sacc[2][kThreads * 4];
    const int c0 = blockIdx.x * pl.CT;
    const int ct = min(pl.CT, C - c0);
    const int t = threadIdx.x;
    const int r = t / pl.row_u;                  // which image of the iteration
    const int u = t - r * pl.row_u;              // unit inside the row
    const bool lane_on = (r < pl.npi) && (u * VEC < ct * P);

    // element i of this thread's unit: channel cc[i] of the tile, position e[i] in its plane; the LDS
    // slot (cc*npi + r)*P + e keeps every channel's partial sums contiguous for the combine below
    float g[VEC], bt[VEC], a_gx[VEC], a_g[VEC];
    int slot[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
        a_gx[i] = 0.0f;
        a_g[i] = 0.0f;
        const int er = u * VEC + i;
        const int cc = lane_on ? er / P : 0;
        slot[i] = (cc * pl.npi + r) * P + (er - cc * P);
        g[i] = gamma[c0 + cc];
        bt[i] = beta[c0 + cc];
    }
    if (lane_on) {
        const int it0 = blockIdx.y * pl.ips, it1 = min(pl.iters, it0 + pl.ips);
        const size_t img = static_cast<size_t>(C) * P;
        size_t off = (static_cast<size_t>(it0 * pl.npi + r) * C + c0) * P + static_cast<size_t>(u) * VEC;
        const size_t hop = img * pl.npi;
        int n = it0 * pl.npi + r;
        U vdy{}, vxh{};
        if (n < N) {
            vdy = *reinterpret_cast<const U *>(dy + off);
            vxh = *reinterpret_cast<const U *>(xh + off);
        }
        for (int it = it0; it < it1 && n < N; ++it, n += pl.npi, off += hop) {
            U ndy{}, nxh{};
            if (it + 1 < it1 && n + pl.npi < N) {            // next image's loads fly during this one's math
                ndy = *reinterpret_cast<const U *>(dy + off + hop);
                nxh = *reinterpret_cast<const U *>(xh + off + hop);
            }
            float d[VEC], x[VEC], o[VEC];
            Unit<VEC>::get(vdy, d);
            Unit<VEC>::get(vxh, x);
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                float dz = d[i];
                if (RELU) dz = (__fadd_rn(__fmul_rn(g[i], x[i]), bt[i]) > 0.0f) ? dz : 0.0f;
                o[i] = dz * g[i];
                a_gx[i] = fmaf(dz, x[i], a_gx[i]);
                a_g[i] += dz;
            }
            *reinterpret_cast<U *>(dx + off) = Unit<VEC>::make(o);
            vdy = ndy;
            vxh = nxh;
        }
    }
    if (lane_on) {        // slots of channels beyond a partial tile are never read
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            sacc[0][slot[i]] = a_gx[i];
            sacc[1][slot[i]] = a_g[i];
        }
    }
    __syncthreads();
    // one wavefront per channel: all 64 lanes stride over its npi*P contiguous slots, f64 butterfly
    const int wave = t >> 6, lane = t & 63;
    const int cnt = pl.npi * P;
    for (int cc = wave; cc < ct; cc += kThreads / kWave) {
        const float *p0 = sacc[0] + cc * cnt, *p1 = sacc[1] + cc * cnt;
        double sgx = 0.0, sg = 0.0;
        for (int e = lane; e < cnt; e += kWave) {
            sgx += static_cast<double>(p0[e]);
            sg += static_cast<double>(p1[e]);
        }
        sgx = wave_sum(sgx);
        sg = wave_sum(sg);
        if (lane == 0) {
            part[(static_cast<size_t>(blockIdx.y) * 2 + 0) * C + c0 + cc] = sgx;
            part[(static_cast<size_t>(blockIdx.y) * 2 + 1) * C + c0 + cc] = sg;
        }
    }
}

// Planes larger than 256 units: one channel per workgroup, every thread strides over the plane.
template <int VEC, bool RELU>
__global__ __launch_bounds__(kThreads) void k_affine_bwd_large(
    const float *__restrict__ dy, const float *__restrict__ xh, const float *__restrict__ gamma,
    const float *__restrict__ beta, float *__restrict__ dx, double *__restrict__ part, int N, int C,
    int P, BwdPlan pl) {
    using U = typename Unit<VEC>::T;
    __shared__ double red[8];
    const int c = blockIdx.x;
    const float g = gamma[c], bt = beta[c];
    double a_gx = 0.0, a_g = 0.0;
    const int n0 = blockIdx.y * pl.ips, n1 = min(N, n0 + pl.ips);
    for (int n = n0; n < n1; ++n) {
        const size_t base = (static_cast<size_t>(n) * C + c) * P;
        float p_gx = 0.0f, p_g = 0.0f;
        for (int u = threadIdx.x; u < pl.row_u; u += kThreads) {
            const size_t off = base + static_cast<size_t>(u) * VEC;
            const U vdy = *reinterpret_cast<const U *>(dy + off);
            const U vxh = *reinterpret_cast<const U *>(xh + off);
            float d[VEC], x[VEC], o[VEC];
            Unit<VEC>::get(vdy, d);
            Unit<VEC>::get(vxh, x);
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                float dz = d[i];
                if (RELU) dz = (__fadd_rn(__fmul_rn(g, x[i]), bt) > 0.0f) ? dz : 0.0f;
                o[i] = dz * g;
                p_gx = fmaf(dz, x[i], p_gx);
                p_g += dz;
            }
            *reinterpret_cast<U *>(dx + off) = Unit<VEC>::make(o);
        }
        a_gx += static_cast<double>(p_gx);
        a_g += static_cast<double>(p_g);
    }
    a_gx = block_sum(a_gx, red);
    a_g = block_sum(a_g, red + 4);
    if (threadIdx.x == 0) {
        part[(static_cast<size_t>(blockIdx.y) * 2 + 0) * C + c] = a_gx;
        part[(static_cast<size_t>(blockIdx.y) * 2 + 1) * C + c] = a_g;
    }
}

// Finish: dgamma[c], dbeta[c] = fixed-order sum of the NS partials.
__global__ __launch_bounds__(kThreads) void k_reduce_partials(
    const double *__restrict__ part, int NS, int C, float *__restrict__ dgamma, float *__restrict__ dbeta) {
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= 2 * C) return;
    const int which = i / C, c = i - which * C;
    double acc = 0.0;
    for (int sp = 0; sp < NS; ++sp) acc += part[(static_cast<size_t>(sp) * 2 + which) * C + c];
    (which == 0 ? dgamma : dbeta)[c] = static_cast<float>(acc);
}

// Fused finish of a passport layer's backward: RPW output channels per workgroup.
//   dgamma[co] = sum_split part + dgamma_extra[co] + dloss * d(sign loss)/dgamma ;  dbeta likewise
//   dW[co, :]  = dgamma[co] * m_scale + dbeta[co] * m_bias
// Every wavefront finishes the (few) partial sums itself -- identical fixed-order arithmetic in all four,
// so no LDS hand-off or barrier delays the dW stores.
template <bool VEC, int RPW>
__global__ __launch_bounds__(kThreads) void k_passport_bwd_finish(
    const double *__restrict__ part, int NS, int C, const float *__restrict__ gamma,
    const float *__restrict__ b, float alpha, float margin, float l2, const float *__restrict__ dloss,
    const float *__restrict__ dgamma_extra, const float *__restrict__ dbeta_extra,
    const double *__restrict__ s, int K, float *__restrict__ dgamma, float *__restrict__ dbeta,
    float *__restrict__ dW) {
    const int co0 = blockIdx.x * RPW;
    const int lane = threadIdx.x & 63;
    float dgv[RPW], dbv[RPW];
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const int co = min(co0 + r, C - 1);
        double ag = 0.0, ab = 0.0;
        for (int sp = lane; sp < NS; sp += kWave) {
            ag += part[(static_cast<size_t>(sp) * 2 + 0) * C + co];
            ab += part[(static_cast<size_t>(sp) * 2 + 1) * C + co];
        }
        ag = wave_sum(ag);
        ab = wave_sum(ab);
        float dg = static_cast<float>(ag), db = static_cast<float>(ab);
        if (dgamma_extra) dg += dgamma_extra[co];
        if (dbeta_extra) db += dbeta_extra[co];
        if (dloss) dg += dloss[0] * sign_loss_grad1(gamma[co], b[co], alpha, margin, l2);
        dgv[r] = dg;
        dbv[r] = db;
        if (threadIdx.x == 0 && co0 + r < C) {
            dgamma[co] = dg;
            dbeta[co] = db;
        }
    }
    write_dw_rows<VEC, RPW>(dW, s, C, K, co0, dgv, dbv);
}

// ---------------------------------------------------------------------------------------------
// host-side launch helpers
// ---------------------------------------------------------------------------------------------
int grid_for(size_t work_items) {
    size_t g = (work_items + kThreads - 1) / kThreads;
    if (g > static_cast<size_t>(kMaxGrid)) g = kMaxGrid;
    if (g < 1) g = 1;
    return static_cast<int>(g);
}

int launch_affine_fwd(const float *xhat, const float *gamma, const float *beta, float *y, int N, int C,
                      int HW, int relu, bool with_sign, const SignArgs &sa, hipStream_t st) {
    const size_t total = static_cast<size_t>(N) * C * HW;
    if (total >= (1ull << 31)) return fail(DEEPIPR_EINVAL, "affine_relu_fwd: tensor has >= 2^31 elements");
    const FastDiv cdiv = make_fastdiv(static_cast<unsigned>(C));
    ProfScope prof(DEEPIPR_K_AFFINE_FWD, st);
    const bool vec = (HW % 4 == 0) && aligned16(xhat) && aligned16(y);
    if (vec) {
        const unsigned n4 = static_cast<unsigned>(total / 4);
        const FastDiv pdiv = make_fastdiv(static_cast<unsigned>(HW / 4));
        const int grid = grid_for(n4) + (with_sign ? 1 : 0);
        if (relu)
            hipLaunchKernelGGL(k_affine_fwd_v4<true>, dim3(grid), dim3(kThreads), 0, st,
                               reinterpret_cast<const float4 *>(xhat), gamma, beta,
                               reinterpret_cast<float4 *>(y), n4, pdiv, cdiv, static_cast<unsigned>(C),
                               with_sign ? 1 : 0, sa);
        else
            hipLaunchKernelGGL(k_affine_fwd_v4<false>, dim3(grid), dim3(kThreads), 0, st,
                               reinterpret_cast<const float4 *>(xhat), gamma, beta,
                               reinterpret_cast<float4 *>(y), n4, pdiv, cdiv, static_cast<unsigned>(C),
                               with_sign ? 1 : 0, sa);
    } else {
        const FastDiv pdiv = make_fastdiv(static_cast<unsigned>(HW));
        const int grid = grid_for(total) + (with_sign ? 1 : 0);
        if (relu)
            hipLaunchKernelGGL(k_affine_fwd_s<true>, dim3(grid), dim3(kThreads), 0, st, xhat, gamma, beta, y,
                               static_cast<unsigned>(total), pdiv, cdiv, static_cast<unsigned>(C),
                               with_sign ? 1 : 0, sa);
        else
            hipLaunchKernelGGL(k_affine_fwd_s<false>, dim3(grid), dim3(kThreads), 0, st, xhat, gamma, beta, y,
                               static_cast<unsigned>(total), pdiv, cdiv, static_cast<unsigned>(C),
                               with_sign ? 1 : 0, sa);
    }
    return check_launch("affine_relu_fwd");
}

template <int VEC, bool RELU>
void launch_bwd_t(const float *dy, const float *xh, const float *g, const float *bt, float *dx,
                  double *part, int N, int C, int P, const BwdPlan &pl, hipStream_t st) {
    const dim3 grid(pl.tiles, pl.NS);
    if (pl.large)
        hipLaunchKernelGGL((k_affine_bwd_large<VEC, RELU>), grid, dim3(kThreads), 0, st, dy, xh, g, bt, dx,
                           part, N, C, P, pl);
    else
        hipLaunchKernelGGL((k_affine_bwd_small<VEC, RELU>), grid, dim3(kThreads), 0, st, dy, xh, g, bt, dx,
                           part, N, C, P, pl);
}

int launch_affine_bwd(const float *dy, const float *xh, const float *g, const float *bt, float *dx,
                      double *part, int N, int C, int P, int relu, BwdPlan *plan_out, hipStream_t st) {
    const bool can_vec = aligned16(dy) && aligned16(xh) && aligned16(dx);
    const BwdPlan pl = plan_bwd(N, C, P, can_vec);
    if (pl.NS > 65535) return fail(DEEPIPR_EINVAL, "affine_relu_bwd: too many batch splits");
    ProfScope prof(DEEPIPR_K_AFFINE_BWD, st);
    if (pl.VEC == 4) {
        if (relu) launch_bwd_t<4, true>(dy, xh, g, bt, dx, part, N, C, P, pl, st);
        else launch_bwd_t<4, false>(dy, xh, g, bt, dx, part, N, C, P, pl, st);
    } else {
        if (relu) launch_bwd_t<1, true>(dy, xh, g, bt, dx, part, N, C, P, pl, st);
        else launch_bwd_t<1, false>(dy, xh, g, bt, dx, part, N, C, P, pl, st);
    }
    *plan_out = pl;
    return check_launch("affine_relu_bwd");
}

size_t bwd_workspace_bytes(int N, int C, int HW) {
    // the plan depends on pointer alignment only through VEC; size for the worse (more splits) case
    const BwdPlan a = plan_bwd(N, C, HW, true), b = plan_bwd(N, C, HW, false);
    const int ns = a.NS > b.NS ? a.NS : b.NS;
    return static_cast<size_t>(ns) * 2 * C * sizeof(double);
}

bool bad_dims(int N, int C, int HW) { return N <= 0 || C <= 0 || HW <= 0; }

}  // namespace

// =============================================================================================
// C ABI
// =============================================================================================
extern "C" {

int deepipr_abi_version(void) { return DEEPIPR_ABI_VERSION; }

const char *deepipr_last_error(void) { return g_err; }

int deepipr_profile_enable(int on) {
    std::lock_guard<std::mutex> lk(g_prof.mu);
    if (on) {
        for (int i = 0; i < DEEPIPR_PROFILE_KERNELS; ++i) { g_prof.total_ms[i] = 0.0; g_prof.launches[i] = 0; }
    }
    g_prof.on = on != 0;
    return DEEPIPR_OK;
}

int deepipr_profile_read(int kernel, double *total_ms, long long *launches) {
    if (kernel < 0 || kernel >= DEEPIPR_PROFILE_KERNELS || !total_ms || !launches)
        return fail(DEEPIPR_EINVAL, "profile_read: bad argument");
    std::lock_guard<std::mutex> lk(g_prof.mu);
    for (auto &p : g_prof.pending) {           // drain everything recorded so far
        float ms = 0.0f;
        if (hipEventSynchronize(p.b) == hipSuccess && hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            g_prof.total_ms[p.k] += ms;
            g_prof.launches[p.k] += 1;
        }
        g_prof.pool.push_back(p.a);
        g_prof.pool.push_back(p.b);
    }
    g_prof.pending.clear();
    *total_ms = g_prof.total_ms[kernel];
    *launches = g_prof.launches[kernel];
    return DEEPIPR_OK;
}

int deepipr_pooled_patch_mean(const float *keys, int nkeys, int B, int Ci, int H, int W, int kh, int kw,
                              int stride, int pad, double *m_out, void *stream) {
    if (!keys || !m_out || nkeys <= 0 || B <= 0 || Ci <= 0 || H <= 0 || W <= 0 || kh <= 0 || kw <= 0 ||
        stride <= 0 || pad < 0)
        return fail(DEEPIPR_EINVAL, "pooled_patch_mean: bad argument");
    const int Ho = (H + 2 * pad - kh) / stride + 1, Wo = (W + 2 * pad - kw) / stride + 1;
    if (Ho <= 0 || Wo <= 0) return fail(DEEPIPR_EINVAL, "pooled_patch_mean: empty conv output");
    const int K = Ci * kh * kw;
    const dim3 grid((K + 3) / 4, nkeys);
    ProfScope prof(DEEPIPR_K_POOLED_PATCH_MEAN, static_cast<hipStream_t>(stream));
    hipLaunchKernelGGL(k_pooled_patch_mean, grid, dim3(kThreads), 0, static_cast<hipStream_t>(stream), keys,
                       B, Ci, H, W, kh, kw, stride, pad, Ho, Wo, m_out);
    return check_launch("pooled_patch_mean");
}

int deepipr_gamma_beta_fwd(const float *W, const double *s, int Co, int K, float *gamma, float *beta,
                           void *stream) {
    if (!W || !s || !gamma || !beta || Co <= 0 || K <= 0) return fail(DEEPIPR_EINVAL, "gamma_beta_fwd: bad argument");
    hipStream_t st = static_cast<hipStream_t>(stream);
    ProfScope prof(DEEPIPR_K_GAMMA_BETA_FWD, st);
    const bool vec = K % 4 == 0 && aligned16(W) && aligned16(s);
    if (Co >= 2 * kRowPairMinCo) {               // enough rows to fill the chip with two per workgroup
        const dim3 grid((Co + 1) / 2);
        if (vec) hipLaunchKernelGGL((k_gamma_beta<true, 2>), grid, dim3(kThreads), 0, st, W, s, Co, K, gamma, beta);
        else hipLaunchKernelGGL((k_gamma_beta<false, 2>), grid, dim3(kThreads), 0, st, W, s, Co, K, gamma, beta);
    } else {
        if (vec) hipLaunchKernelGGL((k_gamma_beta<true, 1>), dim3(Co), dim3(kThreads), 0, st, W, s, Co, K, gamma, beta);
        else hipLaunchKernelGGL((k_gamma_beta<false, 1>), dim3(Co), dim3(kThreads), 0, st, W, s, Co, K, gamma, beta);
    }
    return check_launch("gamma_beta_fwd");
}

int deepipr_gamma_beta_bwd(const float *dgamma, const float *dbeta, const double *s, int Co, int K,
                           float *dW, void *stream) {
    if (!dgamma || !dbeta || !s || !dW || Co <= 0 || K <= 0)
        return fail(DEEPIPR_EINVAL, "gamma_beta_bwd: bad argument");
    hipStream_t st = static_cast<hipStream_t>(stream);
    ProfScope prof(DEEPIPR_K_GAMMA_BETA_BWD, st);
    const bool vec = K % 4 == 0 && aligned16(dW) && aligned16(s);
    if (Co >= 2 * kRowPairMinCo) {
        const dim3 grid((Co + 1) / 2);
        if (vec) hipLaunchKernelGGL((k_gamma_beta_bwd<true, 2>), grid, dim3(kThreads), 0, st, dgamma, dbeta, s, Co, K, dW);
        else hipLaunchKernelGGL((k_gamma_beta_bwd<false, 2>), grid, dim3(kThreads), 0, st, dgamma, dbeta, s, Co, K, dW);
    } else {
        if (vec) hipLaunchKernelGGL((k_gamma_beta_bwd<true, 1>), dim3(Co), dim3(kThreads), 0, st, dgamma, dbeta, s, Co, K, dW);
        else hipLaunchKernelGGL((k_gamma_beta_bwd<false, 1>), dim3(Co), dim3(kThreads), 0, st, dgamma, dbeta, s, Co, K, dW);
    }
    return check_launch("gamma_beta_bwd");
}

size_t deepipr_gamma_beta_dkey_workspace_bytes(int Ci, int kh, int kw) {
    return static_cast<size_t>(kDkeySplit) * 2 * Ci * kh * kw * sizeof(double);
}

int deepipr_gamma_beta_dkey(const float *dgamma, const float *dbeta, const float *W, int Co, int B, int Ci,
                            int H, int Wd, int kh, int kw, int stride, int pad, float *dkeys,
                            void *workspace, void *stream) {
    if (!dgamma || !dbeta || !W || !dkeys || !workspace || Co <= 0 || B <= 0 || Ci <= 0 || H <= 0 ||
        Wd <= 0 || kh <= 0 || kw <= 0 || stride <= 0 || pad < 0)
        return fail(DEEPIPR_EINVAL, "gamma_beta_dkey: bad argument");
    const int Ho = (H + 2 * pad - kh) / stride + 1, Wo = (Wd + 2 * pad - kw) / stride + 1;
    if (Ho <= 0 || Wo <= 0) return fail(DEEPIPR_EINVAL, "gamma_beta_dkey: empty conv output");
    const int K = Ci * kh * kw;
    hipStream_t st = static_cast<hipStream_t>(stream);
    double *part = static_cast<double *>(workspace);
    ProfScope prof(DEEPIPR_K_DKEY, st);
    hipLaunchKernelGGL(k_dkey_colsum, dim3((K + kThreads - 1) / kThreads, kDkeySplit), dim3(kThreads), 0, st,
                       dgamma, dbeta, W, Co, K, part);
    const int total = 2 * B * Ci * H * Wd;
    const double inv_n = 1.0 / (static_cast<double>(B) * Ho * Wo);
    hipLaunchKernelGGL(k_dkey_gather, dim3((total + kThreads - 1) / kThreads), dim3(kThreads), 0, st, part, K,
                       B, Ci, H, Wd, kh, kw, stride, pad, Ho, Wo, inv_n, dkeys);
    return check_launch("gamma_beta_dkey");
}

int deepipr_affine_relu_fwd(const float *xhat, const float *gamma, const float *beta, float *y, int N, int C,
                            int HW, int relu, void *stream) {
    if (!xhat || !gamma || !beta || !y || bad_dims(N, C, HW))
        return fail(DEEPIPR_EINVAL, "affine_relu_fwd: bad argument");
    SignArgs sa{};
    return launch_affine_fwd(xhat, gamma, beta, y, N, C, HW, relu, false, sa, static_cast<hipStream_t>(stream));
}

size_t deepipr_affine_relu_bwd_workspace_bytes(int N, int C, int HW) {
    if (bad_dims(N, C, HW)) return 0;
    return bwd_workspace_bytes(N, C, HW);
}

int deepipr_affine_relu_bwd(const float *dy, const float *xhat, const float *gamma, const float *beta,
                            float *dxhat, float *dgamma, float *dbeta, int N, int C, int HW, int relu,
                            void *workspace, void *stream) {
    if (!dy || !xhat || !gamma || !beta || !dxhat || !dgamma || !dbeta || !workspace || bad_dims(N, C, HW))
        return fail(DEEPIPR_EINVAL, "affine_relu_bwd: bad argument");
    hipStream_t st = static_cast<hipStream_t>(stream);
    BwdPlan pl;
    double *part = static_cast<double *>(workspace);
    int rc = launch_affine_bwd(dy, xhat, gamma, beta, dxhat, part, N, C, HW, relu, &pl, st);
    if (rc != DEEPIPR_OK) return rc;
    ProfScope prof(DEEPIPR_K_REDUCE_PARTIALS, st);
    hipLaunchKernelGGL(k_reduce_partials, dim3((2 * C + kThreads - 1) / kThreads), dim3(kThreads), 0, st, part,
                       pl.NS, C, dgamma, dbeta);
    return check_launch("affine_relu_bwd(finish)");
}

int deepipr_sign_loss_fwd(const float *gamma, const float *b, float alpha, float margin, float l2, int C,
                          float *loss, float *acc, int8_t *bits, void *stream) {
    if (!gamma || !b || !loss || !acc || C <= 0) return fail(DEEPIPR_EINVAL, "sign_loss_fwd: bad argument");
    ProfScope prof(DEEPIPR_K_SIGN_LOSS_FWD, static_cast<hipStream_t>(stream));
    hipLaunchKernelGGL(k_sign_loss_fwd, dim3(1), dim3(kThreads), 0, static_cast<hipStream_t>(stream), gamma, b,
                       alpha, margin, l2, C, loss, acc, bits);
    return check_launch("sign_loss_fwd");
}

int deepipr_sign_loss_bwd(const float *dloss, const float *gamma, const float *b, float alpha, float margin,
                          float l2, int C, float *dgamma, void *stream) {
    if (!dloss || !gamma || !b || !dgamma || C <= 0) return fail(DEEPIPR_EINVAL, "sign_loss_bwd: bad argument");
    ProfScope prof(DEEPIPR_K_SIGN_LOSS_BWD, static_cast<hipStream_t>(stream));
    hipLaunchKernelGGL(k_sign_loss_bwd, dim3((C + kThreads - 1) / kThreads), dim3(kThreads), 0,
                       static_cast<hipStream_t>(stream), dloss, gamma, b, alpha, margin, l2, C, dgamma);
    return check_launch("sign_loss_bwd");
}

int deepipr_passport_fwd(const float *xhat, const float *W, const double *s, const float *b,
                         float alpha, float margin, float l2, int N, int C, int HW, int K, int relu, float *y,
                         float *gamma, float *beta, float *loss, float *acc, int8_t *bits, void *stream) {
    if (!xhat || !W || !s || !y || !gamma || !beta || bad_dims(N, C, HW) || K <= 0)
        return fail(DEEPIPR_EINVAL, "passport_fwd: bad argument");
    const bool with_sign = loss != nullptr;
    if (with_sign && (!b || !acc)) return fail(DEEPIPR_EINVAL, "passport_fwd: sign loss needs b and acc");
    int rc = deepipr_gamma_beta_fwd(W, s, C, K, gamma, beta, stream);
    if (rc != DEEPIPR_OK) return rc;
    SignArgs sa{b, alpha, margin, l2, loss, acc, bits};
    return launch_affine_fwd(xhat, gamma, beta, y, N, C, HW, relu, with_sign, sa, static_cast<hipStream_t>(stream));
}

size_t deepipr_passport_bwd_workspace_bytes(int N, int C, int HW) {
    return deepipr_affine_relu_bwd_workspace_bytes(N, C, HW);
}

int deepipr_passport_bwd(const float *dy, const float *xhat, const float *gamma, const float *beta,
                         const double *s, const float *b, float alpha, float margin, float l2,
                         const float *dloss, const float *dgamma_extra, const float *dbeta_extra, int N, int C,
                         int HW, int K, int relu,
                         float *dxhat, float *dW, float *dgamma, float *dbeta, void *workspace, void *stream) {
    if (!dy || !xhat || !gamma || !beta || !s || !dxhat || !dW || !dgamma || !dbeta || !workspace ||
        bad_dims(N, C, HW) || K <= 0)
        return fail(DEEPIPR_EINVAL, "passport_bwd: bad argument");
    if (dloss && !b) return fail(DEEPIPR_EINVAL, "passport_bwd: sign loss needs b");
    hipStream_t st = static_cast<hipStream_t>(stream);
    BwdPlan pl;
    double *part = static_cast<double *>(workspace);
    int rc = launch_affine_bwd(dy, xhat, gamma, beta, dxhat, part, N, C, HW, relu, &pl, st);
    if (rc != DEEPIPR_OK) return rc;
    ProfScope prof(DEEPIPR_K_PASSPORT_BWD_FINISH, st);
    const bool vec = K % 4 == 0 && aligned16(dW) && aligned16(s);
#define DEEPIPR_FINISH_ARGS part, pl.NS, C, gamma, b, alpha, margin, l2, dloss, dgamma_extra, dbeta_extra, s, K, dgamma, dbeta, dW
    if (C >= 2 * kRowPairMinCo) {
        const dim3 grid((C + 1) / 2);
        if (vec) hipLaunchKernelGGL((k_passport_bwd_finish<true, 2>), grid, dim3(kThreads), 0, st, DEEPIPR_FINISH_ARGS);
        else hipLaunchKernelGGL((k_passport_bwd_finish<false, 2>), grid, dim3(kThreads), 0, st, DEEPIPR_FINISH_ARGS);
    } else {
        if (vec) hipLaunchKernelGGL((k_passport_bwd_finish<true, 1>), dim3(C), dim3(kThreads), 0, st, DEEPIPR_FINISH_ARGS);
        else hipLaunchKernelGGL((k_passport_bwd_finish<false, 1>), dim3(C), dim3(kThreads), 0, st, DEEPIPR_FINISH_ARGS);
    }
#undef DEEPIPR_FINISH_ARGS
    return check_launch("passport_bwd(finish)");
}

}  // extern "C"
