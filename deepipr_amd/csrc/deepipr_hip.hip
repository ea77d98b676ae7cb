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
//     combines go through LDS, cross-workgroup combines are fixed-order partial sums (no float atomics):
//     bit-reproducible.  They are finished in the prologue of the NEXT kernel, except in the
//     register-resident single-pass kernels (k_bn_res_*), where the few workgroups of one channel exchange
//     two doubles inside the launch (self-validating 8-byte {payload, tag} granules, sc1 stores / loads, bounded wait).
//   * 16 B per lane (float4) global accesses wherever the plane size allows; streaming grids are sized to
//     >= 4 workgroups per CU (256 CUs) and capped at 2048 with grid-stride loops; the resident kernels use
//     one 1024-thread workgroup per CU and keep the layer's activations in the 128 MB register file.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <atomic>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/deepipr_hip.h"
#include "deepipr_conv_plan.h"

namespace {

constexpr int kThreads = 256;           // 4 wavefronts of 64
constexpr int kWave = 64;
constexpr int kMaxGrid = 2048;          // 8 workgroups per CU on 256 CUs
constexpr int kDkeySplit = 16;
constexpr int kRowPairMinCo = 256;     // W rows are processed two per workgroup once Co >= 512

thread_local char g_err[512] = "";

int device_cu_count();                  // defined with the single-pass planner below

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
    static const bool debug_sync = getenv("DEEPIPR_DEBUG_SYNC") != nullptr;      // triage only: never set in production
    if (debug_sync) {
        fprintf(stderr, "[deepipr] %s\n", what);
        e = hipDeviceSynchronize();
        if (e != hipSuccess) return fail(DEEPIPR_ELAUNCH, "%s (sync): %s", what, hipGetErrorString(e));
    }
    return DEEPIPR_OK;
}

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// ---------------------------------------------------------------------------------------------
// Opt-in in-situ kernel timing (deepipr_profile_*): when enabled, every kernel is dispatched with a
// start and a stop hipEvent attached to its own dispatch packet; durations are read back later with
// deepipr_profile_read.  Off by default (one bool load per launch); must stay off during hipGraph capture.
struct ProfState {
    std::mutex mu;
    bool on = false;
    std::vector<hipEvent_t> pool;
    struct Pending { int k; hipEvent_t a, b; double bytes; int scope; };
    std::vector<Pending> pending;
    double total_ms[DEEPIPR_PROFILE_KERNELS] = {};
    long long launches[DEEPIPR_PROFILE_KERNELS] = {};
    double total_bytes[DEEPIPR_PROFILE_KERNELS] = {};   // algorithmic bytes of the timed launches
    // the same three, of the launches issued while a caller-declared scope was open (deepipr_profile_scope: the
    // passport layers of a net, so that their kernels can be reported apart from the plain norm layers')
    int scope = 0;
    double scope_ms[DEEPIPR_PROFILE_KERNELS] = {};
    long long scope_launches[DEEPIPR_PROFILE_KERNELS] = {};
    double scope_bytes[DEEPIPR_PROFILE_KERNELS] = {};
};
ProfState g_prof;

struct ProfScope {
    int k;
    hipEvent_t a = nullptr, b = nullptr;
    bool used = false;
    double bytes = 0.0;             // algorithmic HBM bytes of this launch (streaming kernels set it)
    ProfScope(int kernel, hipStream_t) : k(kernel) {
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
        if (!a || !b) a = b = nullptr;
    }
    ~ProfScope() {
        if (!a) return;
        std::lock_guard<std::mutex> lk(g_prof.mu);
        if (used) g_prof.pending.push_back({k, a, b, bytes, g_prof.scope});
        else { g_prof.pool.push_back(a); g_prof.pool.push_back(b); }
    }
};

// Launch `kernel`; when timing is on, the FIRST launch of the scope carries the scope's two events as the
// dispatch's own start/stop events (hipExtLaunchKernelGGL): their elapsed time is the kernel's execution time,
// the same quantity rocprofv3's kernel trace reports, with no event-record overhead in it.
#define DEEPIPR_LAUNCH(prof, kernel, grid, block, st, ...)                                              \
    do {                                                                                                \
        if ((prof).a && !(prof).used) {                                                                 \
            (prof).used = true;                                                                         \
            hipExtLaunchKernelGGL(kernel, grid, block, 0, st, (prof).a, (prof).b, 0, __VA_ARGS__);      \
        } else {                                                                                        \
            hipLaunchKernelGGL(kernel, grid, block, 0, st, __VA_ARGS__);                                \
        }                                                                                               \
    } while (0)

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

// ReLU as ATen computes it (clamp_min: NaN propagates).  fmaxf(NaN, 0) would return 0 and silently swallow a
// poisoned statistic (res_exchange time-out) or an upstream NaN.
__device__ __forceinline__ float relu1(float y) { return y < 0.0f ? 0.0f : y; }

// y = relu?(g*x + b) with the reference's two roundings (aten::mul then aten::add).
template <bool RELU>
__device__ __forceinline__ float affine1(float x, float g, float b) {
    float y = __fadd_rn(__fmul_rn(g, x), b);
    return RELU ? relu1(y) : y;
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
// as[r], ab[r] = this thread's partial dot products of rows co0..co0+RPW-1 with the two pooled vectors.
template <bool VEC, int RPW>
__device__ __forceinline__ void gemv_rows(const float *__restrict__ W, const double *__restrict__ s, int Co, int K,
                                          int co0, double (&as)[RPW], double (&ab)[RPW]) {
    const double *ss = s, *sb = s + K;
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
}

template <bool VEC, int RPW>
__global__ __launch_bounds__(kThreads) void k_gamma_beta(
    const float *__restrict__ W, const double *__restrict__ s, int Co, int K,
    float *__restrict__ gamma, float *__restrict__ beta) {
    __shared__ double red[8 * RPW];
    const int co0 = blockIdx.x * RPW;
    double as[RPW], ab[RPW];
    gemv_rows<VEC, RPW>(W, s, Co, K, co0, as, ab);
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
// ACC: dW already holds the data convolution's wgrad; the rank-2 update is added to it in place (8 B per weight
// instead of a fresh 4 B write + autograd's 12 B add kernel).
template <bool VEC, int RPW, bool ACC = false>
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
                float4 *dst = reinterpret_cast<float4 *>(dW + static_cast<size_t>(co0 + r) * K) + q;
                if (ACC) {
                    const float4 w = *dst;
                    o = make_float4(w.x + o.x, w.y + o.y, w.z + o.z, w.w + o.w);
                }
                *dst = o;
            }
        }
    } else {
        for (int k = threadIdx.x; k < K; k += kThreads) {
            const float ms = static_cast<float>(ss[k]), mb = static_cast<float>(sb[k]);
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                if (co0 + r >= Co) break;
                float *dst = dW + static_cast<size_t>(co0 + r) * K + k;
                const float o = fmaf(dg[r], ms, db[r] * mb);
                *dst = ACC ? *dst + o : o;
            }
        }
    }
}

template <bool VEC, int RPW, bool ACC>
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
    write_dw_rows<VEC, RPW, ACC>(dW, s, Co, K, co0, dg, db);
}

// ---- batched / deep-pipelined form -------------------------------------------------------------------------
// One launch computes gamma and beta of up to kGemvMaxLayers passport layers (ResNet18: the five layer4 weights,
// 33.6 MB): grid = sum over layers of ceil(Co / RPW) workgroups, the per-layer descriptors travel by value in the
// kernel arguments (no device table to maintain; frozen with the launch in a hipGraph).  Round 2 launched one
// k_gamma_beta per layer with ONE 4-wave workgroup per CU, whose threads walked K/1024 dependent trips of two float4
// loads: 5 us per 9.4 MB weight = 0.17 of the HBM roofline, latency- not bandwidth-bound (PMC: 1.07x the algorithmic
// bytes).  Here every thread issues ALL its loads of W (kGemvPre float4 per row, K <= 5120 in one trip) before the first
// FMA, and the batch puts 5-10 workgroups on every CU.
constexpr int kGemvMaxLayers = 16;
constexpr int kGemvPre = 5;

struct GemvLayer {
    const float *W;
    const double *m;            // [2][K] pooled patches (scale key, bias key)
    float *gamma, *beta;
    int Co, K, block0, vec;     // first workgroup of this layer; vec: K % 4 == 0 and 16-byte aligned rows
};
struct GemvBatch {
    GemvLayer L[kGemvMaxLayers];
    int n;
};

template <int RPW>
__device__ __forceinline__ void gemv_rows_pre(const float *__restrict__ W, const double *__restrict__ s, int Co, int K,
                                              int co0, double (&as)[RPW], double (&ab)[RPW]) {
    const double2 *ss2 = reinterpret_cast<const double2 *>(s);
    const double2 *sb2 = reinterpret_cast<const double2 *>(s + K);
    const float4 *row[RPW];
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        as[r] = 0.0;
        ab[r] = 0.0;
        row[r] = reinterpret_cast<const float4 *>(W + static_cast<size_t>(min(co0 + r, Co - 1)) * K);
    }
    const int K4 = K / 4;
    for (int base = 0; base < K4; base += kGemvPre * kThreads) {
        float4 w[RPW][kGemvPre];
#pragma unroll
        for (int p = 0; p < kGemvPre; ++p) {
            const int q = base + p * kThreads + static_cast<int>(threadIdx.x);
#pragma unroll
            for (int r = 0; r < RPW; ++r) w[r][p] = q < K4 ? row[r][q] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
#pragma unroll
        for (int p = 0; p < kGemvPre; ++p) {
            const int q = base + p * kThreads + static_cast<int>(threadIdx.x);
            if (q < K4) {
                const double2 s0 = ss2[2 * q], s1 = ss2[2 * q + 1];
                const double2 b0 = sb2[2 * q], b1 = sb2[2 * q + 1];
#pragma unroll
                for (int r = 0; r < RPW; ++r) {
                    as[r] = fma(static_cast<double>(w[r][p].x), s0.x, as[r]);
                    as[r] = fma(static_cast<double>(w[r][p].y), s0.y, as[r]);
                    as[r] = fma(static_cast<double>(w[r][p].z), s1.x, as[r]);
                    as[r] = fma(static_cast<double>(w[r][p].w), s1.y, as[r]);
                    ab[r] = fma(static_cast<double>(w[r][p].x), b0.x, ab[r]);
                    ab[r] = fma(static_cast<double>(w[r][p].y), b0.y, ab[r]);
                    ab[r] = fma(static_cast<double>(w[r][p].z), b1.x, ab[r]);
                    ab[r] = fma(static_cast<double>(w[r][p].w), b1.y, ab[r]);
                }
            }
        }
    }
}

template <int RPW>
__global__ __launch_bounds__(kThreads) void k_gamma_beta_multi(GemvBatch B) {
    __shared__ double red[8 * RPW];
    int li = 0;
    for (int i = 1; i < B.n; ++i)
        if (static_cast<int>(blockIdx.x) >= B.L[i].block0) li = i;          // uniform over the workgroup
    const float *W = B.L[li].W;
    const double *m = B.L[li].m;
    float *gamma = B.L[li].gamma, *beta = B.L[li].beta;
    const int Co = B.L[li].Co, K = B.L[li].K;
    const int co0 = (static_cast<int>(blockIdx.x) - B.L[li].block0) * RPW;
    double as[RPW], ab[RPW];
    if (B.L[li].vec) gemv_rows_pre<RPW>(W, m, Co, K, co0, as, ab);
    else gemv_rows<false, RPW>(W, m, Co, K, co0, as, ab);
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

// The rank-2 update dW[co, :] (+)= dgamma[co] * m_scale + dbeta[co] * m_bias of several layers in one launch, the
// accumulate form with all of a thread's loads of dW in flight before the first store.
struct Rank2Layer {
    const float *dg, *db;
    const double *m;
    float *dW;
    int Co, K, block0, vec;
};
struct Rank2Batch {
    Rank2Layer L[kGemvMaxLayers];
    int n;
};

template <int RPW, bool ACC>
__global__ __launch_bounds__(kThreads) void k_rank2_multi(Rank2Batch B) {
    int li = 0;
    for (int i = 1; i < B.n; ++i)
        if (static_cast<int>(blockIdx.x) >= B.L[i].block0) li = i;
    const double *s = B.L[li].m;
    float *dW = B.L[li].dW;
    const int Co = B.L[li].Co, K = B.L[li].K;
    const int co0 = (static_cast<int>(blockIdx.x) - B.L[li].block0) * RPW;
    float dg[RPW], db[RPW];
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const int co = min(co0 + r, Co - 1);
        dg[r] = B.L[li].dg[co];
        db[r] = B.L[li].db[co];
    }
    if (!B.L[li].vec) {
        write_dw_rows<false, RPW, ACC>(dW, s, Co, K, co0, dg, db);
        return;
    }
    const double2 *ss2 = reinterpret_cast<const double2 *>(s);
    const double2 *sb2 = reinterpret_cast<const double2 *>(s + K);
    const int K4 = K / 4;
    for (int base = 0; base < K4; base += kGemvPre * kThreads) {
        float4 w[RPW][kGemvPre];
        if (ACC) {
#pragma unroll
            for (int p = 0; p < kGemvPre; ++p) {
                const int q = base + p * kThreads + static_cast<int>(threadIdx.x);
#pragma unroll
                for (int r = 0; r < RPW; ++r)
                    w[r][p] = (q < K4 && co0 + r < Co)
                                  ? reinterpret_cast<const float4 *>(dW + static_cast<size_t>(co0 + r) * K)[q]
                                  : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            }
        }
#pragma unroll
        for (int p = 0; p < kGemvPre; ++p) {
            const int q = base + p * kThreads + static_cast<int>(threadIdx.x);
            if (q < K4) {
                const double2 s0 = ss2[2 * q], s1 = ss2[2 * q + 1];
                const double2 b0 = sb2[2 * q], b1 = sb2[2 * q + 1];
                const float4 ms = make_float4(static_cast<float>(s0.x), static_cast<float>(s0.y),
                                              static_cast<float>(s1.x), static_cast<float>(s1.y));
                const float4 mb = make_float4(static_cast<float>(b0.x), static_cast<float>(b0.y),
                                              static_cast<float>(b1.x), static_cast<float>(b1.y));
#pragma unroll
                for (int r = 0; r < RPW; ++r) {
                    if (co0 + r >= Co) break;
                    float4 o;                     // the same roundings as write_dw_rows: fmaf(dg, ms, db * mb) (+ dW)
                    o.x = fmaf(dg[r], ms.x, db[r] * mb.x);
                    o.y = fmaf(dg[r], ms.y, db[r] * mb.y);
                    o.z = fmaf(dg[r], ms.z, db[r] * mb.z);
                    o.w = fmaf(dg[r], ms.w, db[r] * mb.w);
                    if (ACC) o = make_float4(w[r][p].x + o.x, w[r][p].y + o.y, w[r][p].z + o.z, w[r][p].w + o.w);
                    reinterpret_cast<float4 *>(dW + static_cast<size_t>(co0 + r) * K)[q] = o;
                }
            }
        }
    }
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
    // float4 units need every row start (n*C + c0)*P to be a multiple of 4 floats: true when P % 4 == 0, or --
    // 7x7 / 9x9 ... maps -- when C % 4 == 0 and tiles are 4 channels (a unit may then straddle two channels,
    // which the kernels handle per element)
    const bool odd_plane_vec = can_vec && P % 4 != 0 && C % 4 == 0 && P <= kThreads;
    p.VEC = (can_vec && (P % 4 == 0 || odd_plane_vec)) ? 4 : 1;
    const int pu = (P % 4 == 0 || p.VEC == 1) ? P / p.VEC : 0;      // units per plane (0: planes share units)
    p.large = !odd_plane_vec && pu > kThreads;
    if (odd_plane_vec) {
        p.CT = 4;
        p.row_u = P;                                                // 4 planes = P float4 units
        p.npi = kThreads / p.row_u;
        p.tiles = C / 4;
        p.iters = (N + p.npi - 1) / p.npi;
    } else if (!p.large) {
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
    if (pl.CT == 1) {     // one channel per workgroup: butterfly + four LDS words (see k_bn_walk_small)
        double sgx = 0.0, sg = 0.0;
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            sgx += static_cast<double>(a_gx[i]);
            sg += static_cast<double>(a_g[i]);
        }
        double *red = reinterpret_cast<double *>(&sacc[0][0]);
        sgx = block_sum(sgx, red);
        sg = block_sum(sg, red + 4);
        if (t == 0) {
            part[(static_cast<size_t>(blockIdx.y) * 2 + 0) * C + c0] = sgx;
            part[(static_cast<size_t>(blockIdx.y) * 2 + 1) * C + c0] = sg;
        }
        return;
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
// One wavefront per channel: lanes stride over the splits, butterfly at the end (NS reaches the batch size when
// the partials come from the GroupNorm-fused backward).
__global__ __launch_bounds__(kThreads) void k_reduce_partials(
    const double *__restrict__ part, int NS, int C, float *__restrict__ dgamma, float *__restrict__ dbeta) {
    const int c = blockIdx.x * (kThreads / kWave) + (threadIdx.x >> 6);
    if (c >= C) return;                                   // uniform per wavefront
    const int lane = threadIdx.x & 63;
    double ag = 0.0, ab = 0.0;
    for (int sp = lane; sp < NS; sp += kWave) {
        ag += part[(static_cast<size_t>(sp) * 2 + 0) * C + c];
        ab += part[(static_cast<size_t>(sp) * 2 + 1) * C + c];
    }
    ag = wave_sum(ag);
    ab = wave_sum(ab);
    if (lane == 0) {
        dgamma[c] = static_cast<float>(ag);
        dbeta[c] = static_cast<float>(ab);
    }
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
    if (dW) write_dw_rows<VEC, RPW>(dW, s, C, K, co0, dgv, dbv);      // dW == nullptr: the caller accumulates it later
}


// ============================================================================================
// BatchNorm(affine=False, models/layers/passportconv2d.py:58) fused into the passport layer.
//
//   forward :  k_bn_walk<STATS>   per-(split,channel) sums of x, x^2                      (4 B/elt)
//              k_gamma_beta       + finish: mean / invstd / running stats -> channel table
//              k_bn_affine_fwd    y = relu(gamma*((x-mean)*invstd) + beta)               (8 B/elt)
//   backward:  k_bn_walk<BWD>     sums of dz*xhat, dz   (dz = dy masked by the recomputed ReLU) (8 B/elt)
//              k_passport_bwd_finish  dgamma, dbeta, dW + table entries c2 = sum(dz)/M, c3 = sum(dz*xhat)/M
//              k_bn_affine_bwd    dx = gamma*invstd*(dz - c2 - xhat*c3)                  (12 B/elt)
//
// The normalised activation xhat is never materialised; x (the conv output) is the only saved
// activation.  Channel table: tbl[c][8] = {mean, invstd, gamma, beta, c2, c3, -, -}.
// ============================================================================================
constexpr int kTbl = 8;
enum WalkMode { WALK_STATS = 1, WALK_BN_BWD = 2 };

struct BnFinishArgs {
    const double *part;        // [NS][2][C] sums of (x-K) and (x-K)^2 (training) or nullptr (use running stats)
    // The sums are SHIFTED by K[c] = x[0][c][0] (the channel's first element, any sample of the distribution):
    // var = E[(x-K)^2] - E[x-K]^2 has no catastrophic cancellation when |mean| >> std, unlike E[x^2] - mean^2 on
    // fp32 partials (ATen gets the same robustness from Welford updates).
    const float *shift_src;    // x; K[c] = shift_src[c * HW]
    int HW;
    int NS;
    double inv_m;              // 1 / (N*HW)
    double unbias;             // M / (M-1)
    float eps, momentum;
    float *running_mean, *running_var;   // may be nullptr
    long long *num_batches_tracked;      // may be nullptr
    float *tbl;                // [C][8]
};

// All 64 lanes of one wavefront call this for channel c; lane 0 writes.
__device__ __forceinline__ void bn_finish_channel(const BnFinishArgs &f, int C, int c, float gamma, float beta,
                                                  int lane) {
    float mean, invstd;
    if (f.part) {
        double s1 = 0.0, s2 = 0.0;
        for (int sp = lane; sp < f.NS; sp += kWave) {
            s1 += f.part[(static_cast<size_t>(sp) * 2 + 0) * C + c];
            s2 += f.part[(static_cast<size_t>(sp) * 2 + 1) * C + c];
        }
        s1 = wave_sum(s1);
        s2 = wave_sum(s2);
        const double dmu = s1 * f.inv_m;               // mean - K
        double var = s2 * f.inv_m - dmu * dmu;         // biased variance, f64, from the shifted sums
        if (var < 0.0) var = 0.0;
        const double mu = static_cast<double>(f.shift_src[static_cast<size_t>(c) * f.HW]) + dmu;
        mean = static_cast<float>(mu);
        invstd = static_cast<float>(1.0 / sqrt(var + static_cast<double>(f.eps)));
        if (lane == 0 && f.running_mean) {
            f.running_mean[c] = (1.0f - f.momentum) * f.running_mean[c] + f.momentum * mean;
            f.running_var[c] = (1.0f - f.momentum) * f.running_var[c] +
                               f.momentum * static_cast<float>(var * f.unbias);
        }
    } else {                                            // evaluation: running statistics
        mean = f.running_mean[c];
        invstd = static_cast<float>(1.0 / sqrt(static_cast<double>(f.running_var[c]) + static_cast<double>(f.eps)));
    }
    if (lane == 0) {
        float4 *t = reinterpret_cast<float4 *>(f.tbl + static_cast<size_t>(c) * kTbl);
        t[0] = make_float4(mean, invstd, gamma, beta);
        t[1] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
}

// gamma/beta GEMV + BN finish for the same rows (one launch between the stats pass and the apply pass).
template <bool VEC, int RPW>
__global__ __launch_bounds__(kThreads) void k_gamma_beta_bn(
    const float *__restrict__ W, const double *__restrict__ s, int Co, int K,
    float *__restrict__ gamma, float *__restrict__ beta, BnFinishArgs f) {
    __shared__ double red[8 * RPW];
    const int co0 = blockIdx.x * RPW;
    double as[RPW], ab[RPW];
    gemv_rows<VEC, RPW>(W, s, Co, K, co0, as, ab);
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        as[r] = block_sum(as[r], red + 8 * r);
        ab[r] = block_sum(ab[r], red + 8 * r + 4);
        if (co0 + r < Co) {
            const float g = static_cast<float>(as[r]), bt = static_cast<float>(ab[r]);
            if (threadIdx.x == 0) {
                gamma[co0 + r] = g;
                beta[co0 + r] = bt;
            }
            if (threadIdx.x < kWave) bn_finish_channel(f, Co, co0 + r, g, bt, threadIdx.x);
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && f.part && f.num_batches_tracked) *f.num_batches_tracked += 1;
}

// Public branch (learnable gamma/beta given): table only, one wavefront per channel.
__global__ __launch_bounds__(kThreads) void k_bn_table(const float *__restrict__ gamma,
                                                       const float *__restrict__ beta, int C, BnFinishArgs f) {
    const int c = blockIdx.x * (kThreads / kWave) + (threadIdx.x >> 6);
    if (c < C) bn_finish_channel(f, C, c, gamma[c], beta[c], threadIdx.x & 63);
    if (blockIdx.x == 0 && threadIdx.x == 0 && f.part && f.num_batches_tracked) *f.num_batches_tracked += 1;
}

// ---- channel-walk reduction passes (same tiling as k_affine_bwd_*, nothing written but the partials) ----
template <int MODE, bool RELU>
__device__ __forceinline__ void walk_accum(float d, float x, float mean, float invstd, float g, float bt,
                                           float &a0, float &a1) {
    if (MODE == WALK_STATS) {                      // `mean` carries the channel's shift K here
        const float xs = x - mean;
        a0 += xs;
        a1 = fmaf(xs, xs, a1);
    } else {
        const float xh = (x - mean) * invstd;
        float dz = d;
        if (RELU) dz = (__fadd_rn(__fmul_rn(g, xh), bt) > 0.0f) ? dz : 0.0f;
        a0 = fmaf(dz, xh, a0);
        a1 += dz;
    }
}

template <int MODE, int VEC, bool RELU>
__global__ __launch_bounds__(kThreads) void k_bn_walk_small(
    const float *__restrict__ dy, const float *__restrict__ xin, const float *__restrict__ tbl,
    double *__restrict__ part, int N, int C, int P, BwdPlan pl) {
    using U = typename Unit<VEC>::T;
    __shared__ float sacc[2][kThreads * 4];
    const int c0 = blockIdx.x * pl.CT;
    const int ct = min(pl.CT, C - c0);
    const int t = threadIdx.x;
    const int r = t / pl.row_u;
    const int u = t - r * pl.row_u;
    const bool lane_on = (r < pl.npi) && (u * VEC < ct * P);
    float mean[VEC], istd[VEC], g[VEC], bt[VEC], a0[VEC], a1[VEC];
    int slot[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
        a0[i] = 0.0f;
        a1[i] = 0.0f;
        const int er = u * VEC + i;
        const int cc = lane_on ? er / P : 0;
        slot[i] = (cc * pl.npi + r) * P + (er - cc * P);
        mean[i] = istd[i] = g[i] = bt[i] = 0.0f;
        if (MODE == WALK_BN_BWD) {
            const float4 c4 = *reinterpret_cast<const float4 *>(tbl + static_cast<size_t>(c0 + cc) * kTbl);
            mean[i] = c4.x;
            istd[i] = c4.y;
            g[i] = c4.z;
            bt[i] = c4.w;
        } else {
            mean[i] = xin[static_cast<size_t>(c0 + cc) * P];          // shift K[c] = x[0][c][0]
        }
    }
    if (lane_on) {
        const int it0 = blockIdx.y * pl.ips, it1 = min(pl.iters, it0 + pl.ips);
        size_t off = (static_cast<size_t>(it0 * pl.npi + r) * C + c0) * P + static_cast<size_t>(u) * VEC;
        const size_t hop = static_cast<size_t>(C) * P * pl.npi;
        int n = it0 * pl.npi + r;
        U vdy{}, vx{};
        if (n < N) {
            vx = *reinterpret_cast<const U *>(xin + off);
            if (MODE == WALK_BN_BWD) vdy = *reinterpret_cast<const U *>(dy + off);
        }
        for (int it = it0; it < it1 && n < N; ++it, n += pl.npi, off += hop) {
            U ndy{}, nx{};
            if (it + 1 < it1 && n + pl.npi < N) {
                nx = *reinterpret_cast<const U *>(xin + off + hop);
                if (MODE == WALK_BN_BWD) ndy = *reinterpret_cast<const U *>(dy + off + hop);
            }
            float d[VEC], x[VEC];
            Unit<VEC>::get(vdy, d);
            Unit<VEC>::get(vx, x);
#pragma unroll
            for (int i = 0; i < VEC; ++i) walk_accum<MODE, RELU>(d[i], x[i], mean[i], istd[i], g[i], bt[i], a0[i], a1[i]);
            vdy = ndy;
            vx = nx;
        }
    }
    if (pl.CT == 1) {
        // one channel per workgroup (planes of >= 64 floats): every thread holds partial sums of that channel,
        // so a butterfly per wave + four LDS words finish it -- no staging of 1024 partials for one wave to re-read
        double s0 = 0.0, s1 = 0.0;
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            s0 += static_cast<double>(a0[i]);
            s1 += static_cast<double>(a1[i]);
        }
        double *red = reinterpret_cast<double *>(&sacc[0][0]);
        s0 = block_sum(s0, red);
        s1 = block_sum(s1, red + 4);
        if (t == 0) {
            part[(static_cast<size_t>(blockIdx.y) * 2 + 0) * C + c0] = s0;
            part[(static_cast<size_t>(blockIdx.y) * 2 + 1) * C + c0] = s1;
        }
        return;
    }
    if (lane_on) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            sacc[0][slot[i]] = a0[i];
            sacc[1][slot[i]] = a1[i];
        }
    }
    __syncthreads();
    const int wave = t >> 6, lane = t & 63;
    const int cnt = pl.npi * P;
    for (int cc = wave; cc < ct; cc += kThreads / kWave) {
        const float *p0 = sacc[0] + cc * cnt, *p1 = sacc[1] + cc * cnt;
        double s0 = 0.0, s1 = 0.0;
        for (int e = lane; e < cnt; e += kWave) {
            s0 += static_cast<double>(p0[e]);
            s1 += static_cast<double>(p1[e]);
        }
        s0 = wave_sum(s0);
        s1 = wave_sum(s1);
        if (lane == 0) {
            part[(static_cast<size_t>(blockIdx.y) * 2 + 0) * C + c0 + cc] = s0;
            part[(static_cast<size_t>(blockIdx.y) * 2 + 1) * C + c0 + cc] = s1;
        }
    }
}

template <int MODE, int VEC, bool RELU>
__global__ __launch_bounds__(kThreads) void k_bn_walk_large(
    const float *__restrict__ dy, const float *__restrict__ xin, const float *__restrict__ tbl,
    double *__restrict__ part, int N, int C, int P, BwdPlan pl) {
    using U = typename Unit<VEC>::T;
    __shared__ double red[8];
    const int c = blockIdx.x;
    float mean = 0.0f, istd = 0.0f, g = 0.0f, bt = 0.0f;
    if (MODE == WALK_BN_BWD) {
        const float4 c4 = *reinterpret_cast<const float4 *>(tbl + static_cast<size_t>(c) * kTbl);
        mean = c4.x;
        istd = c4.y;
        g = c4.z;
        bt = c4.w;
    } else {
        mean = xin[static_cast<size_t>(c) * P];                       // shift K[c] = x[0][c][0]
    }
    double A0 = 0.0, A1 = 0.0;
    const int n0 = blockIdx.y * pl.ips, n1 = min(N, n0 + pl.ips);
    for (int n = n0; n < n1; ++n) {
        const size_t base = (static_cast<size_t>(n) * C + c) * P;
        float p0 = 0.0f, p1 = 0.0f;
        for (int u = threadIdx.x; u < pl.row_u; u += kThreads) {
            const size_t off = base + static_cast<size_t>(u) * VEC;
            U vdy{};
            const U vx = *reinterpret_cast<const U *>(xin + off);
            if (MODE == WALK_BN_BWD) vdy = *reinterpret_cast<const U *>(dy + off);
            float d[VEC], x[VEC];
            Unit<VEC>::get(vdy, d);
            Unit<VEC>::get(vx, x);
#pragma unroll
            for (int i = 0; i < VEC; ++i) walk_accum<MODE, RELU>(d[i], x[i], mean, istd, g, bt, p0, p1);
        }
        A0 += static_cast<double>(p0);
        A1 += static_cast<double>(p1);
    }
    A0 = block_sum(A0, red);
    A1 = block_sum(A1, red + 4);
    if (threadIdx.x == 0) {
        part[(static_cast<size_t>(blockIdx.y) * 2 + 0) * C + c] = A0;
        part[(static_cast<size_t>(blockIdx.y) * 2 + 1) * C + c] = A1;
    }
}

// ---- streaming passes ----
template <bool RELU>
__device__ __forceinline__ float bn_affine1(float x, const float4 &c) {       // c = {mean, invstd, gamma, beta}
    const float y = __fadd_rn(__fmul_rn(c.z, (x - c.x) * c.y), c.w);
    return RELU ? relu1(y) : y;
}

template <bool RELU>
__global__ __launch_bounds__(kThreads) void k_bn_affine_fwd_v4(
    const float4 *__restrict__ x, const float *__restrict__ tbl, float4 *__restrict__ y, unsigned n4,
    FastDiv p4div, FastDiv cdiv, unsigned C, int with_sign, SignArgs sa, const float *__restrict__ gamma) {
    __shared__ double red[12];
    unsigned nblk = gridDim.x;
    if (with_sign) {
        nblk -= 1;
        if (blockIdx.x == nblk) {
            sign_loss_block(gamma, sa.b, sa.alpha, sa.margin, sa.l2, static_cast<int>(C), sa.loss, sa.acc,
                            sa.bits, red);
            return;
        }
    }
    const unsigned step = nblk * kThreads;
    unsigned q = blockIdx.x * kThreads + threadIdx.x;
    for (; q + step < n4; q += 2 * step) {
        const unsigned q1 = q + step;
        const float4 v0 = x[q], v1 = x[q1];
        const unsigned p0 = fdiv(q, p4div), p1 = fdiv(q1, p4div);
        const unsigned c0 = p0 - fdiv(p0, cdiv) * C, c1 = p1 - fdiv(p1, cdiv) * C;
        const float4 k0 = *reinterpret_cast<const float4 *>(tbl + static_cast<size_t>(c0) * kTbl);
        const float4 k1 = *reinterpret_cast<const float4 *>(tbl + static_cast<size_t>(c1) * kTbl);
        y[q] = make_float4(bn_affine1<RELU>(v0.x, k0), bn_affine1<RELU>(v0.y, k0), bn_affine1<RELU>(v0.z, k0),
                           bn_affine1<RELU>(v0.w, k0));
        y[q1] = make_float4(bn_affine1<RELU>(v1.x, k1), bn_affine1<RELU>(v1.y, k1), bn_affine1<RELU>(v1.z, k1),
                            bn_affine1<RELU>(v1.w, k1));
    }
    if (q < n4) {
        const unsigned plane = fdiv(q, p4div);
        const unsigned c = plane - fdiv(plane, cdiv) * C;
        const float4 k = *reinterpret_cast<const float4 *>(tbl + static_cast<size_t>(c) * kTbl);
        const float4 v = x[q];
        y[q] = make_float4(bn_affine1<RELU>(v.x, k), bn_affine1<RELU>(v.y, k), bn_affine1<RELU>(v.z, k),
                           bn_affine1<RELU>(v.w, k));
    }
}

template <bool RELU>
__global__ __launch_bounds__(kThreads) void k_bn_affine_fwd_s(
    const float *__restrict__ x, const float *__restrict__ tbl, float *__restrict__ y, unsigned n, FastDiv pdiv,
    FastDiv cdiv, unsigned C, int with_sign, SignArgs sa, const float *__restrict__ gamma) {
    __shared__ double red[12];
    unsigned nblk = gridDim.x;
    if (with_sign) {
        nblk -= 1;
        if (blockIdx.x == nblk) {
            sign_loss_block(gamma, sa.b, sa.alpha, sa.margin, sa.l2, static_cast<int>(C), sa.loss, sa.acc,
                            sa.bits, red);
            return;
        }
    }
    const unsigned step = nblk * kThreads;
    for (unsigned i = blockIdx.x * kThreads + threadIdx.x; i < n; i += step) {
        const unsigned plane = fdiv(i, pdiv);
        const unsigned c = plane - fdiv(plane, cdiv) * C;
        const float4 k = *reinterpret_cast<const float4 *>(tbl + static_cast<size_t>(c) * kTbl);
        y[i] = bn_affine1<RELU>(x[i], k);
    }
}

// dx = gamma*invstd * (dz - c2 - xhat*c3), dz = dy masked by the recomputed ReLU; c = table rows.
template <bool RELU>
__device__ __forceinline__ float bn_bwd1(float d, float x, const float4 &a, const float4 &b) {
    const float xh = (x - a.x) * a.y;
    float dz = d;
    if (RELU) dz = (__fadd_rn(__fmul_rn(a.z, xh), a.w) > 0.0f) ? dz : 0.0f;
    return (a.z * a.y) * (dz - b.x - xh * b.y);
}

template <bool RELU>
__global__ __launch_bounds__(kThreads) void k_bn_affine_bwd_v4(
    const float4 *__restrict__ dy, const float4 *__restrict__ x, const float *__restrict__ tbl,
    float4 *__restrict__ dx, unsigned n4, FastDiv p4div, FastDiv cdiv, unsigned C) {
    const unsigned step = gridDim.x * kThreads;
    unsigned q = blockIdx.x * kThreads + threadIdx.x;
    for (; q + step < n4; q += 2 * step) {
        const unsigned q1 = q + step;
        const float4 d0 = dy[q], x0 = x[q], d1 = dy[q1], x1 = x[q1];
        const unsigned p0 = fdiv(q, p4div), p1 = fdiv(q1, p4div);
        const unsigned c0 = p0 - fdiv(p0, cdiv) * C, c1 = p1 - fdiv(p1, cdiv) * C;
        const float4 *t0 = reinterpret_cast<const float4 *>(tbl + static_cast<size_t>(c0) * kTbl);
        const float4 *t1 = reinterpret_cast<const float4 *>(tbl + static_cast<size_t>(c1) * kTbl);
        const float4 a0 = t0[0], b0 = t0[1], a1 = t1[0], b1 = t1[1];
        dx[q] = make_float4(bn_bwd1<RELU>(d0.x, x0.x, a0, b0), bn_bwd1<RELU>(d0.y, x0.y, a0, b0),
                            bn_bwd1<RELU>(d0.z, x0.z, a0, b0), bn_bwd1<RELU>(d0.w, x0.w, a0, b0));
        dx[q1] = make_float4(bn_bwd1<RELU>(d1.x, x1.x, a1, b1), bn_bwd1<RELU>(d1.y, x1.y, a1, b1),
                             bn_bwd1<RELU>(d1.z, x1.z, a1, b1), bn_bwd1<RELU>(d1.w, x1.w, a1, b1));
    }
    if (q < n4) {
        const unsigned plane = fdiv(q, p4div);
        const unsigned c = plane - fdiv(plane, cdiv) * C;
        const float4 *t = reinterpret_cast<const float4 *>(tbl + static_cast<size_t>(c) * kTbl);
        const float4 a = t[0], b = t[1];
        const float4 d = dy[q], v = x[q];
        dx[q] = make_float4(bn_bwd1<RELU>(d.x, v.x, a, b), bn_bwd1<RELU>(d.y, v.y, a, b),
                            bn_bwd1<RELU>(d.z, v.z, a, b), bn_bwd1<RELU>(d.w, v.w, a, b));
    }
}

template <bool RELU>
__global__ __launch_bounds__(kThreads) void k_bn_affine_bwd_s(
    const float *__restrict__ dy, const float *__restrict__ x, const float *__restrict__ tbl,
    float *__restrict__ dx, unsigned n, FastDiv pdiv, FastDiv cdiv, unsigned C) {
    const unsigned step = gridDim.x * kThreads;
    for (unsigned i = blockIdx.x * kThreads + threadIdx.x; i < n; i += step) {
        const unsigned plane = fdiv(i, pdiv);
        const unsigned c = plane - fdiv(plane, cdiv) * C;
        const float4 *t = reinterpret_cast<const float4 *>(tbl + static_cast<size_t>(c) * kTbl);
        dx[i] = bn_bwd1<RELU>(dy[i], x[i], t[0], t[1]);
    }
}


// ---- float4 streaming passes for planes that are not a multiple of 4 floats (7x7 maps): the channel is
// ---- resolved per element (a float4 can straddle two planes); needs only total % 4 == 0.
template <bool RELU>
__global__ __launch_bounds__(kThreads) void k_affine_fwd_v4g(
    const float4 *__restrict__ x, const float *__restrict__ gamma, const float *__restrict__ beta,
    float4 *__restrict__ y, unsigned n4, FastDiv pdiv, FastDiv cdiv, unsigned C, int with_sign, SignArgs sa) {
    __shared__ double red[12];
    unsigned nblk = gridDim.x;
    if (with_sign) {
        nblk -= 1;
        if (blockIdx.x == nblk) {
            sign_loss_block(gamma, sa.b, sa.alpha, sa.margin, sa.l2, static_cast<int>(C), sa.loss, sa.acc,
                            sa.bits, red);
            return;
        }
    }
    const unsigned step = nblk * kThreads;
    for (unsigned q = blockIdx.x * kThreads + threadIdx.x; q < n4; q += step) {
        const float4 v = x[q];
        float in[4] = {v.x, v.y, v.z, v.w}, out[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned plane = fdiv(4 * q + i, pdiv);
            const unsigned c = plane - fdiv(plane, cdiv) * C;
            out[i] = affine1<RELU>(in[i], gamma[c], beta[c]);
        }
        y[q] = make_float4(out[0], out[1], out[2], out[3]);
    }
}

template <bool RELU>
__global__ __launch_bounds__(kThreads) void k_bn_affine_fwd_v4g(
    const float4 *__restrict__ x, const float *__restrict__ tbl, float4 *__restrict__ y, unsigned n4, FastDiv pdiv,
    FastDiv cdiv, unsigned C, int with_sign, SignArgs sa, const float *__restrict__ gamma) {
    __shared__ double red[12];
    unsigned nblk = gridDim.x;
    if (with_sign) {
        nblk -= 1;
        if (blockIdx.x == nblk) {
            sign_loss_block(gamma, sa.b, sa.alpha, sa.margin, sa.l2, static_cast<int>(C), sa.loss, sa.acc,
                            sa.bits, red);
            return;
        }
    }
    const unsigned step = nblk * kThreads;
    for (unsigned q = blockIdx.x * kThreads + threadIdx.x; q < n4; q += step) {
        const float4 v = x[q];
        float in[4] = {v.x, v.y, v.z, v.w}, out[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned plane = fdiv(4 * q + i, pdiv);
            const unsigned c = plane - fdiv(plane, cdiv) * C;
            out[i] = bn_affine1<RELU>(in[i], *reinterpret_cast<const float4 *>(tbl + static_cast<size_t>(c) * kTbl));
        }
        y[q] = make_float4(out[0], out[1], out[2], out[3]);
    }
}

template <bool RELU>
__global__ __launch_bounds__(kThreads) void k_bn_affine_bwd_v4g(
    const float4 *__restrict__ dy, const float4 *__restrict__ x, const float *__restrict__ tbl,
    float4 *__restrict__ dx, unsigned n4, FastDiv pdiv, FastDiv cdiv, unsigned C) {
    const unsigned step = gridDim.x * kThreads;
    for (unsigned q = blockIdx.x * kThreads + threadIdx.x; q < n4; q += step) {
        const float4 d = dy[q], v = x[q];
        float din[4] = {d.x, d.y, d.z, d.w}, xin[4] = {v.x, v.y, v.z, v.w}, out[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned plane = fdiv(4 * q + i, pdiv);
            const unsigned c = plane - fdiv(plane, cdiv) * C;
            const float4 *t = reinterpret_cast<const float4 *>(tbl + static_cast<size_t>(c) * kTbl);
            out[i] = bn_bwd1<RELU>(din[i], xin[i], t[0], t[1]);
        }
        dx[q] = make_float4(out[0], out[1], out[2], out[3]);
    }
}

// Finish of the fused backward: as k_passport_bwd_finish plus the table for the apply pass.
// W-less form (dW == nullptr, s == nullptr) serves the public branch with learnable gamma/beta.
struct BnBwdFinishArgs {
    const float *tbl_in;       // forward table (mean, invstd, gamma, beta)
    float *tbl_out;            // + c2, c3
    double inv_m;              // 1/(N*HW); 0 in evaluation mode (statistics are constants: no mean terms)
};

template <bool VEC, int RPW>
__global__ __launch_bounds__(kThreads) void k_passport_bn_bwd_finish(
    const double *__restrict__ part, int NS, int C, const float *__restrict__ b, float alpha, float margin,
    float l2, const float *__restrict__ dloss, const float *__restrict__ dgamma_extra,
    const float *__restrict__ dbeta_extra, const double *__restrict__ s, int K, float *__restrict__ dgamma,
    float *__restrict__ dbeta, float *__restrict__ dW, BnBwdFinishArgs f) {
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
        const float4 t0 = *reinterpret_cast<const float4 *>(f.tbl_in + static_cast<size_t>(co) * kTbl);
        float dg = static_cast<float>(ag), db = static_cast<float>(ab);
        if (threadIdx.x == 0 && co0 + r < C) {
            float4 *t = reinterpret_cast<float4 *>(f.tbl_out + static_cast<size_t>(co) * kTbl);
            t[0] = t0;
            t[1] = make_float4(static_cast<float>(ab * f.inv_m), static_cast<float>(ag * f.inv_m), 0.0f, 0.0f);
        }
        if (dgamma_extra) dg += dgamma_extra[co];
        if (dbeta_extra) db += dbeta_extra[co];
        if (dloss) dg += dloss[0] * sign_loss_grad1(t0.z, b[co], alpha, margin, l2);
        dgv[r] = dg;
        dbv[r] = db;
        if (threadIdx.x == 0 && co0 + r < C) {
            dgamma[co] = dg;
            dbeta[co] = db;
        }
    }
    if (dW) write_dw_rows<VEC, RPW>(dW, s, C, K, co0, dgv, dbv);
}


// ============================================================================================
// Register-resident single-pass form of the BatchNorm-fused layer.
//
// 256 CUs hold 128 MB of vector registers -- more than any activation of the CIFAR-shape nets (33.5 MB at most).
// When a layer's x (backward: dy and x) fits, every workgroup loads its slice ONCE into registers, the channel
// sums are formed, and y (dx) is computed from the registers: 8 B/element forward and 12 B/element backward
// instead of 12 and 20, and one launch instead of three.
//   * workgroup = (channel group cb, batch slice s): G adjacent channels (G > 1 only for planes shorter than a
//     128-byte line) x `nps` samples; thread t keeps float4 units t, t+T, ... (F4 of them, a template constant);
//   * S == 1: the workgroup owns its channels outright, nothing leaves the CU between the two phases;
//   * S > 1 (fewer channels than CUs): the S workgroups of a channel exchange their two partial sums in-launch as
//     data-tagged granules (res_exchange below), summed in slice order: every partner gets bit-identical
//     statistics.  All S*CB workgroups must be co-resident: the host only takes this path with T = 1024 and a
//     grid <= the CU count, on a device it does not share with a concurrent kernel (the caller withholds `sync`
//     otherwise); the wait is bounded, an expired one poisons the statistics with NaN and raises
//     sync[kSyncTimeoutWord].
// Fixed-order sums throughout: bit-reproducible run to run.
// ============================================================================================
// Exchange buffer (`sync`, DEEPIPR_SYNC_WORDS 32-bit words = granules of 8 bytes): for every slice count S in
// {2, 4, 8, 16} a region of kXchChannels x S slots of 4 granules, for S in {32, 64} (maps too large for one pass, see
// plan_resident) a region of 2048 granules each (two sets of 256 / S channel slots: the ranges kernels alternate between
// them, ABI v11); then the time-out word.  A slot is addressed by the channel's index WITHIN the launch (a launch splits at
// most 256 / S channels), so the channel ranges of one layer reuse the slots.
constexpr int kXchChannels = 256;                  // channels are split only when C < CUs, i.e. C <= 255
constexpr int kXchMaxSlices = 64;                  // 4 granules per slice: up to 256 granules, four per lane of wave 0
constexpr int kXchPow2Granules = kXchChannels * (2 + 4 + 8 + 16) * 4 + 2 * 2048;     // the regions of S = 2 .. 16 (+ two retired ones)
// ... and ONE REGION PER SLICE COUNT 2 .. 64 for the layers that run as channel ranges (plan_resident: any slice count there, two
// slot sets of 256 / S channels each = at most 2 048 granules): a slice derives the exchange's tag from its OWN slot's history, so
// launches with different S must never share slots (the first round-6 attempt let S = 18 use the region of 32: the partners
// computed different tags and every wait expired)
constexpr int kXchAnyRegion = 2048;
constexpr int kXchGranules = kXchPow2Granules + 63 * kXchAnyRegion;
constexpr int kSyncTimeoutWord = 2 * kXchGranules;
static_assert(kSyncTimeoutWord == DEEPIPR_SYNC_TIMEOUT_WORD && kSyncTimeoutWord + 16 == DEEPIPR_SYNC_WORDS, "header out of step");
constexpr unsigned kSpinLimit = 1u << 22;          // x s_sleep(2) + one poll: a few seconds

struct ResPlan {
    int T, F4;            // threads per workgroup, float4 units per thread
    int S, nps;           // batch slices per channel group, samples per slice
    int G, q4, gq;        // channels per workgroup, float4 per plane, G*q4
    int blocks;           // workgroups of this launch: (channels of the pass / G) * S
    int c_off;            // first channel of this launch (channel-range passes of a map too large for one pass)
    int cpp, passes;      // host side: channels per pass, number of passes (1: the whole layer in one launch)
    int stagger;          // ranges kernels: odd channel groups start this many s_sleep(127) late (phases of neighbours interleave)
    int xoff;             // granule offset of this launch's exchange region (xch_region)
    FastDiv gqdiv;
#ifdef DEEPIPR_TEST_HOOKS
    // Measurement / test build only (`make trace`: libdeepipr_hip_trace.so; the production library has none of this):
    unsigned spin;        // bound of the exchange wait (kSpinLimit; tests shorten it)
    int drop;             // this slice never publishes its partial sums (-1 = none): forces the time-out path
    int xcd_map;          // slices of one channel are placed on workgroups with equal blockIdx % 8 (one XCD)
    unsigned long long *trace;   // [block][8] wall-clock stamps of the kernel's phases
#endif
};
#ifdef DEEPIPR_TEST_HOOKS
#define DEEPIPR_RES_SPIN(pl) ((pl).spin)
#define DEEPIPR_RES_DROP(pl) ((pl).drop)
#define DEEPIPR_RES_XCD(pl) ((pl).xcd_map)
#else
#define DEEPIPR_RES_SPIN(pl) (kSpinLimit)
#define DEEPIPR_RES_DROP(pl) (-1)
#define DEEPIPR_RES_XCD(pl) (0)
#endif

// Phase stamps of one workgroup (thread 0): 0 entry, 1 loads consumed + block sums done, 2 exchange done,
// 3 channel table ready (forward), 4 all stores issued.  Compiled in only with -DDEEPIPR_TRACE (`make trace` builds
// libdeepipr_hip_trace.so for tools/res_trace.py): the stamps cost registers, and k_bn_res_bwd<1024, 8> sits at the
// 128-VGPR limit of four waves per SIMD -- with them it spills 100 B per lane to scratch (measured as +75 % HBM
// traffic by the PMC passes of round 2).
__device__ __forceinline__ void res_stamp(const ResPlan &pl, int slot) {
#ifdef DEEPIPR_TRACE
    if (pl.trace && threadIdx.x == 0)
        pl.trace[static_cast<size_t>(blockIdx.x) * 8 + slot] = wall_clock64();
#endif
}

// Per-local-channel sums of (a, b) over the workgroup; every thread gets the sums of ITS channel.
// red: 2 * (T/64) * 8 doubles of LDS.
template <int T>
__device__ __forceinline__ void res_block_sums(double &a, double &b, const ResPlan &pl, int c_local, double *red) {
    constexpr int NW = T / kWave;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (pl.G == 1) {
        a = wave_sum(a);
        b = wave_sum(b);
    } else {                                   // lanes with equal (lane % gq) / q4 share a channel
        for (int off = kWave / 2; off >= pl.gq; off >>= 1) {
            a += __shfl_xor(a, off, kWave);
            b += __shfl_xor(b, off, kWave);
        }
        for (int off = pl.q4 >> 1; off > 0; off >>= 1) {
            a += __shfl_xor(a, off, kWave);
            b += __shfl_xor(b, off, kWave);
        }
    }
    __syncthreads();
    if (lane < pl.gq && lane == c_local * pl.q4) {          // first lane of each local channel (G == 1: lane 0)
        red[(wave * 8 + c_local) * 2 + 0] = a;
        red[(wave * 8 + c_local) * 2 + 1] = b;
    }
    __syncthreads();
    a = 0.0;
    b = 0.0;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        a += red[(w * 8 + c_local) * 2 + 0];
        b += red[(w * 8 + c_local) * 2 + 1];
    }
}

// The same for four values at once (the dual kernels: two layers' sums): one barrier pair.  Two calls of the two-value
// form in a row made the compiler keep the first call's 2 * T/64 LDS reads in flight across the second (+58 VGPRs on
// the 1024-thread instances).  red: 4 * (T/64) * 8 doubles of LDS.  Per value the summation order is the two-value
// form's (waves in index order), so the results are bit-identical to it.
template <int T>
__device__ __forceinline__ void res_block_sums4(double &a, double &b, double &c, double &d, const ResPlan &pl,
                                                int c_local, double *red) {
    constexpr int NW = T / kWave;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (pl.G == 1) {
        a = wave_sum(a);
        b = wave_sum(b);
        c = wave_sum(c);
        d = wave_sum(d);
    } else {
        for (int off = kWave / 2; off >= pl.gq; off >>= 1) {
            a += __shfl_xor(a, off, kWave);
            b += __shfl_xor(b, off, kWave);
            c += __shfl_xor(c, off, kWave);
            d += __shfl_xor(d, off, kWave);
        }
        for (int off = pl.q4 >> 1; off > 0; off >>= 1) {
            a += __shfl_xor(a, off, kWave);
            b += __shfl_xor(b, off, kWave);
            c += __shfl_xor(c, off, kWave);
            d += __shfl_xor(d, off, kWave);
        }
    }
    __syncthreads();
    if (lane < pl.gq && lane == c_local * pl.q4) {
        double *row = red + (wave * 8 + c_local) * 4;
        row[0] = a;
        row[1] = b;
        row[2] = c;
        row[3] = d;
    }
    __syncthreads();
    a = b = c = d = 0.0;
#pragma unroll 2
    for (int w = 0; w < NW; ++w) {
        const double *row = red + (w * 8 + c_local) * 4;
        a += row[0];
        b += row[1];
        c += row[2];
        d += row[3];
    }
}

// In-launch exchange of one channel's two partial sums (doubles) between its S slice workgroups, run by WAVE 0.
//
// Transport: data-tagged granules (MI355X_MICROARCH.md, hand-off price list: "handoff-1to1", the cheapest valid
// form -- one naturally aligned 8-byte {payload, tag} written by ONE sc1 store needs no ordering at all, neither a
// drained flag nor an agent fence).  A slice publishes its two doubles as four granules {32 payload bits, tag};
// lane i of wave 0 then polls granule i of the channel (4*S <= 64 lanes) with sc1 loads until its tag is the
// call's tag, and every lane sums the S partials in slice order from wave shuffles: all partners get bit-identical
// statistics.  The first version of this kernel (round 1) used sc1 payload stores -> vmcnt(0) drain -> returning
// ticket atomic -> poll -> payload loads: four dependent round trips, measured at 4.0-4.8 us of a 17.9 us launch
// (profiles/r02_res_trace_*.log); this form has one store and (usually) one or two polls.
//
// Tag: a per-slot call counter kept in the granule itself.  A slice reads its OWN slot's old tag when the kernel
// starts and publishes old + 1; the slots of one channel in one S-region are always used together, so all partners
// compute the same tag without any host-side state (launch arguments are frozen in a replayed hipGraph) and without
// ever resetting the buffer.  A slice that never arrives (test hook `drop`) or an expired wait leaves the tags out
// of step for good: every later call on that channel times out too, until the host re-zeroes the buffer.
// An expired wait poisons the statistics with NaN and raises sync[kSyncTimeoutWord].
struct ResXch {
    unsigned long long *gran;     // this channel's S * 4 granules
    unsigned expect;              // tag of this call
};

__device__ __forceinline__ unsigned long long xch_load(const unsigned long long *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // global_load ... sc1: bypasses L1
}

// granule offset of the exchange region of a launch (host side: ResPlan::xoff).  Power-of-two S up to 16 of the one-launch
// forms: a region of 256 channel slots; the channel-range form (any S): the region of exactly this S.
inline int xch_region(int S, bool ranges) {
    if (ranges) return kXchPow2Granules + (S - 2) * kXchAnyRegion;
    return kXchChannels * (S - 2) * 4;                                      // 2 + 4 + ... + S/2 = S - 2   (S = 2, 4, 8, 16)
}

// Called by every thread at kernel entry (only wave 0 needs it; one L2 round trip hidden behind the bulk loads).
__device__ __forceinline__ ResXch res_xch_begin(unsigned *sync, int xoff, int c, int s, int S) {
    ResXch x;
    x.gran = reinterpret_cast<unsigned long long *>(sync) + xoff + static_cast<size_t>(c) * S * 4;
    x.expect = 0;
    if (threadIdx.x < kWave) x.expect = static_cast<unsigned>(xch_load(x.gran + s * 4) >> 32) + 1u;
    return x;
}

__device__ __forceinline__ void res_exchange(double &s0, double &s1, const ResXch &x, int s, int S, unsigned *sync,
                                             unsigned spin_limit, int drop) {
    const int lane = threadIdx.x;                      // wave 0
    const unsigned long long b0 = static_cast<unsigned long long>(__double_as_longlong(s0));
    const unsigned long long b1 = static_cast<unsigned long long>(__double_as_longlong(s1));
    if (lane < 4 && s != drop) {
        const unsigned long long src = lane < 2 ? b0 : b1;
        const unsigned long long half = (lane & 1) ? (src >> 32) : (src & 0xffffffffull);
        __hip_atomic_store(x.gran + s * 4 + lane, (static_cast<unsigned long long>(x.expect) << 32) | half,
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);             // ONE 8-byte sc1 store
    }
    // lane i polls granules i, i + 64, i + 128, i + 192 (S <= 16: the first one only)
    constexpr int kRounds = kXchMaxSlices * 4 / kWave;
    unsigned long long v[kRounds];
    bool ok[kRounds];
#pragma unroll
    for (int r = 0; r < kRounds; ++r) {
        v[r] = 0;
        ok[r] = lane + r * kWave >= 4 * S;
    }
    bool expired = false;
    unsigned spins = 0;
    while (true) {
        bool all = true;
#pragma unroll
        for (int r = 0; r < kRounds; ++r) {
            if (!ok[r]) {
                v[r] = xch_load(x.gran + lane + r * kWave);
                ok[r] = static_cast<unsigned>(v[r] >> 32) == x.expect;
            }
            all = all && ok[r];
        }
        if (__all(all)) break;
        __builtin_amdgcn_s_sleep(2);
        if (++spins > spin_limit) {
            expired = true;
            break;
        }
    }
    double t0 = 0.0, t1 = 0.0;
    for (int sp = 0; sp < S; ++sp) {                   // slice order: identical rounding in every partner
        const int r = sp >> 4;                          // granule 4 * sp + j sits in round (4 * sp + j) / 64 = sp / 16
        const unsigned pay = static_cast<unsigned>(r == 0 ? v[0] : r == 1 ? v[1] : r == 2 ? v[2] : v[3]);
        const int g = (4 * sp) & (kWave - 1);
        const unsigned long long lo0 = __shfl(pay, g, kWave), hi0 = __shfl(pay, g + 1, kWave);
        const unsigned long long lo1 = __shfl(pay, g + 2, kWave), hi1 = __shfl(pay, g + 3, kWave);
        t0 += __longlong_as_double(static_cast<long long>((hi0 << 32) | lo0));
        t1 += __longlong_as_double(static_cast<long long>((hi1 << 32) | lo1));
    }
    if (expired) {
        if (lane == 0) __hip_atomic_store(sync + kSyncTimeoutWord, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        t0 = t1 = __longlong_as_double(0x7ff8000000000000LL);
    }
    s0 = t0;
    s1 = t1;
}

// blockIdx -> (channel group cb, slice s).  XCD-aware: workgroup b runs on XCD b % 8 (observed dispatch order, used
// for speed only), so the S slices of a channel are given indices with equal b % 8: their granules stay in one L2.
__device__ __forceinline__ void res_block_coords(const ResPlan &pl, int &cb, int &s) {
    const int b = blockIdx.x;
    if (DEEPIPR_RES_XCD(pl)) {
        const int span = 8 * pl.S, g = b / span, r = b - g * span;
        s = r >> 3;
        cb = g * 8 + (r & 7);
    } else {
        cb = b / pl.S;
        s = b - cb * pl.S;
    }
}

template <int T>
__device__ __forceinline__ void sign_loss_block_t(const float *__restrict__ gamma, const SignArgs &sa, int C,
                                                  double *red /* >= 3 * T/64 doubles */) {
    constexpr int NW = T / kWave;
    double v[3] = {0.0, 0.0, 0.0};
    for (int c = threadIdx.x; c < C; c += T) {
        const float g = gamma[c], bb = sa.b[c];
        const float z = __fadd_rn(__fmul_rn(-bb, g), sa.margin);
        v[0] += static_cast<double>(__fmul_rn(sa.alpha, fmaxf(z, 0.0f)));
        v[1] += static_cast<double>(__fmul_rn(g, g));
        const int sg = (g > 0.0f) - (g < 0.0f);
        const int sb = (bb > 0.0f) - (bb < 0.0f);
        v[2] += (sg == sb) ? 1.0 : 0.0;
        if (sa.bits) sa.bits[c] = static_cast<int8_t>(sg);
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        v[i] = wave_sum(v[i]);
        if ((threadIdx.x & 63) == 0) red[i * NW + (threadIdx.x >> 6)] = v[i];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double t[3] = {0.0, 0.0, 0.0};
        for (int i = 0; i < 3; ++i)
            for (int w = 0; w < NW; ++w) t[i] += red[i * NW + w];
        if (sa.loss) *sa.loss = static_cast<float>(t[0] + static_cast<double>(sa.l2) * t[1]);
        if (sa.acc) *sa.acc = static_cast<float>(t[2] / static_cast<double>(C));
    }
}

// All workgroups of a ranges launch are symmetric, so without help they stay in phase for the whole launch: the chip reads, then
// exchanges (HBM idle), then writes.  The workgroups of every other channel start late by about half a range: their loads fall
// into their neighbours' exchange and store phases for the rest of the launch.  (All S partners of a channel wait together.)
__device__ __forceinline__ void res_stagger(const ResPlan &pl, int cb) {
    if (cb & 1)
        for (int i = 0; i < pl.stagger; ++i) __builtin_amdgcn_s_sleep(127);
}

// One channel range of the forward kernel: workgroup (cb, s) takes channel(s) c_off + cb * G of its batch slice.  `xc`: the
// exchange slots of this range (S > 1); first: the range that counts the call (num_batches_tracked).
template <int T, int F4, bool LOOP = false>
__device__ __forceinline__ void bn_res_fwd_range(
    const float4 *__restrict__ x, float4 *__restrict__ y, const float *__restrict__ gamma,
    const float *__restrict__ beta, int relu, int N, int C, const ResPlan &pl, const BnFinishArgs &f,
    unsigned *sync, const float4 *__restrict__ residual, double *red, double *xch, int cb, int s, int c_off,
    const ResXch &xc, bool first) {
    int t = threadIdx.x;
    int n0 = s * pl.nps;
    // inside k_bn_res_fwd_ranges' loop: nothing derived from the thread index or the slice may be hoisted out of it (the
    // unit -> address arithmetic is range-invariant: hipcc kept 2 * F4 values of it alive across the loop and spilled)
    if (LOOP) asm volatile("" : "+v"(t), "+s"(n0));
    const int c0 = c_off + cb * pl.G;
    const int units = max(0, min(N, n0 + pl.nps) - n0) * pl.gq;
    const int c_local = (pl.G == 1) ? 0 : (t % pl.gq) / pl.q4;        // T % gq == 0: the same for all of t's units
    // shift of the statistics (see BnFinishArgs): the channel's first element, identical in all S slices
    res_stamp(pl, 0);
    // what the channel table needs besides the sums, loaded up front
    const int c_mine = c0 + c_local;
    const float g_pre = gamma[c_mine], b_pre = beta[c_mine];
    const bool owner = s == 0 && t < pl.gq && t == c_local * pl.q4;           // one thread per channel
    float rm_old = 0.0f, rv_old = 0.0f;
    if (owner && f.running_mean) {
        rm_old = f.running_mean[c_mine];
        rv_old = f.running_var[c_mine];
    }
    const float K = reinterpret_cast<const float *>(x)[static_cast<size_t>(c0 + c_local) * pl.q4 * 4];
    // float4 index of unit j of this workgroup's slice; kept in registers for the write phase only while that is
    // affordable (F4 <= 8): the F4 = 12 / 16 instances sit at the 128-VGPR limit and recompute it instead
    constexpr bool kKeepIdx = F4 <= (LOOP ? 6 : 8);           // (the loop over ranges costs the 8-unit instance its last registers)
    auto unit_index = [&](int j) {
        const unsigned row = fdiv(static_cast<unsigned>(j), pl.gqdiv);
        return static_cast<unsigned>((static_cast<size_t>(n0 + row) * C + c0) * pl.q4 + (j - row * pl.gq));
    };
    // the shortcut of a folded residual tail is fetched in the SAME load phase (while registers allow): read after
    // the statistics it would be a second, latency-bound load phase in front of the stores
    constexpr bool kPreRes = F4 <= (LOOP ? 6 : 8);
    float4 v[F4];
    float4 rs[kPreRes ? F4 : 1];
    unsigned idx[kKeepIdx ? F4 : 1];
#pragma unroll
    for (int k = 0; k < F4; ++k) {
        const int j = t + k * T;
        v[k] = make_float4(K, K, K, K);                                // unused units contribute (K-K) = 0
        if (kKeepIdx) idx[k] = 0;
        if (kPreRes) rs[k] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (j < units) {
            const unsigned i = unit_index(j);
            if (kKeepIdx) idx[k] = i;
            v[k] = x[i];
            if (kPreRes && residual) rs[k] = residual[i];
        }
    }
    float a0 = 0.0f, a1 = 0.0f;
#pragma unroll
    for (int k = 0; k < F4; ++k) {
        const float dx0 = v[k].x - K, dx1 = v[k].y - K, dx2 = v[k].z - K, dx3 = v[k].w - K;
        a0 += (dx0 + dx1) + (dx2 + dx3);
        a1 = fmaf(dx0, dx0, a1);
        a1 = fmaf(dx1, dx1, a1);
        a1 = fmaf(dx2, dx2, a1);
        a1 = fmaf(dx3, dx3, a1);
    }
    double s1 = static_cast<double>(a0), s2 = static_cast<double>(a1);
    res_block_sums<T>(s1, s2, pl, c_local, red);
    res_stamp(pl, 1);
    if (pl.S > 1) {                                   // G == 1 here
        if (t < kWave) {
            res_exchange(s1, s2, xc, s, pl.S, sync, DEEPIPR_RES_SPIN(pl), DEEPIPR_RES_DROP(pl));
            if (t == 0) {
                xch[0] = s1;
                xch[1] = s2;
            }
        }
        __syncthreads();
        s1 = xch[0];
        s2 = xch[1];
    }
    res_stamp(pl, 2);
    // Every thread holds its channel's sums: each forms mean / invstd itself (f64, identical arithmetic in all of
    // them) from values preloaded at kernel entry -- no LDS broadcast, no barrier and no global-load latency between
    // the statistics and the write phase; the channel's owner thread also updates the running statistics and the
    // table (stores nobody waits for).
    const double dmu = s1 * f.inv_m;                                   // mean - K
    double var = s2 * f.inv_m - dmu * dmu;
    if (var < 0.0) var = 0.0;
    const float mean = static_cast<float>(static_cast<double>(K) + dmu);
    const float invstd = static_cast<float>(1.0 / sqrt(var + static_cast<double>(f.eps)));
    const float4 ch = make_float4(mean, invstd, g_pre, b_pre);
    if (owner) {
        if (f.running_mean) {
            f.running_mean[c_mine] = (1.0f - f.momentum) * rm_old + f.momentum * mean;
            f.running_var[c_mine] = (1.0f - f.momentum) * rv_old + f.momentum * static_cast<float>(var * f.unbias);
        }
        float4 *row = reinterpret_cast<float4 *>(f.tbl + static_cast<size_t>(c_mine) * kTbl);
        row[0] = ch;
        row[1] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
    if (first && blockIdx.x == 0 && t == 0 && f.num_batches_tracked) *f.num_batches_tracked += 1;
    res_stamp(pl, 3);
#pragma unroll
    for (int k = 0; k < F4; ++k) {
        const int j = t + k * T;
        if (j < units) {
            float4 o;
            if (relu) {
                o = make_float4(bn_affine1<true>(v[k].x, ch), bn_affine1<true>(v[k].y, ch),
                                bn_affine1<true>(v[k].z, ch), bn_affine1<true>(v[k].w, ch));
            } else {
                o = make_float4(bn_affine1<false>(v[k].x, ch), bn_affine1<false>(v[k].y, ch),
                                bn_affine1<false>(v[k].z, ch), bn_affine1<false>(v[k].w, ch));
            }
            const unsigned i = kKeepIdx ? idx[k] : unit_index(j);
            if (residual) {                          // the block's tail: relu(layer output + shortcut)
                const float4 r = kPreRes ? rs[k] : residual[i];
                o = make_float4(relu1(o.x + r.x), relu1(o.y + r.y), relu1(o.z + r.z), relu1(o.w + r.w));
            }
            y[i] = o;
        }
    }
    res_stamp(pl, 4);
}

template <int T, int F4>
__global__ __launch_bounds__(T) void k_bn_res_fwd(
    const float4 *__restrict__ x, float4 *__restrict__ y, const float *__restrict__ gamma,
    const float *__restrict__ beta, int relu, int N, int C, ResPlan pl, BnFinishArgs f, double *part,
    unsigned *sync, int with_sign, SignArgs sa, const float4 *__restrict__ residual) {
    constexpr int NW = T / kWave;
    __shared__ double red[2 * NW * 8];
    __shared__ double xch[2];
    if (with_sign && static_cast<int>(blockIdx.x) == pl.blocks) {
        sign_loss_block_t<T>(gamma, sa, C, red);
        return;
    }
    int cb, s;
    res_block_coords(pl, cb, s);
    ResXch xc{};
    if (pl.S > 1) xc = res_xch_begin(sync, pl.xoff, cb, s, pl.S);              // slot = the channel's index within this launch
    bn_res_fwd_range<T, F4>(x, y, gamma, beta, relu, N, C, pl, f, sync, residual, red, xch, cb, s, pl.c_off, xc, true);
}

// Maps beyond the register file (ImageNet geometry): ALL channel ranges of the layer in ONE launch.  Round 5 launched the
// kernel above once per range (ResNet50 at batch 256: 189 forward + 390 backward launches of ~30 us, each with its own ramp
// and a read phase that cannot start before the previous launch's write phase has drained: 0.40 - 0.46 of the HBM peak).
// Here workgroup (cb, s) walks the ranges cb, cb + cpp, cb + 2 cpp, ... itself; the workgroups drift apart, so one's stores
// overlap another's loads.  The exchange alternates between two slot sets (channel indices cb and cb + cpp of the S-region: a
// region has 256 slots, a range uses 256 / S): a workgroup may publish range r + 2 into the set of range r only after it has
// finished range r + 1, which needs every partner's r + 1 granule, which a partner writes only after it has READ range r.
template <int T, int F4>
__global__ __launch_bounds__(T) void k_bn_res_fwd_ranges(
    const float4 *__restrict__ x, float4 *__restrict__ y, const float *__restrict__ gamma,
    const float *__restrict__ beta, int relu, int N, int C, ResPlan pl, BnFinishArgs f, unsigned *sync,
    const float4 *__restrict__ residual) {
    constexpr int NW = T / kWave;
    __shared__ double red[2 * NW * 8];
    __shared__ double xch[2];
    int cb, s;
    res_block_coords(pl, cb, s);
    res_stagger(pl, cb);
    for (int r = 0; r < pl.passes; ++r) {
        const int c_off = r * pl.cpp;
        if (cb * pl.G >= C - c_off) break;                             // the last range may be short (whole workgroups leave)
        // this range's slot set; its tag is read afresh (this workgroup's own granule of range r - 2: old + 1 again)
        const ResXch xc = res_xch_begin(sync, pl.xoff, cb + (r & 1) * pl.cpp, s, pl.S);
        bn_res_fwd_range<T, F4, true>(x, y, gamma, beta, relu, N, C, pl, f, sync, residual, red, xch, cb, s, c_off, xc, r == 0);
    }
}

struct ResBwdArgs {
    const float *b;                 // signature bits (sign loss), may be nullptr
    float alpha, margin, l2;
    const float *dloss, *dgamma_extra, *dbeta_extra;
    float *dgamma, *dbeta;
    double inv_m;                   // 1/(N*HW); 0 in evaluation mode
    // fused residual tail (all nullptr when the layer has none): the upstream gradient is (dy + dy2) masked by
    // tail_out > 0; that masked gradient is also the shortcut's gradient and is written to dres
    const float4 *dy2, *tail_out;
    float4 *dres;
};

__device__ __forceinline__ void res_bwd_prep(float d, float xv, const float4 &ch, int relu, float &dz, float &xh) {
    xh = (xv - ch.x) * ch.y;
    dz = d;
    if (relu) dz = (__fadd_rn(__fmul_rn(ch.z, xh), ch.w) > 0.0f) ? d : 0.0f;
}

template <int T, int F4, bool LOOP = false>
__device__ __forceinline__ void bn_res_bwd_range(
    const float4 *__restrict__ dy, const float4 *__restrict__ x, const float *__restrict__ tbl,
    float4 *__restrict__ dx, int relu, int N, int C, const ResPlan &pl, unsigned *sync, const ResBwdArgs &a,
    double *red, double *xch, int cb, int s, int c_off, const ResXch &xc) {
    int t = threadIdx.x;
    int n0 = s * pl.nps;
    if (LOOP) asm volatile("" : "+v"(t), "+s"(n0));           // (bn_res_fwd_range)
    const int c0 = c_off + cb * pl.G;
    const int units = max(0, min(N, n0 + pl.nps) - n0) * pl.gq;
    const int c_local = (pl.G == 1) ? 0 : (t % pl.gq) / pl.q4;
    res_stamp(pl, 0);
    const float4 ch = *reinterpret_cast<const float4 *>(tbl + static_cast<size_t>(c0 + c_local) * kTbl);
    float4 dz[F4], xh[F4];
    unsigned idx[F4];
#pragma unroll
    for (int k = 0; k < F4; ++k) {
        const int j = t + k * T;
        dz[k] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        xh[k] = make_float4(ch.x, ch.x, ch.x, ch.x);          // -> xhat 0 for the unused units
        idx[k] = 0;
        if (j < units) {
            const unsigned row = fdiv(static_cast<unsigned>(j), pl.gqdiv);
            idx[k] = static_cast<unsigned>((static_cast<size_t>(n0 + row) * C + c0) * pl.q4 + (j - row * pl.gq));
            dz[k] = dy[idx[k]];
            xh[k] = x[idx[k]];
            if (a.tail_out) {
                const float4 o = a.tail_out[idx[k]];
                float4 d = dz[k];
                if (a.dy2) {
                    const float4 e = a.dy2[idx[k]];
                    d = make_float4(d.x + e.x, d.y + e.y, d.z + e.z, d.w + e.w);
                }
                d = make_float4(o.x > 0.0f ? d.x : 0.0f, o.y > 0.0f ? d.y : 0.0f, o.z > 0.0f ? d.z : 0.0f,
                                o.w > 0.0f ? d.w : 0.0f);
                a.dres[idx[k]] = d;
                dz[k] = d;
            } else if (a.dy2) {                     // two consumers, no tail (the stem): their gradients summed here
                const float4 e = a.dy2[idx[k]];
                dz[k] = make_float4(dz[k].x + e.x, dz[k].y + e.y, dz[k].z + e.z, dz[k].w + e.w);
            }
        }
    }
    float a0 = 0.0f, a1 = 0.0f;
#pragma unroll
    for (int k = 0; k < F4; ++k) {
        float4 d = dz[k], v = xh[k];
        res_bwd_prep(d.x, v.x, ch, relu, dz[k].x, xh[k].x);
        res_bwd_prep(d.y, v.y, ch, relu, dz[k].y, xh[k].y);
        res_bwd_prep(d.z, v.z, ch, relu, dz[k].z, xh[k].z);
        res_bwd_prep(d.w, v.w, ch, relu, dz[k].w, xh[k].w);
        a0 = fmaf(dz[k].x, xh[k].x, a0);
        a0 = fmaf(dz[k].y, xh[k].y, a0);
        a0 = fmaf(dz[k].z, xh[k].z, a0);
        a0 = fmaf(dz[k].w, xh[k].w, a0);
        a1 += (dz[k].x + dz[k].y) + (dz[k].z + dz[k].w);
    }
    double ag = static_cast<double>(a0), ab = static_cast<double>(a1);
    res_block_sums<T>(ag, ab, pl, c_local, red);
    res_stamp(pl, 1);
    if (pl.S > 1) {
        if (t < kWave) {
            res_exchange(ag, ab, xc, s, pl.S, sync, DEEPIPR_RES_SPIN(pl), DEEPIPR_RES_DROP(pl));
            if (t == 0) {
                xch[0] = ag;
                xch[1] = ab;
            }
        }
        __syncthreads();
        ag = xch[0];
        ab = xch[1];
    }
    res_stamp(pl, 2);
    if (s == 0 && t < pl.gq && t == c_local * pl.q4) {
        const int c = c0 + c_local;
        float dg = static_cast<float>(ag), db = static_cast<float>(ab);
        if (a.dgamma_extra) dg += a.dgamma_extra[c];
        if (a.dbeta_extra) db += a.dbeta_extra[c];
        if (a.dloss) dg += a.dloss[0] * sign_loss_grad1(ch.z, a.b[c], a.alpha, a.margin, a.l2);
        a.dgamma[c] = dg;
        a.dbeta[c] = db;
    }
    const float c2 = static_cast<float>(ab * a.inv_m), c3 = static_cast<float>(ag * a.inv_m);
    const float sc = ch.z * ch.y;
#pragma unroll
    for (int k = 0; k < F4; ++k) {
        const int j = t + k * T;
        if (j < units) {
            dx[idx[k]] = make_float4(sc * (dz[k].x - c2 - xh[k].x * c3), sc * (dz[k].y - c2 - xh[k].y * c3),
                                     sc * (dz[k].z - c2 - xh[k].z * c3), sc * (dz[k].w - c2 - xh[k].w * c3));
        }
    }
    res_stamp(pl, 4);
}

template <int T, int F4>
__global__ __launch_bounds__(T) void k_bn_res_bwd(
    const float4 *__restrict__ dy, const float4 *__restrict__ x, const float *__restrict__ tbl,
    float4 *__restrict__ dx, int relu, int N, int C, ResPlan pl, double *part, unsigned *sync, ResBwdArgs a) {
    constexpr int NW = T / kWave;
    __shared__ double red[2 * NW * 8];
    __shared__ double xch[2];
    int cb, s;
    res_block_coords(pl, cb, s);
    ResXch xc{};
    if (pl.S > 1) xc = res_xch_begin(sync, pl.xoff, cb, s, pl.S);
    bn_res_bwd_range<T, F4>(dy, x, tbl, dx, relu, N, C, pl, sync, a, red, xch, cb, s, pl.c_off, xc);
}

// all channel ranges of a map beyond the register file in one launch (k_bn_res_fwd_ranges)
template <int T, int F4>
__global__ __launch_bounds__(T) void k_bn_res_bwd_ranges(
    const float4 *__restrict__ dy, const float4 *__restrict__ x, const float *__restrict__ tbl,
    float4 *__restrict__ dx, int relu, int N, int C, ResPlan pl, unsigned *sync, ResBwdArgs a) {
    constexpr int NW = T / kWave;
    __shared__ double red[2 * NW * 8];
    __shared__ double xch[2];
    int cb, s;
    res_block_coords(pl, cb, s);
    res_stagger(pl, cb);
    for (int r = 0; r < pl.passes; ++r) {
        const int c_off = r * pl.cpp;
        if (cb * pl.G >= C - c_off) break;
        const ResXch xc = res_xch_begin(sync, pl.xoff, cb + (r & 1) * pl.cpp, s, pl.S);
        bn_res_bwd_range<T, F4, true>(dy, x, tbl, dx, relu, N, C, pl, sync, a, red, xch, cb, s, c_off, xc);
    }
}

// ============================================================================================
// A projection block's LAST TWO norm layers and its tail in one launch per direction
// (models/resnet_passport.py:67-85: out = relu(bn_2(conv_2(h)) + bn_s(conv_s(x))) with a 1x1 stride-2 shortcut conv).
// The two norm layers are independent until the add, have the same shape and both own their channels the same way,
// so one workgroup keeps its slice of BOTH conv outputs in registers:
//   forward   read a, b; write out                     12 B/elt  (separate launches: 8 for the shortcut layer + 12
//                                                                 for the tail-folded layer = 20, and one launch more)
//   backward  read dy (+ dy2), out, a, b; write da, db  28 B/elt  (separate: 24 + 12 = 36): the masked gradient
//                                                                 d = (dy + dy2)[out > 0] never leaves the registers
// Same plan, same unit -> thread mapping, same reduction order and the same per-element arithmetic as the separate
// kernels: results are bit-identical to them (tests/test_norm_kernels_gpu.py:
// test_dual_tail_kernels_equal_the_two_separate_fused_layers).  W-less form only (learnable gamma / beta:
// plain ConvBlocks), batch statistics, no inner ReLU.  S > 1: the second layer's partial sums are exchanged through
// the slots of channel index cb + C (never used by a split launch of C channels: a region has 256 slots, a launch
// splits at most 256 / S of them).
// ============================================================================================
struct DualFwdArgs {
    const float4 *xb;
    const float *gamma_b, *beta_b;
    float momentum_b, eps_b;
    float *running_mean_b, *running_var_b;
    long long *num_batches_tracked_b;
    float *tbl_b;
};

template <int T, int F4>
__global__ __launch_bounds__(T) void k_bn_dual_fwd(
    const float4 *__restrict__ xa, float4 *__restrict__ y, const float *__restrict__ gamma_a,
    const float *__restrict__ beta_a, int relu_a, int relu_b, int N, int C, ResPlan pl, BnFinishArgs f,
    DualFwdArgs d, unsigned *sync) {
    constexpr int NW = T / kWave;
    __shared__ double red[4 * NW * 8];
    __shared__ double xch[4];
    const int t = threadIdx.x;
    int cb, s;
    res_block_coords(pl, cb, s);
    const int c0 = pl.c_off + cb * pl.G;
    const int n0 = s * pl.nps;
    const int units = max(0, min(N, n0 + pl.nps) - n0) * pl.gq;
    const int c_local = (pl.G == 1) ? 0 : (t % pl.gq) / pl.q4;
    ResXch xca{}, xcb{};
    if (pl.S > 1) {
        xca = res_xch_begin(sync, pl.xoff, cb, s, pl.S);
        xcb = res_xch_begin(sync, pl.xoff, cb + C, s, pl.S);
    }
    const int c_mine = c0 + c_local;
    const float ga = gamma_a[c_mine], ba = beta_a[c_mine], gb = d.gamma_b[c_mine], bb = d.beta_b[c_mine];
    const bool owner = s == 0 && t < pl.gq && t == c_local * pl.q4;
    float rma = 0.0f, rva = 0.0f, rmb = 0.0f, rvb = 0.0f;
    if (owner && f.running_mean) {
        rma = f.running_mean[c_mine];
        rva = f.running_var[c_mine];
    }
    if (owner && d.running_mean_b) {
        rmb = d.running_mean_b[c_mine];
        rvb = d.running_var_b[c_mine];
    }
    const size_t k_at = static_cast<size_t>(c_mine) * pl.q4 * 4;
    const float Ka = reinterpret_cast<const float *>(xa)[k_at], Kb = reinterpret_cast<const float *>(d.xb)[k_at];
    float4 va[F4], vb[F4];
    unsigned idx[F4];
#pragma unroll
    for (int k = 0; k < F4; ++k) {
        const int j = t + k * T;
        va[k] = make_float4(Ka, Ka, Ka, Ka);
        vb[k] = make_float4(Kb, Kb, Kb, Kb);
        idx[k] = 0;
        if (j < units) {
            const unsigned row = fdiv(static_cast<unsigned>(j), pl.gqdiv);
            idx[k] = static_cast<unsigned>((static_cast<size_t>(n0 + row) * C + c0) * pl.q4 + (j - row * pl.gq));
            va[k] = xa[idx[k]];
            vb[k] = d.xb[idx[k]];
        }
    }
    float a0 = 0.0f, a1 = 0.0f, b0 = 0.0f, b1 = 0.0f;
#pragma unroll
    for (int k = 0; k < F4; ++k) {
        const float p0 = va[k].x - Ka, p1 = va[k].y - Ka, p2 = va[k].z - Ka, p3 = va[k].w - Ka;
        a0 += (p0 + p1) + (p2 + p3);
        a1 = fmaf(p0, p0, a1);
        a1 = fmaf(p1, p1, a1);
        a1 = fmaf(p2, p2, a1);
        a1 = fmaf(p3, p3, a1);
        const float q0 = vb[k].x - Kb, q1 = vb[k].y - Kb, q2 = vb[k].z - Kb, q3 = vb[k].w - Kb;
        b0 += (q0 + q1) + (q2 + q3);
        b1 = fmaf(q0, q0, b1);
        b1 = fmaf(q1, q1, b1);
        b1 = fmaf(q2, q2, b1);
        b1 = fmaf(q3, q3, b1);
    }
    double sa1 = static_cast<double>(a0), sa2 = static_cast<double>(a1);
    double sb1 = static_cast<double>(b0), sb2 = static_cast<double>(b1);
    res_block_sums4<T>(sa1, sa2, sb1, sb2, pl, c_local, red);
    if (pl.S > 1) {                                   // G == 1 here
        if (t < kWave) {
            res_exchange(sa1, sa2, xca, s, pl.S, sync, DEEPIPR_RES_SPIN(pl), DEEPIPR_RES_DROP(pl));
            res_exchange(sb1, sb2, xcb, s, pl.S, sync, DEEPIPR_RES_SPIN(pl), DEEPIPR_RES_DROP(pl));
            if (t == 0) {
                xch[0] = sa1;
                xch[1] = sa2;
                xch[2] = sb1;
                xch[3] = sb2;
            }
        }
        __syncthreads();
        sa1 = xch[0];
        sa2 = xch[1];
        sb1 = xch[2];
        sb2 = xch[3];
    }
    const double dmua = sa1 * f.inv_m, dmub = sb1 * f.inv_m;
    double vara = sa2 * f.inv_m - dmua * dmua, varb = sb2 * f.inv_m - dmub * dmub;
    if (vara < 0.0) vara = 0.0;
    if (varb < 0.0) varb = 0.0;
    const float mean_a = static_cast<float>(static_cast<double>(Ka) + dmua);
    const float mean_b = static_cast<float>(static_cast<double>(Kb) + dmub);
    const float inv_a = static_cast<float>(1.0 / sqrt(vara + static_cast<double>(f.eps)));
    const float inv_b = static_cast<float>(1.0 / sqrt(varb + static_cast<double>(d.eps_b)));
    const float4 cha = make_float4(mean_a, inv_a, ga, ba), chb = make_float4(mean_b, inv_b, gb, bb);
    if (owner) {
        if (f.running_mean) {
            f.running_mean[c_mine] = (1.0f - f.momentum) * rma + f.momentum * mean_a;
            f.running_var[c_mine] = (1.0f - f.momentum) * rva + f.momentum * static_cast<float>(vara * f.unbias);
        }
        if (d.running_mean_b) {
            d.running_mean_b[c_mine] = (1.0f - d.momentum_b) * rmb + d.momentum_b * mean_b;
            d.running_var_b[c_mine] = (1.0f - d.momentum_b) * rvb + d.momentum_b * static_cast<float>(varb * f.unbias);
        }
        float4 *row = reinterpret_cast<float4 *>(f.tbl + static_cast<size_t>(c_mine) * kTbl);
        row[0] = cha;
        row[1] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        row = reinterpret_cast<float4 *>(d.tbl_b + static_cast<size_t>(c_mine) * kTbl);
        row[0] = chb;
        row[1] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
    if (blockIdx.x == 0 && t == 0) {
        if (f.num_batches_tracked) *f.num_batches_tracked += 1;
        if (d.num_batches_tracked_b) *d.num_batches_tracked_b += 1;
    }
#pragma unroll
    for (int k = 0; k < F4; ++k) {
        const int j = t + k * T;
        if (j < units) {
            float4 o, r;
            if (relu_a)
                o = make_float4(bn_affine1<true>(va[k].x, cha), bn_affine1<true>(va[k].y, cha),
                                bn_affine1<true>(va[k].z, cha), bn_affine1<true>(va[k].w, cha));
            else
                o = make_float4(bn_affine1<false>(va[k].x, cha), bn_affine1<false>(va[k].y, cha),
                                bn_affine1<false>(va[k].z, cha), bn_affine1<false>(va[k].w, cha));
            if (relu_b)
                r = make_float4(bn_affine1<true>(vb[k].x, chb), bn_affine1<true>(vb[k].y, chb),
                                bn_affine1<true>(vb[k].z, chb), bn_affine1<true>(vb[k].w, chb));
            else
                r = make_float4(bn_affine1<false>(vb[k].x, chb), bn_affine1<false>(vb[k].y, chb),
                                bn_affine1<false>(vb[k].z, chb), bn_affine1<false>(vb[k].w, chb));
            y[idx[k]] = make_float4(relu1(o.x + r.x), relu1(o.y + r.y), relu1(o.z + r.z), relu1(o.w + r.w));
        }
    }
}

struct DualBwdArgs {
    const float4 *dy2;            // second incoming gradient of the block's output, may be nullptr
    const float4 *out;            // the block's output (mask of the outer ReLU)
    const float4 *xb;
    const float *tbl_b;
    float4 *dxb;
    float *dgamma_a, *dbeta_a, *dgamma_b, *dbeta_b;
    double inv_m;
};

template <int T, int F4>
__global__ __launch_bounds__(T) void k_bn_dual_bwd(
    const float4 *__restrict__ dy, const float4 *__restrict__ xa, const float *__restrict__ tbl_a,
    float4 *__restrict__ dxa, int relu_a, int relu_b, int N, int C, ResPlan pl, unsigned *sync, DualBwdArgs a) {
    constexpr int NW = T / kWave;
    __shared__ double red[4 * NW * 8];
    __shared__ double xch[4];
    const int t = threadIdx.x;
    int cb, s;
    res_block_coords(pl, cb, s);
    const int c0 = pl.c_off + cb * pl.G;
    const int n0 = s * pl.nps;
    const int units = max(0, min(N, n0 + pl.nps) - n0) * pl.gq;
    const int c_local = (pl.G == 1) ? 0 : (t % pl.gq) / pl.q4;
    ResXch xca{}, xcb{};
    if (pl.S > 1) {
        xca = res_xch_begin(sync, pl.xoff, cb, s, pl.S);
        xcb = res_xch_begin(sync, pl.xoff, cb + C, s, pl.S);
    }
    const float4 cha = *reinterpret_cast<const float4 *>(tbl_a + static_cast<size_t>(c0 + c_local) * kTbl);
    const float4 chb = *reinterpret_cast<const float4 *>(a.tbl_b + static_cast<size_t>(c0 + c_local) * kTbl);
    float4 dz[F4], ha[F4], hb[F4];
    unsigned idx[F4];
#pragma unroll
    for (int k = 0; k < F4; ++k) {
        const int j = t + k * T;
        dz[k] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        ha[k] = make_float4(cha.x, cha.x, cha.x, cha.x);      // -> xhat 0 for the unused units
        hb[k] = make_float4(chb.x, chb.x, chb.x, chb.x);
        idx[k] = 0;
        if (j < units) {
            const unsigned row = fdiv(static_cast<unsigned>(j), pl.gqdiv);
            idx[k] = static_cast<unsigned>((static_cast<size_t>(n0 + row) * C + c0) * pl.q4 + (j - row * pl.gq));
            float4 g = dy[idx[k]];
            ha[k] = xa[idx[k]];
            hb[k] = a.xb[idx[k]];
            const float4 o = a.out[idx[k]];
            if (a.dy2) {
                const float4 e = a.dy2[idx[k]];
                g = make_float4(g.x + e.x, g.y + e.y, g.z + e.z, g.w + e.w);
            }
            dz[k] = make_float4(o.x > 0.0f ? g.x : 0.0f, o.y > 0.0f ? g.y : 0.0f, o.z > 0.0f ? g.z : 0.0f,
                                o.w > 0.0f ? g.w : 0.0f);
        }
    }
    // dz_a = d masked by layer a's own ReLU (recomputed from xhat, as in k_bn_res_bwd), dz_b likewise: only d and the
    // two xhat stay in registers, the masks are formed again in the write phase (same expression, same bits)
    float a0 = 0.0f, a1 = 0.0f, b0 = 0.0f, b1 = 0.0f;
#pragma unroll
    for (int k = 0; k < F4; ++k) {
        const float4 g = dz[k], va = ha[k], vb = hb[k];
        float4 za, zb;
        res_bwd_prep(g.x, va.x, cha, relu_a, za.x, ha[k].x);
        res_bwd_prep(g.y, va.y, cha, relu_a, za.y, ha[k].y);
        res_bwd_prep(g.z, va.z, cha, relu_a, za.z, ha[k].z);
        res_bwd_prep(g.w, va.w, cha, relu_a, za.w, ha[k].w);
        res_bwd_prep(g.x, vb.x, chb, relu_b, zb.x, hb[k].x);
        res_bwd_prep(g.y, vb.y, chb, relu_b, zb.y, hb[k].y);
        res_bwd_prep(g.z, vb.z, chb, relu_b, zb.z, hb[k].z);
        res_bwd_prep(g.w, vb.w, chb, relu_b, zb.w, hb[k].w);
        a0 = fmaf(za.x, ha[k].x, a0);
        a0 = fmaf(za.y, ha[k].y, a0);
        a0 = fmaf(za.z, ha[k].z, a0);
        a0 = fmaf(za.w, ha[k].w, a0);
        a1 += (za.x + za.y) + (za.z + za.w);
        b0 = fmaf(zb.x, hb[k].x, b0);
        b0 = fmaf(zb.y, hb[k].y, b0);
        b0 = fmaf(zb.z, hb[k].z, b0);
        b0 = fmaf(zb.w, hb[k].w, b0);
        b1 += (zb.x + zb.y) + (zb.z + zb.w);
    }
    double aga = static_cast<double>(a0), ab = static_cast<double>(a1);
    double agb = static_cast<double>(b0), ab2 = static_cast<double>(b1);
    res_block_sums4<T>(aga, ab, agb, ab2, pl, c_local, red);
    if (pl.S > 1) {
        if (t < kWave) {
            res_exchange(aga, ab, xca, s, pl.S, sync, DEEPIPR_RES_SPIN(pl), DEEPIPR_RES_DROP(pl));
            res_exchange(agb, ab2, xcb, s, pl.S, sync, DEEPIPR_RES_SPIN(pl), DEEPIPR_RES_DROP(pl));
            if (t == 0) {
                xch[0] = aga;
                xch[1] = ab;
                xch[2] = agb;
                xch[3] = ab2;
            }
        }
        __syncthreads();
        aga = xch[0];
        ab = xch[1];
        agb = xch[2];
        ab2 = xch[3];
    }
    if (s == 0 && t < pl.gq && t == c_local * pl.q4) {
        const int c = c0 + c_local;
        a.dgamma_a[c] = static_cast<float>(aga);
        a.dbeta_a[c] = static_cast<float>(ab);
        a.dgamma_b[c] = static_cast<float>(agb);
        a.dbeta_b[c] = static_cast<float>(ab2);
    }
    const float c2a = static_cast<float>(ab * a.inv_m), c3a = static_cast<float>(aga * a.inv_m);
    const float c2b = static_cast<float>(ab2 * a.inv_m), c3b = static_cast<float>(agb * a.inv_m);
    const float sca = cha.z * cha.y, scb = chb.z * chb.y;
#pragma unroll
    for (int k = 0; k < F4; ++k) {
        const int j = t + k * T;
        if (j < units) {
            const float4 g = dz[k];
            auto masked = [](float dv, float xh, const float4 &ch, int relu) {
                return (relu && !(__fadd_rn(__fmul_rn(ch.z, xh), ch.w) > 0.0f)) ? 0.0f : dv;
            };
            const float4 za = make_float4(masked(g.x, ha[k].x, cha, relu_a), masked(g.y, ha[k].y, cha, relu_a),
                                          masked(g.z, ha[k].z, cha, relu_a), masked(g.w, ha[k].w, cha, relu_a));
            const float4 zb = make_float4(masked(g.x, hb[k].x, chb, relu_b), masked(g.y, hb[k].y, chb, relu_b),
                                          masked(g.z, hb[k].z, chb, relu_b), masked(g.w, hb[k].w, chb, relu_b));
            dxa[idx[k]] = make_float4(sca * (za.x - c2a - ha[k].x * c3a), sca * (za.y - c2a - ha[k].y * c3a),
                                      sca * (za.z - c2a - ha[k].z * c3a), sca * (za.w - c2a - ha[k].w * c3a));
            a.dxb[idx[k]] = make_float4(scb * (zb.x - c2b - hb[k].x * c3b), scb * (zb.y - c2b - hb[k].y * c3b),
                                        scb * (zb.z - c2b - hb[k].z * c3b), scb * (zb.w - c2b - hb[k].w * c3b));
        }
    }
}

// ============================================================================================
// GroupNorm / InstanceNorm (affine=False, models/layers/passportconv2d.py:59-62: GroupNorm(o // 16, o),
// InstanceNorm2d(o)) fused with the passport affine + ReLU, register-resident like k_bn_res_*.
//
// The statistics of these norms live on one (sample, group) chunk = cpg adjacent channels of one sample =
// U = cpg*HW/4 contiguous float4 -- no cross-workgroup dependency at all.  A thread group of TG lanes
// (a power of two: part of a wavefront for small chunks, up to 1024 threads for 64 KB chunks) owns one chunk,
// loads it once (F4 float4 per lane), forms mean / invstd, and writes y from registers: 8 B/element.
// Backward, same ownership: dz (ReLU mask recomputed), gz = gamma*dz and xhat stay in registers,
//   dx = invstd * (gz - mean_chunk(gz) - xhat * mean_chunk(gz*xhat))                          12 B/element,
// and the per-channel sums of dz*xhat and dz (dgamma / dbeta before the batch reduction) go through LDS in unit
// order to part[n][2][C]; k_passport_bwd_finish then reduces over n in fixed order and writes dgamma, dbeta, dW.
// ============================================================================================
struct GnPlan {
    int F4, TG, T;        // float4 units per lane, lanes per chunk (power of two), threads per workgroup
    int U, q4, cpg;       // float4 units per chunk, per plane; channels per group
    int groups, chunks;   // groups per sample, N * groups
    FastDiv q4div;
};

// Sum of (a, b) over the TG lanes of this thread's chunk; every lane of the chunk gets the result.
// red: 2 * 16 doubles of LDS (used when TG > 64).
__device__ __forceinline__ void gn_group_sums(double &a, double &b, int TG, double *red) {
    const int lim = TG < kWave ? TG : kWave;
    for (int off = lim >> 1; off > 0; off >>= 1) {
        a += __shfl_xor(a, off, kWave);
        b += __shfl_xor(b, off, kWave);
    }
    if (TG > kWave) {                              // uniform over the launch
        const int wave = threadIdx.x >> 6, wpg = TG >> 6, g0 = (wave / wpg) * wpg;
        __syncthreads();
        if ((threadIdx.x & 63) == 0) {
            red[wave * 2 + 0] = a;
            red[wave * 2 + 1] = b;
        }
        __syncthreads();
        a = 0.0;
        b = 0.0;
        for (int w = 0; w < wpg; ++w) {
            a += red[(g0 + w) * 2 + 0];
            b += red[(g0 + w) * 2 + 1];
        }
    }
}

template <int F4>
__global__ void k_gn_fwd(const float4 *__restrict__ x, float4 *__restrict__ y, const float *__restrict__ gamma,
                         const float *__restrict__ beta, float *__restrict__ stats, int relu, GnPlan pl, float eps) {
    __shared__ double red[32];
    const int t = threadIdx.x;
    const int chunk = blockIdx.x * (pl.T / pl.TG) + t / pl.TG;
    const int lane = t & (pl.TG - 1);
    const bool on = chunk < pl.chunks;
    const size_t base = static_cast<size_t>(on ? chunk : 0) * pl.U;
    // shifted sums (see BnFinishArgs): K = the chunk's first element
    const float K = reinterpret_cast<const float *>(x + base)[0];
    float4 v[F4];
#pragma unroll
    for (int k = 0; k < F4; ++k) {
        const int u = lane + k * pl.TG;
        v[k] = make_float4(K, K, K, K);
        if (on && u < pl.U) v[k] = x[base + u];
    }
    float a0 = 0.0f, a1 = 0.0f;
#pragma unroll
    for (int k = 0; k < F4; ++k) {
        const float dx0 = v[k].x - K, dx1 = v[k].y - K, dx2 = v[k].z - K, dx3 = v[k].w - K;
        a0 += (dx0 + dx1) + (dx2 + dx3);
        a1 = fmaf(dx0, dx0, a1);
        a1 = fmaf(dx1, dx1, a1);
        a1 = fmaf(dx2, dx2, a1);
        a1 = fmaf(dx3, dx3, a1);
    }
    double s1 = static_cast<double>(a0), s2 = static_cast<double>(a1);
    gn_group_sums(s1, s2, pl.TG, red);
    const double inv_m = 1.0 / (static_cast<double>(pl.U) * 4.0);
    const double dmu = s1 * inv_m;
    double var = s2 * inv_m - dmu * dmu;
    if (var < 0.0) var = 0.0;
    const float mean = static_cast<float>(static_cast<double>(K) + dmu);
    const float invstd = static_cast<float>(1.0 / sqrt(var + static_cast<double>(eps)));
    if (on && lane == 0) *reinterpret_cast<float2 *>(stats + static_cast<size_t>(chunk) * 2) = make_float2(mean, invstd);
    const int c0 = (on ? chunk % pl.groups : 0) * pl.cpg;
#pragma unroll
    for (int k = 0; k < F4; ++k) {
        const int u = lane + k * pl.TG;
        if (on && u < pl.U) {
            const int c = c0 + static_cast<int>(fdiv(static_cast<unsigned>(u), pl.q4div));
            const float4 ch = make_float4(mean, invstd, gamma ? gamma[c] : 1.0f, beta ? beta[c] : 0.0f);
            float4 o;
            if (relu)
                o = make_float4(bn_affine1<true>(v[k].x, ch), bn_affine1<true>(v[k].y, ch),
                                bn_affine1<true>(v[k].z, ch), bn_affine1<true>(v[k].w, ch));
            else
                o = make_float4(bn_affine1<false>(v[k].x, ch), bn_affine1<false>(v[k].y, ch),
                                bn_affine1<false>(v[k].z, ch), bn_affine1<false>(v[k].w, ch));
            y[base + u] = o;
        }
    }
}

template <int F4>
__global__ void k_gn_bwd(const float4 *__restrict__ dy, const float4 *__restrict__ x,
                         const float *__restrict__ stats, const float *__restrict__ gamma,
                         const float *__restrict__ beta, float4 *__restrict__ dx, double *__restrict__ part, int relu,
                         int C, GnPlan pl) {
    extern __shared__ __attribute__((aligned(16))) float unit_sums[];      // [chunks per workgroup][U][2]
    __shared__ double red[32];
    const int t = threadIdx.x;
    const int cpb = pl.T / pl.TG;
    const int slot = t / pl.TG;
    const int chunk = blockIdx.x * cpb + slot;
    const int lane = t & (pl.TG - 1);
    const bool on = chunk < pl.chunks;
    const size_t base = static_cast<size_t>(on ? chunk : 0) * pl.U;
    const int c0 = (on ? chunk % pl.groups : 0) * pl.cpg;
    float2 st = make_float2(0.0f, 1.0f);
    if (on) st = *reinterpret_cast<const float2 *>(stats + static_cast<size_t>(chunk) * 2);
    float4 gz[F4], xh[F4];
#pragma unroll
    for (int k = 0; k < F4; ++k) {
        const int u = lane + k * pl.TG;
        gz[k] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        xh[k] = make_float4(st.x, st.x, st.x, st.x);
        if (on && u < pl.U) {
            gz[k] = dy[base + u];
            xh[k] = x[base + u];
        }
    }
    float a0 = 0.0f, a1 = 0.0f;
    float *mine = unit_sums + static_cast<size_t>(slot) * pl.U * 2;
#pragma unroll
    for (int k = 0; k < F4; ++k) {
        const int u = lane + k * pl.TG;
        const int c = c0 + ((on && u < pl.U) ? static_cast<int>(fdiv(static_cast<unsigned>(u), pl.q4div)) : 0);
        const float4 ch = make_float4(st.x, st.y, gamma ? gamma[c] : 1.0f, beta ? beta[c] : 0.0f);
        const float4 d = gz[k], v = xh[k];
        float4 dz;
        res_bwd_prep(d.x, v.x, ch, relu, dz.x, xh[k].x);
        res_bwd_prep(d.y, v.y, ch, relu, dz.y, xh[k].y);
        res_bwd_prep(d.z, v.z, ch, relu, dz.z, xh[k].z);
        res_bwd_prep(d.w, v.w, ch, relu, dz.w, xh[k].w);
        if (on && u < pl.U) {                       // this unit's share of its channel's dgamma / dbeta
            float p0 = dz.x * xh[k].x;
            p0 = fmaf(dz.y, xh[k].y, p0);
            p0 = fmaf(dz.z, xh[k].z, p0);
            p0 = fmaf(dz.w, xh[k].w, p0);
            *reinterpret_cast<float2 *>(mine + static_cast<size_t>(u) * 2) =
                make_float2(p0, (dz.x + dz.y) + (dz.z + dz.w));
        }
        gz[k] = make_float4(ch.z * dz.x, ch.z * dz.y, ch.z * dz.z, ch.z * dz.w);
        a0 = fmaf(gz[k].x, xh[k].x, a0);
        a0 = fmaf(gz[k].y, xh[k].y, a0);
        a0 = fmaf(gz[k].z, xh[k].z, a0);
        a0 = fmaf(gz[k].w, xh[k].w, a0);
        a1 += (gz[k].x + gz[k].y) + (gz[k].z + gz[k].w);
    }
    double s2 = static_cast<double>(a0), s1 = static_cast<double>(a1);
    gn_group_sums(s2, s1, pl.TG, red);
    const double inv_m = 1.0 / (static_cast<double>(pl.U) * 4.0);
    const float c2 = static_cast<float>(s1 * inv_m), c3 = static_cast<float>(s2 * inv_m);
#pragma unroll
    for (int k = 0; k < F4; ++k) {
        const int u = lane + k * pl.TG;
        if (on && u < pl.U)
            dx[base + u] = make_float4(st.y * (gz[k].x - c2 - xh[k].x * c3), st.y * (gz[k].y - c2 - xh[k].y * c3),
                                       st.y * (gz[k].z - c2 - xh[k].z * c3), st.y * (gz[k].w - c2 - xh[k].w * c3));
    }
    // per-channel sums of this workgroup's chunks: `gsz` lanes (a power of two >= min(q4, 64)) per (chunk, channel)
    // job, 64 / gsz jobs per wavefront pass, fixed order
    __syncthreads();
    const int wave = t >> 6, wl = t & 63, nw = pl.T >> 6;
    int gsz = 1;
    while (gsz < pl.q4 && gsz < kWave) gsz <<= 1;
    const int jpw = kWave / gsz, jobs = cpb * pl.cpg;
    const int sub = wl / gsz, e0 = wl - sub * gsz;
    for (int j0 = wave * jpw; j0 < jobs; j0 += nw * jpw) {
        const int job = j0 + sub;
        const int sl = job / pl.cpg, cc = job - sl * pl.cpg;
        const int ck = blockIdx.x * cpb + sl;
        const bool live = job < jobs && ck < pl.chunks;
        double g = 0.0, b = 0.0;
        if (live) {
            const float *p = unit_sums + (static_cast<size_t>(sl) * pl.U + static_cast<size_t>(cc) * pl.q4) * 2;
            for (int e = e0; e < pl.q4; e += gsz) {
                const float2 q = *reinterpret_cast<const float2 *>(p + static_cast<size_t>(e) * 2);
                g += static_cast<double>(q.x);
                b += static_cast<double>(q.y);
            }
        }
        for (int off = gsz >> 1; off > 0; off >>= 1) {
            g += __shfl_xor(g, off, kWave);
            b += __shfl_xor(b, off, kWave);
        }
        if (live && e0 == 0) {
            const int n = ck / pl.groups, c = (ck - n * pl.groups) * pl.cpg + cc;
            part[(static_cast<size_t>(n) * 2 + 0) * C + c] = g;
            part[(static_cast<size_t>(n) * 2 + 1) * C + c] = b;
        }
    }
}

// ============================================================================================
// SGD with momentum and weight decay over one flat parameter buffer (experiments/classification.py:47-50:
// optim.SGD(lr, momentum=0.9, weight_decay=1e-4); torch semantics: g += wd*p; buf = mu*buf + g; p -= lr*buf).
// One streaming pass: reads p, g, buf and writes p, buf = 20 B per parameter.  `grad_scale` folds the
// 1/world_size of a summed all-reduce into the same pass.
// ============================================================================================
// `hp` != nullptr: {lr, momentum, weight decay, grad scale} are read from device memory instead of the launch
// arguments, so a step captured in a hipGraph follows a learning-rate schedule (the host rewrites the four floats
// between replays; launch arguments are frozen at capture time).
__global__ __launch_bounds__(kThreads) void k_sgd_momentum_v4(float4 *__restrict__ p, const float4 *__restrict__ g,
                                                              float4 *__restrict__ buf, size_t n4, float lr,
                                                              float mu, float wd, float gs,
                                                              const float *__restrict__ hp) {
    if (hp) {
        lr = hp[0];
        mu = hp[1];
        wd = hp[2];
        gs = hp[3];
    }
    const size_t step = static_cast<size_t>(gridDim.x) * kThreads;
    for (size_t i = static_cast<size_t>(blockIdx.x) * kThreads + threadIdx.x; i < n4; i += step) {
        float4 pv = p[i], bv = buf[i];
        const float4 gv = g[i];
        float d;
        d = fmaf(wd, pv.x, gs * gv.x); bv.x = fmaf(mu, bv.x, d); pv.x = fmaf(-lr, bv.x, pv.x);
        d = fmaf(wd, pv.y, gs * gv.y); bv.y = fmaf(mu, bv.y, d); pv.y = fmaf(-lr, bv.y, pv.y);
        d = fmaf(wd, pv.z, gs * gv.z); bv.z = fmaf(mu, bv.z, d); pv.z = fmaf(-lr, bv.z, pv.z);
        d = fmaf(wd, pv.w, gs * gv.w); bv.w = fmaf(mu, bv.w, d); pv.w = fmaf(-lr, bv.w, pv.w);
        p[i] = pv;
        buf[i] = bv;
    }
}

__global__ __launch_bounds__(kThreads) void k_sgd_momentum_s(float *__restrict__ p, const float *__restrict__ g,
                                                             float *__restrict__ buf, size_t n, float lr, float mu,
                                                             float wd, float gs, const float *__restrict__ hp) {
    if (hp) {
        lr = hp[0];
        mu = hp[1];
        wd = hp[2];
        gs = hp[3];
    }
    const size_t step = static_cast<size_t>(gridDim.x) * kThreads;
    for (size_t i = static_cast<size_t>(blockIdx.x) * kThreads + threadIdx.x; i < n; i += step) {
        const float d = fmaf(wd, p[i], gs * g[i]);
        const float b = fmaf(mu, buf[i], d);
        buf[i] = b;
        p[i] = fmaf(-lr, b, p[i]);
    }
}


// The same update with the gradients read WHERE AUTOGRAD LEFT THEM: one workgroup per table entry
// {gradient address, offset into the flat parameter / momentum buffers, element count <= kSgdChunk}.  No packing pass
// (FlatSGD's `cat` of 44.7 MB of gradients into the flat gradient buffer: 26 us and 89 MB per step) is needed when
// the gradients need not be contiguous, i.e. without a gradient all-reduce; the addresses are stable when the step
// is replayed from a hipGraph, so the table is built once at capture time.
constexpr int kSgdChunk = 4096;

__global__ __launch_bounds__(kThreads) void k_sgd_momentum_multi(float *__restrict__ p, float *__restrict__ buf,
                                                                 const long long *__restrict__ table,
                                                                 const float *__restrict__ hp) {
    const long long *e = table + 3 * static_cast<size_t>(blockIdx.x);
    const float *__restrict__ g = reinterpret_cast<const float *>(static_cast<uintptr_t>(e[0]));
    const size_t off = static_cast<size_t>(e[1]);
    const int n = static_cast<int>(e[2]);
    const float lr = hp[0], mu = hp[1], wd = hp[2], gs = hp[3];
    float *pp = p + off, *bb = buf + off;
    if ((n & 3) == 0 && (reinterpret_cast<uintptr_t>(g) & 15u) == 0 && (off & 3u) == 0) {
        const float4 *g4 = reinterpret_cast<const float4 *>(g);
        float4 *p4 = reinterpret_cast<float4 *>(pp), *b4 = reinterpret_cast<float4 *>(bb);
        for (int i = threadIdx.x; i < n / 4; i += kThreads) {
            float4 pv = p4[i], bv = b4[i];
            const float4 gv = g4[i];
            float d;
            d = fmaf(wd, pv.x, gs * gv.x); bv.x = fmaf(mu, bv.x, d); pv.x = fmaf(-lr, bv.x, pv.x);
            d = fmaf(wd, pv.y, gs * gv.y); bv.y = fmaf(mu, bv.y, d); pv.y = fmaf(-lr, bv.y, pv.y);
            d = fmaf(wd, pv.z, gs * gv.z); bv.z = fmaf(mu, bv.z, d); pv.z = fmaf(-lr, bv.z, pv.z);
            d = fmaf(wd, pv.w, gs * gv.w); bv.w = fmaf(mu, bv.w, d); pv.w = fmaf(-lr, bv.w, pv.w);
            p4[i] = pv;
            b4[i] = bv;
        }
    } else {
        for (int i = threadIdx.x; i < n; i += kThreads) {
            const float d = fmaf(wd, pp[i], gs * g[i]);
            const float b = fmaf(mu, bb[i], d);
            bb[i] = b;
            pp[i] = fmaf(-lr, b, pp[i]);
        }
    }
}


// ============================================================================================
// Head of the train step: mean cross-entropy + top-1 accuracy of logits [N][C] in ONE launch, and its backward in one
// (experiments/trainer.py:136,149: F.cross_entropy(pred, target) and accuracy(pred, target)[0] -- log_softmax,
// nll_loss, topk, eq, sum, mul_ and their backward: ~12 launch-latency-bound ATen kernels per step).
// One workgroup; a wavefront per row (lanes stride over the classes), f64 accumulation, fixed-order combine.
// ============================================================================================
// Forward = two launches: k_ce_rows (one wavefront per row, all rows in parallel: per-row logsumexp, loss term and
// arg-max hit) and k_ce_finish (one workgroup: fixed-order f64 sum over the rows -> mean loss, top-1 percent).
// A first version looped one workgroup's 16 wavefronts over all rows: 20 us for 128 x 10 logits, a chain of dependent
// load latencies per row; with a wavefront per row the pair takes ~2 x 3 us.
__global__ __launch_bounds__(kThreads) void k_ce_rows(const float *__restrict__ logits,
                                                      const long long *__restrict__ target, int N, int C,
                                                      float *__restrict__ lse, double *__restrict__ part /* [N][2] */) {
    const int n = blockIdx.x * (kThreads / kWave) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (n >= N) return;                                           // whole wavefront
    const float *row = logits + static_cast<size_t>(n) * C;
    const long long t = target[n];
    float mx = -INFINITY;
    int arg = 0x7fffffff;
    for (int c = lane; c < C; c += kWave) {
        const float v = row[c];
        if (v > mx || (v == mx && c < arg)) {
            mx = v;
            arg = c;
        }
    }
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) {               // arg-max butterfly: larger value, then lower index
        const float ov = __shfl_xor(mx, off, kWave);
        const int oa = __shfl_xor(arg, off, kWave);
        if (ov > mx || (ov == mx && oa < arg)) {
            mx = ov;
            arg = oa;
        }
    }
    double sum = 0.0;
    for (int c = lane; c < C; c += kWave) sum += static_cast<double>(expf(row[c] - mx));     // terms <= 1, f64 sum
    sum = wave_sum(sum);
    if (lane == 0) {
        const double l = static_cast<double>(mx) + log(sum);
        lse[n] = static_cast<float>(l);
        // a label outside [0, C) (ATen: device assert; also F.cross_entropy's ignore_index = -100, which this head does
        // not implement) must not read out of bounds and must not pass silently: the loss becomes NaN
        const bool valid = t >= 0 && t < static_cast<long long>(C);
        part[2 * static_cast<size_t>(n) + 0] = valid ? l - static_cast<double>(row[t])
                                                     : __longlong_as_double(0x7ff8000000000000LL);
        part[2 * static_cast<size_t>(n) + 1] = (valid && static_cast<long long>(arg) == t) ? 1.0 : 0.0;
    }
}

__global__ __launch_bounds__(kThreads) void k_ce_finish(const double *__restrict__ part, int N,
                                                        float *__restrict__ loss, float *__restrict__ top1_pct) {
    __shared__ double red[8];
    double a = 0.0, b = 0.0;
    for (int n = threadIdx.x; n < N; n += kThreads) {
        a += part[2 * static_cast<size_t>(n) + 0];
        b += part[2 * static_cast<size_t>(n) + 1];
    }
    a = block_sum(a, red);
    b = block_sum(b, red + 4);
    if (threadIdx.x == 0) {
        *loss = static_cast<float>(a / N);
        *top1_pct = static_cast<float>(b * (100.0 / N));
    }
}

__global__ __launch_bounds__(kThreads) void k_ce_bwd(const float *__restrict__ dloss, const float *__restrict__ logits,
                                                     const long long *__restrict__ target,
                                                     const float *__restrict__ lse, int N, int C,
                                                     float *__restrict__ dlogits) {
    const float scale = dloss[0] / static_cast<float>(N);
    const size_t total = static_cast<size_t>(N) * C;
    const size_t step = static_cast<size_t>(gridDim.x) * kThreads;
    for (size_t i = static_cast<size_t>(blockIdx.x) * kThreads + threadIdx.x; i < total; i += step) {
        const int n = static_cast<int>(i / C), c = static_cast<int>(i - static_cast<size_t>(n) * C);
        const float p = expf(logits[i] - lse[n]);
        const long long tn = target[n];
        const float g = scale * (p - (static_cast<long long>(c) == tn ? 1.0f : 0.0f));
        dlogits[i] = (tn >= 0 && tn < static_cast<long long>(C)) ? g : __int_as_float(0x7fc00000);   // invalid label: NaN row
    }
}


// ============================================================================================
// The classifier of the CIFAR-geometry nets: logits = Linear(avg_pool(x))  (models/resnet_passport.py:226-228 of this package =
// the reference's `out = F.avg_pool2d(out, 4); out = out.view(out.size(0), -1); out = self.linear(out)`, models/resnet_passport.py:
// 127-129 there) in ONE launch, its backward in one.  The library path is five launch-latency-bound kernels forward and backward
// (mean reduce 8 us, a 128 x 512 x 10 GEMM the BLAS library takes 16 us for, two more GEMMs, a broadcast-divide, a bias reduce:
// 65 us of the 3.5 ms step); the data is 4 MB.
//   forward  workgroup = (image, eight classes): the channel means (kept for the backward) through LDS, then a wavefront per class:
//            lanes stride over the channels, butterfly sum (fixed order), + bias.
//   backward the first N x C / 256 workgroups: dx[n][c][:] = (sum_k dlogits[n][k] W[k][c]) / HW, a thread per channel;
//            the others: dW for 64 channels x 8 classes (+ db from the channel block 0 ones): four image groups, each a
//            sequential sum over its images, combined in group order through LDS -- bit-reproducible.
// ============================================================================================
constexpr int kHeadMaxC = 4096, kHeadMaxK = 128;

__global__ __launch_bounds__(kThreads) void k_pooled_linear_fwd(const float *__restrict__ x, const float *__restrict__ W,
                                                                const float *__restrict__ b, float *__restrict__ pooled,
                                                                float *__restrict__ logits, int C, int HW, int K, int kblocks) {
    // workgroup = (image n, block of eight classes): every class block pools the image again (32 KB from L2) rather than wait
    // for one that did -- 100 classes are 13 x N workgroups instead of N
    __shared__ float mean_s[kHeadMaxC];
    const int n = blockIdx.x / kblocks, kb = blockIdx.x - n * kblocks;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const float inv = 1.0f / static_cast<float>(HW);
    const float *xn = x + static_cast<size_t>(n) * C * HW;
    const int q4 = HW / 4;
    if (q4 <= 16 && (q4 & (q4 - 1)) == 0) {
        // consecutive lanes read consecutive float4 of the image (whole cache lines per wavefront); the q4 lanes of a channel
        // meet in a butterfly (fixed order)
        const float4 *x4 = reinterpret_cast<const float4 *>(xn);
        for (int i = t; i < C * q4; i += kThreads) {               // C * q4 is a multiple of the wavefront: no lane drops out
            const float4 v = x4[i];
            float s = (v.x + v.y) + (v.z + v.w);
            for (int off = q4 >> 1; off > 0; off >>= 1) s += __shfl_xor(s, off, kWave);
            if ((i & (q4 - 1)) == 0) {
                const int c = i / q4;
                s *= inv;
                mean_s[c] = s;
                if (kb == 0) pooled[static_cast<size_t>(n) * C + c] = s;
            }
        }
    } else {
        for (int c = t; c < C; c += kThreads) {
            const float4 *row = reinterpret_cast<const float4 *>(xn + static_cast<size_t>(c) * HW);
            float s = 0.0f;
            for (int q = 0; q < q4; ++q) {
                const float4 v = row[q];
                s += (v.x + v.y) + (v.z + v.w);
            }
            s *= inv;
            mean_s[c] = s;
            if (kb == 0) pooled[static_cast<size_t>(n) * C + c] = s;
        }
    }
    __syncthreads();
    for (int k = kb * 8 + wave; k < min(K, kb * 8 + 8); k += kThreads / kWave) {
        const float *wk = W + static_cast<size_t>(k) * C;
        float s = 0.0f;
        for (int c = lane; c < C; c += kWave) s = fmaf(mean_s[c], wk[c], s);
#pragma unroll
        for (int off = kWave / 2; off > 0; off >>= 1) s += __shfl_xor(s, off, kWave);
        if (lane == 0) logits[static_cast<size_t>(n) * K + k] = s + (b ? b[k] : 0.0f);
    }
}

__global__ __launch_bounds__(kThreads) void k_pooled_linear_bwd(const float *__restrict__ dl, const float *__restrict__ W,
                                                                const float *__restrict__ pooled, float *__restrict__ dx,
                                                                float *__restrict__ dW, float *__restrict__ db, int N, int C,
                                                                int HW, int K) {
    __shared__ float sh[kThreads * 8 + kHeadMaxK];
    const int t = threadIdx.x;
    const int slabs = (C + kThreads - 1) / kThreads;               // dx workgroups per image: 256 channels each
    if (static_cast<int>(blockIdx.x) < N * slabs) {
        const int n = blockIdx.x / slabs, c = (blockIdx.x - n * slabs) * kThreads + t;
        float *dls = sh;
        for (int k = t; k < K; k += kThreads) dls[k] = dl[static_cast<size_t>(n) * K + k];
        __syncthreads();
        float g = 0.0f;
        if (c < C) {
            for (int k = 0; k < K; ++k) g = fmaf(dls[k], W[static_cast<size_t>(k) * C + c], g);   // coalesced over the channels
            g *= 1.0f / static_cast<float>(HW);
        }
        float *gs = sh + kHeadMaxK;                                // the slab's 256 channel gradients, then whole cache lines out
        gs[t] = g;
        __syncthreads();
        const int q4 = HW / 4, c0 = (blockIdx.x - n * slabs) * kThreads, live = min(kThreads, C - c0) * q4;
        float4 *o = reinterpret_cast<float4 *>(dx + (static_cast<size_t>(n) * C + c0) * HW);
        for (int i = t; i < live; i += kThreads) {
            const float v = gs[i / q4];
            o[i] = make_float4(v, v, v, v);
        }
        return;
    }
    // dW[k0 .. k0 + 7][c0 .. c0 + 63]: thread = (channel lane, image group g of four); group g sums images g, g + 4, ... in order
    const int cblocks = C / 64, j = blockIdx.x - N * slabs, cb = j % cblocks, kb = j / cblocks;
    const int c = cb * 64 + (t & 63), g = t >> 6, k0 = kb * 8;
    float acc[8], accb[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = accb[e] = 0.0f;
#pragma unroll 4
    for (int n = g; n < N; n += 4) {
        const float pv = pooled[static_cast<size_t>(n) * C + c];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float d = k0 + e < K ? dl[static_cast<size_t>(n) * K + k0 + e] : 0.0f;
            acc[e] = fmaf(d, pv, acc[e]);
            accb[e] += d;
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) sh[t * 8 + e] = acc[e];
    float *bsh = sh + kThreads * 8;
    if ((t & 63) == 0 && cb == 0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) bsh[g * 8 + e] = accb[e];
    }
    __syncthreads();
    if (g == 0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            if (k0 + e < K) {
                const float v = ((sh[t * 8 + e] + sh[(t + 64) * 8 + e]) + sh[(t + 128) * 8 + e]) + sh[(t + 192) * 8 + e];
                dW[static_cast<size_t>(k0 + e) * C + c] = v;
            }
        }
        if (cb == 0 && db && t < 8 && k0 + t < K) db[k0 + t] = ((bsh[t] + bsh[8 + t]) + bsh[16 + t]) + bsh[24 + t];
    }
}

// ============================================================================================
// Residual tail of a block: out = relu(a + b)  (models/resnet_passport.py:77-84: out += shortcut; F.relu(out)).
// One 12 B/elt pass instead of ATen's add (12 B/elt) + relu (8 B/elt); backward is one masked copy shared by
// both inputs: d = dy * [out > 0].
// ============================================================================================
// The step's scalar bookkeeping in one launch: out[0] = t[0] + ... + t[na-1], out[1] = t[na] + ... + t[na+nb-1] (both left to
// right, as the chain of aten::add the reference's `loss + sign_loss` loops are, trainer.py:140-145 / trainer_private.py:163-173),
// out[2] = out[0] + out[1].  Ten 5-us launches of one-element adds per V2 step otherwise.
constexpr int kScalarTermsMax = 48;
struct ScalarTerms {
    const float *p[kScalarTermsMax];
    int na, nb;
};
__global__ void k_scalar_sums(ScalarTerms T, float *__restrict__ out) {
    if (threadIdx.x != 0) return;
    float a = 0.f, b = 0.f;
    for (int i = 0; i < T.na; ++i) a = i ? a + *T.p[i] : *T.p[i];
    for (int i = 0; i < T.nb; ++i) b = i ? b + *T.p[T.na + i] : *T.p[T.na + i];
    out[0] = a;
    out[1] = b;
    out[2] = T.na && T.nb ? a + b : (T.na ? a : b);
}

__global__ __launch_bounds__(kThreads) void k_add_relu_fwd(const float *__restrict__ a, const float *__restrict__ b,
                                                           float *__restrict__ out, size_t n) {
    const size_t step = static_cast<size_t>(gridDim.x) * kThreads;
    const size_t n4 = n / 4;
    const float4 *a4 = reinterpret_cast<const float4 *>(a), *b4 = reinterpret_cast<const float4 *>(b);
    float4 *o4 = reinterpret_cast<float4 *>(out);
    for (size_t i = static_cast<size_t>(blockIdx.x) * kThreads + threadIdx.x; i < n4; i += step) {
        const float4 u = a4[i], v = b4[i];
        o4[i] = make_float4(relu1(u.x + v.x), relu1(u.y + v.y), relu1(u.z + v.z), relu1(u.w + v.w));
    }
    for (size_t i = n4 * 4 + static_cast<size_t>(blockIdx.x) * kThreads + threadIdx.x; i < n; i += step)
        out[i] = relu1(a[i] + b[i]);
}

// TWO: the gradient arrives in two pieces (the two consumers of a block's output: the next block's first conv and
// its identity / projection shortcut); their sum is formed here instead of in a separate add kernel.
template <bool TWO>
__global__ __launch_bounds__(kThreads) void k_relu_bwd(const float *__restrict__ dy, const float *__restrict__ dy2,
                                                       const float *__restrict__ out, float *__restrict__ dx,
                                                       size_t n) {
    const size_t step = static_cast<size_t>(gridDim.x) * kThreads;
    const size_t n4 = n / 4;
    const float4 *d4 = reinterpret_cast<const float4 *>(dy), *e4 = reinterpret_cast<const float4 *>(dy2);
    const float4 *o4 = reinterpret_cast<const float4 *>(out);
    float4 *x4 = reinterpret_cast<float4 *>(dx);
    for (size_t i = static_cast<size_t>(blockIdx.x) * kThreads + threadIdx.x; i < n4; i += step) {
        float4 d = d4[i];
        const float4 o = o4[i];
        if (TWO) {
            const float4 e = e4[i];
            d = make_float4(d.x + e.x, d.y + e.y, d.z + e.z, d.w + e.w);
        }
        x4[i] = make_float4(o.x > 0.0f ? d.x : 0.0f, o.y > 0.0f ? d.y : 0.0f, o.z > 0.0f ? d.z : 0.0f,
                            o.w > 0.0f ? d.w : 0.0f);
    }
    for (size_t i = n4 * 4 + static_cast<size_t>(blockIdx.x) * kThreads + threadIdx.x; i < n; i += step) {
        const float d = TWO ? dy[i] + dy2[i] : dy[i];
        dx[i] = out[i] > 0.0f ? d : 0.0f;
    }
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
    prof.bytes = 8.0 * static_cast<double>(total);
    const bool vec = (HW % 4 == 0) && aligned16(xhat) && aligned16(y);
    if (vec) {
        const unsigned n4 = static_cast<unsigned>(total / 4);
        const FastDiv pdiv = make_fastdiv(static_cast<unsigned>(HW / 4));
        const int grid = grid_for(n4) + (with_sign ? 1 : 0);
        if (relu)
            DEEPIPR_LAUNCH(prof, k_affine_fwd_v4<true>, dim3(grid), dim3(kThreads), st, reinterpret_cast<const float4 *>(xhat), gamma, beta,
                               reinterpret_cast<float4 *>(y), n4, pdiv, cdiv, static_cast<unsigned>(C),
                               with_sign ? 1 : 0, sa);
        else
            DEEPIPR_LAUNCH(prof, k_affine_fwd_v4<false>, dim3(grid), dim3(kThreads), st, reinterpret_cast<const float4 *>(xhat), gamma, beta,
                               reinterpret_cast<float4 *>(y), n4, pdiv, cdiv, static_cast<unsigned>(C),
                               with_sign ? 1 : 0, sa);
    } else if (total % 4 == 0 && aligned16(xhat) && aligned16(y)) {
        const unsigned n4 = static_cast<unsigned>(total / 4);
        const FastDiv pdiv = make_fastdiv(static_cast<unsigned>(HW));
        const int grid = grid_for(n4) + (with_sign ? 1 : 0);
        if (relu)
            DEEPIPR_LAUNCH(prof, k_affine_fwd_v4g<true>, dim3(grid), dim3(kThreads), st,
                           reinterpret_cast<const float4 *>(xhat), gamma, beta, reinterpret_cast<float4 *>(y), n4,
                           pdiv, cdiv, static_cast<unsigned>(C), with_sign ? 1 : 0, sa);
        else
            DEEPIPR_LAUNCH(prof, k_affine_fwd_v4g<false>, dim3(grid), dim3(kThreads), st,
                           reinterpret_cast<const float4 *>(xhat), gamma, beta, reinterpret_cast<float4 *>(y), n4,
                           pdiv, cdiv, static_cast<unsigned>(C), with_sign ? 1 : 0, sa);
    } else {
        const FastDiv pdiv = make_fastdiv(static_cast<unsigned>(HW));
        const int grid = grid_for(total) + (with_sign ? 1 : 0);
        if (relu)
            DEEPIPR_LAUNCH(prof, k_affine_fwd_s<true>, dim3(grid), dim3(kThreads), st, xhat, gamma, beta, y,
                               static_cast<unsigned>(total), pdiv, cdiv, static_cast<unsigned>(C),
                               with_sign ? 1 : 0, sa);
        else
            DEEPIPR_LAUNCH(prof, k_affine_fwd_s<false>, dim3(grid), dim3(kThreads), st, xhat, gamma, beta, y,
                               static_cast<unsigned>(total), pdiv, cdiv, static_cast<unsigned>(C),
                               with_sign ? 1 : 0, sa);
    }
    return check_launch("affine_relu_fwd");
}

template <int VEC, bool RELU>
void launch_bwd_t(ProfScope &prof, const float *dy, const float *xh, const float *g, const float *bt, float *dx,
                  double *part, int N, int C, int P, const BwdPlan &pl, hipStream_t st) {
    const dim3 grid(pl.tiles, pl.NS);
    if (pl.large)
        DEEPIPR_LAUNCH(prof, (k_affine_bwd_large<VEC, RELU>), grid, dim3(kThreads), st, dy, xh, g, bt, dx,
                           part, N, C, P, pl);
    else
        DEEPIPR_LAUNCH(prof, (k_affine_bwd_small<VEC, RELU>), grid, dim3(kThreads), st, dy, xh, g, bt, dx,
                           part, N, C, P, pl);
}

int launch_affine_bwd(const float *dy, const float *xh, const float *g, const float *bt, float *dx,
                      double *part, int N, int C, int P, int relu, BwdPlan *plan_out, hipStream_t st) {
    const bool can_vec = aligned16(dy) && aligned16(xh) && aligned16(dx);
    const BwdPlan pl = plan_bwd(N, C, P, can_vec);
    if (pl.NS > 65535) return fail(DEEPIPR_EINVAL, "affine_relu_bwd: too many batch splits");
    ProfScope prof(DEEPIPR_K_AFFINE_BWD, st);
    prof.bytes = 12.0 * static_cast<double>(N) * C * P;
    if (pl.VEC == 4) {
        if (relu) launch_bwd_t<4, true>(prof, dy, xh, g, bt, dx, part, N, C, P, pl, st);
        else launch_bwd_t<4, false>(prof, dy, xh, g, bt, dx, part, N, C, P, pl, st);
    } else {
        if (relu) launch_bwd_t<1, true>(prof, dy, xh, g, bt, dx, part, N, C, P, pl, st);
        else launch_bwd_t<1, false>(prof, dy, xh, g, bt, dx, part, N, C, P, pl, st);
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

int launch_passport_finish(const double *part, int NS, int C, const float *gamma, const float *b, float alpha,
                           float margin, float l2, const float *dloss, const float *dgamma_extra,
                           const float *dbeta_extra, const double *s, int K, float *dgamma, float *dbeta, float *dW,
                           hipStream_t st) {
    ProfScope prof(DEEPIPR_K_PASSPORT_BWD_FINISH, st);
    const bool vec = dW && K % 4 == 0 && aligned16(dW) && aligned16(s);
#define DEEPIPR_FINISH_ARGS part, NS, C, gamma, b, alpha, margin, l2, dloss, dgamma_extra, dbeta_extra, s, K, dgamma, dbeta, dW
    if (C >= 2 * kRowPairMinCo) {
        const dim3 grid((C + 1) / 2);
        if (vec) DEEPIPR_LAUNCH(prof, (k_passport_bwd_finish<true, 2>), grid, dim3(kThreads), st, DEEPIPR_FINISH_ARGS);
        else DEEPIPR_LAUNCH(prof, (k_passport_bwd_finish<false, 2>), grid, dim3(kThreads), st, DEEPIPR_FINISH_ARGS);
    } else {
        if (vec) DEEPIPR_LAUNCH(prof, (k_passport_bwd_finish<true, 1>), dim3(C), dim3(kThreads), st, DEEPIPR_FINISH_ARGS);
        else DEEPIPR_LAUNCH(prof, (k_passport_bwd_finish<false, 1>), dim3(C), dim3(kThreads), st, DEEPIPR_FINISH_ARGS);
    }
#undef DEEPIPR_FINISH_ARGS
    return check_launch("passport_bwd(finish)");
}

bool bad_dims(int N, int C, int HW) { return N <= 0 || C <= 0 || HW <= 0; }

}  // namespace

// =============================================================================================
// C ABI
// =============================================================================================
extern "C" {

int deepipr_abi_version(void) { return DEEPIPR_ABI_VERSION; }

const char *deepipr_last_error(void) { return g_err; }

// ---------------------------------------------------------------------------------------------
// External events: the hook a captured step offers to work OUTSIDE the graph (include/deepipr_hip.h)
int deepipr_event_create(void **event) {
    if (!event) return fail(DEEPIPR_EINVAL, "event_create: null argument");
    hipEvent_t e = nullptr;
    const hipError_t rc = hipEventCreateWithFlags(&e, hipEventDisableTiming);
    if (rc != hipSuccess) return fail(DEEPIPR_ELAUNCH, "event_create: %s", hipGetErrorString(rc));
    *event = e;
    return DEEPIPR_OK;
}

int deepipr_event_destroy(void *event) {
    if (!event) return DEEPIPR_OK;
    const hipError_t rc = hipEventDestroy(static_cast<hipEvent_t>(event));
    return rc == hipSuccess ? DEEPIPR_OK : fail(DEEPIPR_ELAUNCH, "event_destroy: %s", hipGetErrorString(rc));
}

int deepipr_event_record(void *event, void *stream) {
    if (!event) return fail(DEEPIPR_EINVAL, "event_record: null event");
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipEvent_t ev = static_cast<hipEvent_t>(event);
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    unsigned long long id = 0;
    hipGraph_t graph = nullptr;
    const hipGraphNode_t *deps = nullptr;
    size_t ndeps = 0;
    if (hipStreamGetCaptureInfo_v2(st, &cs, &id, &graph, &deps, &ndeps) != hipSuccess) {
        (void)hipGetLastError();
        cs = hipStreamCaptureStatusNone;
    }
    hipError_t rc;
    if (cs == hipStreamCaptureStatusActive) {
        // On a capturing stream: an event-record NODE behind everything captured so far, and the capture continues
        // behind the node.  Every launch of the graph records the event when its execution reaches the node; a
        // hipStreamWaitEvent issued after the launch call waits for exactly that record.  (The one-call form,
        // hipEventRecordWithFlags(hipEventRecordExternal), returns hipErrorInvalidValue under torch's capture on
        // ROCm 7.0; the explicit node does the same thing.)
        hipGraphNode_t node = nullptr;
        rc = hipGraphAddEventRecordNode(&node, graph, deps, ndeps, ev);
        if (rc == hipSuccess) rc = hipStreamUpdateCaptureDependencies(st, &node, 1, hipStreamSetCaptureDependencies);
    } else {
        rc = hipEventRecord(ev, st);
    }
    return rc == hipSuccess ? DEEPIPR_OK : fail(DEEPIPR_ELAUNCH, "event_record: %s", hipGetErrorString(rc));
}

int deepipr_event_synchronize(void *event) {
    if (!event) return fail(DEEPIPR_EINVAL, "event_synchronize: null event");
    const hipError_t rc = hipEventSynchronize(static_cast<hipEvent_t>(event));
    return rc == hipSuccess ? DEEPIPR_OK : fail(DEEPIPR_ELAUNCH, "event_synchronize: %s", hipGetErrorString(rc));
}

int deepipr_stream_wait_event(void *stream, void *event) {
    if (!event) return fail(DEEPIPR_EINVAL, "stream_wait_event: null event");
    const hipError_t rc = hipStreamWaitEvent(static_cast<hipStream_t>(stream), static_cast<hipEvent_t>(event), 0);
    return rc == hipSuccess ? DEEPIPR_OK : fail(DEEPIPR_ELAUNCH, "stream_wait_event: %s", hipGetErrorString(rc));
}

int deepipr_profile_enable(int on) {
    std::lock_guard<std::mutex> lk(g_prof.mu);
    if (on == 1) {                                   // 1 = start afresh, 2 = resume, 0 = pause
        for (int i = 0; i < DEEPIPR_PROFILE_KERNELS; ++i) {
            g_prof.total_ms[i] = 0.0; g_prof.launches[i] = 0; g_prof.total_bytes[i] = 0.0;
            g_prof.scope_ms[i] = 0.0; g_prof.scope_launches[i] = 0; g_prof.scope_bytes[i] = 0.0;
        }
    }
    g_prof.on = on != 0;
    return DEEPIPR_OK;
}

namespace {
static void prof_drain_locked() {                  // everything recorded so far (static: this namespace sits inside extern "C")
    for (auto &p : g_prof.pending) {
        float ms = 0.0f;
        if (hipEventSynchronize(p.b) == hipSuccess && hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            g_prof.total_ms[p.k] += ms;
            g_prof.launches[p.k] += 1;
            g_prof.total_bytes[p.k] += p.bytes;
            if (p.scope) {
                g_prof.scope_ms[p.k] += ms;
                g_prof.scope_launches[p.k] += 1;
                g_prof.scope_bytes[p.k] += p.bytes;
            }
        }
        g_prof.pool.push_back(p.a);
        g_prof.pool.push_back(p.b);
    }
    g_prof.pending.clear();
}
}  // namespace

int deepipr_profile_read(int kernel, double *total_ms, long long *launches) {
    if (kernel < 0 || kernel >= DEEPIPR_PROFILE_KERNELS || !total_ms || !launches)
        return fail(DEEPIPR_EINVAL, "profile_read: bad argument");
    std::lock_guard<std::mutex> lk(g_prof.mu);
    prof_drain_locked();
    *total_ms = g_prof.total_ms[kernel];
    *launches = g_prof.launches[kernel];
    return DEEPIPR_OK;
}

int deepipr_profile_scope(int scope) {
    std::lock_guard<std::mutex> lk(g_prof.mu);
    g_prof.scope = scope != 0;
    return DEEPIPR_OK;
}

int deepipr_profile_read_scope(int kernel, double *total_ms, long long *launches, double *total_bytes) {
    if (kernel < 0 || kernel >= DEEPIPR_PROFILE_KERNELS || !total_ms || !launches || !total_bytes)
        return fail(DEEPIPR_EINVAL, "profile_read_scope: bad argument");
    std::lock_guard<std::mutex> lk(g_prof.mu);
    prof_drain_locked();
    *total_ms = g_prof.scope_ms[kernel];
    *launches = g_prof.scope_launches[kernel];
    *total_bytes = g_prof.scope_bytes[kernel];
    return DEEPIPR_OK;
}

int deepipr_profile_read_bytes(int kernel, double *total_bytes) {
    if (kernel < 0 || kernel >= DEEPIPR_PROFILE_KERNELS || !total_bytes)
        return fail(DEEPIPR_EINVAL, "profile_read_bytes: bad argument");
    std::lock_guard<std::mutex> lk(g_prof.mu);
    prof_drain_locked();
    *total_bytes = g_prof.total_bytes[kernel];
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
    hipStream_t st0 = static_cast<hipStream_t>(stream);
    ProfScope prof(DEEPIPR_K_POOLED_PATCH_MEAN, st0);
    DEEPIPR_LAUNCH(prof, k_pooled_patch_mean, grid, dim3(kThreads), st0, keys,
                       B, Ci, H, W, kh, kw, stride, pad, Ho, Wo, m_out);
    return check_launch("pooled_patch_mean");
}

int deepipr_gamma_beta_fwd(const float *W, const double *s, int Co, int K, float *gamma, float *beta,
                           void *stream) {
    if (!W || !s || !gamma || !beta || Co <= 0 || K <= 0) return fail(DEEPIPR_EINVAL, "gamma_beta_fwd: bad argument");
    const DeepiprGemvLayer one{W, s, gamma, beta, Co, K};
    return deepipr_gamma_beta_fwd_multi(&one, 1, stream);
}

int deepipr_gamma_beta_fwd_multi(const DeepiprGemvLayer *layers, int n, void *stream) {
    if (!layers || n <= 0 || n > kGemvMaxLayers)
        return fail(DEEPIPR_EINVAL, "gamma_beta_fwd_multi: 1..%d layers per call", kGemvMaxLayers);
    hipStream_t st = static_cast<hipStream_t>(stream);
    double bytes = 0.0;
    for (int i = 0; i < n; ++i) {
        const DeepiprGemvLayer &l = layers[i];
        if (!l.W || !l.m || !l.gamma || !l.beta || l.Co <= 0 || l.K <= 0)
            return fail(DEEPIPR_EINVAL, "gamma_beta_fwd_multi: bad layer %d", i);
        bytes += 4.0 * static_cast<double>(l.Co) * l.K;
    }
    // One row of W per workgroup.  Measured on the five layer4 weights of ResNet18 (33.6 MB, tools/gemv_bench.py with the
    // test build's DEEPIPR_GEMV_ROWS, profiles/r03_gemv_sweep.json): 1 row 10.6 us, 2 rows sharing the pooled vectors
    // 11.4 us, and a form that staged the pooled vectors through LDS for 4 / 8 rows 13.7 / 19.0 us -- the launch is
    // bound by memory-level parallelism (workgroups in flight), not by the L2 traffic of the pooled vectors.
    int rows_per_wg = 1;
#ifdef DEEPIPR_TEST_HOOKS
    if (const char *e = getenv("DEEPIPR_GEMV_ROWS")) {      // tuning (tools/gemv_bench.py)
        if (atoi(e) == 2) rows_per_wg = 2;
    }
#endif
    GemvBatch B{};
    B.n = n;
    int blocks = 0;
    for (int i = 0; i < n; ++i) {
        const DeepiprGemvLayer &l = layers[i];
        B.L[i] = GemvLayer{l.W, l.m, l.gamma, l.beta, l.Co, l.K, blocks,
                           (l.K % 4 == 0 && aligned16(l.W) && aligned16(l.m)) ? 1 : 0};
        blocks += (l.Co + rows_per_wg - 1) / rows_per_wg;
    }
    ProfScope prof(DEEPIPR_K_GAMMA_BETA_FWD, st);
    prof.bytes = bytes;
    if (rows_per_wg == 2) DEEPIPR_LAUNCH(prof, (k_gamma_beta_multi<2>), dim3(blocks), dim3(kThreads), st, B);
    else DEEPIPR_LAUNCH(prof, (k_gamma_beta_multi<1>), dim3(blocks), dim3(kThreads), st, B);
    return check_launch("gamma_beta_fwd");
}

int deepipr_gamma_beta_bwd_multi(const DeepiprRank2Layer *layers, int n, int accumulate, void *stream) {
    if (!layers || n <= 0 || n > kGemvMaxLayers)
        return fail(DEEPIPR_EINVAL, "gamma_beta_bwd_multi: 1..%d layers per call", kGemvMaxLayers);
    hipStream_t st = static_cast<hipStream_t>(stream);
    long long rows = 0;
    double bytes = 0.0;
    for (int i = 0; i < n; ++i) {
        const DeepiprRank2Layer &l = layers[i];
        if (!l.dgamma || !l.dbeta || !l.m || !l.dW || l.Co <= 0 || l.K <= 0)
            return fail(DEEPIPR_EINVAL, "gamma_beta_bwd_multi: bad layer %d", i);
        rows += l.Co;
        bytes += (accumulate ? 8.0 : 4.0) * static_cast<double>(l.Co) * l.K;
    }
    const int cus = device_cu_count();
    const int rpw = rows >= 8ll * (cus > 0 ? cus : 256) ? 2 : 1;
    Rank2Batch B{};
    B.n = n;
    int blocks = 0;
    for (int i = 0; i < n; ++i) {
        const DeepiprRank2Layer &l = layers[i];
        B.L[i] = Rank2Layer{l.dgamma, l.dbeta, l.m, l.dW, l.Co, l.K, blocks,
                            (l.K % 4 == 0 && aligned16(l.dW) && aligned16(l.m)) ? 1 : 0};
        blocks += (l.Co + rpw - 1) / rpw;
    }
    ProfScope prof(DEEPIPR_K_GAMMA_BETA_BWD, st);
    prof.bytes = bytes;
    if (accumulate) {
        if (rpw == 2) DEEPIPR_LAUNCH(prof, (k_rank2_multi<2, true>), dim3(blocks), dim3(kThreads), st, B);
        else DEEPIPR_LAUNCH(prof, (k_rank2_multi<1, true>), dim3(blocks), dim3(kThreads), st, B);
    } else {
        if (rpw == 2) DEEPIPR_LAUNCH(prof, (k_rank2_multi<2, false>), dim3(blocks), dim3(kThreads), st, B);
        else DEEPIPR_LAUNCH(prof, (k_rank2_multi<1, false>), dim3(blocks), dim3(kThreads), st, B);
    }
    return check_launch(accumulate ? "gamma_beta_bwd_acc" : "gamma_beta_bwd");
}

}  // extern "C"

extern "C" {

int deepipr_gamma_beta_bwd(const float *dgamma, const float *dbeta, const double *s, int Co, int K,
                           float *dW, void *stream) {
    if (!dgamma || !dbeta || !s || !dW || Co <= 0 || K <= 0)
        return fail(DEEPIPR_EINVAL, "gamma_beta_bwd: bad argument");
    const DeepiprRank2Layer one{dgamma, dbeta, s, dW, Co, K};
    return deepipr_gamma_beta_bwd_multi(&one, 1, 0, stream);
}

int deepipr_gamma_beta_bwd_acc(const float *dgamma, const float *dbeta, const double *s, int Co, int K,
                               float *dW, void *stream) {
    if (!dgamma || !dbeta || !s || !dW || Co <= 0 || K <= 0)
        return fail(DEEPIPR_EINVAL, "gamma_beta_bwd_acc: bad argument");
    const DeepiprRank2Layer one{dgamma, dbeta, s, dW, Co, K};
    return deepipr_gamma_beta_bwd_multi(&one, 1, 1, stream);
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
    DEEPIPR_LAUNCH(prof, k_dkey_colsum, dim3((K + kThreads - 1) / kThreads, kDkeySplit), dim3(kThreads), st,
                       dgamma, dbeta, W, Co, K, part);
    const int total = 2 * B * Ci * H * Wd;
    const double inv_n = 1.0 / (static_cast<double>(B) * Ho * Wo);
    DEEPIPR_LAUNCH(prof, k_dkey_gather, dim3((total + kThreads - 1) / kThreads), dim3(kThreads), st, part, K,
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
    DEEPIPR_LAUNCH(prof, k_reduce_partials, dim3((C + 3) / 4), dim3(kThreads), st, part,
                       pl.NS, C, dgamma, dbeta);
    return check_launch("affine_relu_bwd(finish)");
}

int deepipr_sign_loss_fwd(const float *gamma, const float *b, float alpha, float margin, float l2, int C,
                          float *loss, float *acc, int8_t *bits, void *stream) {
    if (!gamma || !b || !loss || !acc || C <= 0) return fail(DEEPIPR_EINVAL, "sign_loss_fwd: bad argument");
    hipStream_t st0 = static_cast<hipStream_t>(stream);
    ProfScope prof(DEEPIPR_K_SIGN_LOSS_FWD, st0);
    DEEPIPR_LAUNCH(prof, k_sign_loss_fwd, dim3(1), dim3(kThreads), st0, gamma, b,
                       alpha, margin, l2, C, loss, acc, bits);
    return check_launch("sign_loss_fwd");
}

int deepipr_sign_loss_bwd(const float *dloss, const float *gamma, const float *b, float alpha, float margin,
                          float l2, int C, float *dgamma, void *stream) {
    if (!dloss || !gamma || !b || !dgamma || C <= 0) return fail(DEEPIPR_EINVAL, "sign_loss_bwd: bad argument");
    hipStream_t st0 = static_cast<hipStream_t>(stream);
    ProfScope prof(DEEPIPR_K_SIGN_LOSS_BWD, st0);
    DEEPIPR_LAUNCH(prof, k_sign_loss_bwd, dim3((C + kThreads - 1) / kThreads), dim3(kThreads),
                   st0, dloss, gamma, b, alpha, margin, l2, C, dgamma);
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
    if (!dy || !xhat || !gamma || !beta || !dxhat || !dgamma || !dbeta || !workspace || bad_dims(N, C, HW))
        return fail(DEEPIPR_EINVAL, "passport_bwd: bad argument");
    if (dW && (!s || K <= 0)) return fail(DEEPIPR_EINVAL, "passport_bwd: dW needs m and K");
    if (dloss && !b) return fail(DEEPIPR_EINVAL, "passport_bwd: sign loss needs b");
    hipStream_t st = static_cast<hipStream_t>(stream);
    BwdPlan pl;
    double *part = static_cast<double *>(workspace);
    int rc = launch_affine_bwd(dy, xhat, gamma, beta, dxhat, part, N, C, HW, relu, &pl, st);
    if (rc != DEEPIPR_OK) return rc;
    return launch_passport_finish(part, pl.NS, C, gamma, b, alpha, margin, l2, dloss, dgamma_extra, dbeta_extra, s, K,
                                  dgamma, dbeta, dW, st);
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------
// BatchNorm-fused passport layer
// ---------------------------------------------------------------------------------------------
namespace {

template <int MODE, int VEC, bool RELU>
void launch_walk_t(ProfScope &prof, const float *dy, const float *x, const float *tbl, double *part, int N, int C,
                   int P, const BwdPlan &pl, hipStream_t st) {
    const dim3 grid(pl.tiles, pl.NS);
    if (pl.large)
        DEEPIPR_LAUNCH(prof, (k_bn_walk_large<MODE, VEC, RELU>), grid, dim3(kThreads), st, dy, x, tbl, part, N, C, P, pl);
    else
        DEEPIPR_LAUNCH(prof, (k_bn_walk_small<MODE, VEC, RELU>), grid, dim3(kThreads), st, dy, x, tbl, part, N, C, P, pl);
}

template <int MODE>
int launch_walk(const float *dy, const float *x, const float *tbl, double *part, int N, int C, int P, int relu,
                BwdPlan *plan_out, hipStream_t st) {
    const bool can_vec = aligned16(x) && (MODE == WALK_STATS || aligned16(dy));
    const BwdPlan pl = plan_bwd(N, C, P, can_vec);
    if (pl.NS > 65535) return fail(DEEPIPR_EINVAL, "bn walk: too many batch splits");
    ProfScope prof(MODE == WALK_STATS ? DEEPIPR_K_BN_STATS : DEEPIPR_K_BN_BWD_REDUCE, st);
    prof.bytes = (MODE == WALK_STATS ? 4.0 : 8.0) * static_cast<double>(N) * C * P;
    if (pl.VEC == 4) {
        if (relu) launch_walk_t<MODE, 4, true>(prof, dy, x, tbl, part, N, C, P, pl, st);
        else launch_walk_t<MODE, 4, false>(prof, dy, x, tbl, part, N, C, P, pl, st);
    } else {
        if (relu) launch_walk_t<MODE, 1, true>(prof, dy, x, tbl, part, N, C, P, pl, st);
        else launch_walk_t<MODE, 1, false>(prof, dy, x, tbl, part, N, C, P, pl, st);
    }
    *plan_out = pl;
    return check_launch("bn walk");
}


// ---- register-resident path: planning and launch ----
std::atomic<int> g_resident_mode{1};

int device_cu_count() {
    static std::mutex mu;
    static int cache[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    std::lock_guard<std::mutex> lk(mu);
    if (cache[dev] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 0;
        cache[dev] = n > 0 ? n : -1;
    }
    return cache[dev] > 0 ? cache[dev] : 0;
}

// Planning constants of the single-pass kernels.  Measured per shape with tools/res_tune.py
// (profiles/r02_res_tune_granule_exchange.log):
//  * split_full = 1: with the granule exchange at ~1.2 us, splitting C = 128 layers over 2 workgroups per channel
//    (256 workgroups instead of 128) wins 3-12 %; it lost 0-10 % with the 4 us ticket exchange of round 1;
//  * xcd_map = 0: putting a channel's slices on one XCD helps the plain layers by 2 % (exchange locality) but
//    costs the tail-folded backward 9 % (31.2 -> 34.0 us): one XCD then streams addresses 8 MB apart through its
//    L2 instead of neighbouring channels.  Tails carry most of the bytes, so slices stay on consecutive workgroups;
//  * small_t = 1: 256-thread workgroups for slices of <= 2048 float4.
// The PRODUCTION library has them as constants; the measurement / test build (`make trace`, -DDEEPIPR_TEST_HOOKS)
// turns them into knobs (deepipr_debug_tune) next to the time-out test hooks and the phase stamps.
#ifdef DEEPIPR_TEST_HOOKS
struct ResTune {
    std::atomic<int> split_full{1};   // split channels over slices whenever they do not fill the chip (0: only below half)
    std::atomic<int> xcd_map{0};      // 1: slices of a channel on workgroups with equal blockIdx % 8
    std::atomic<int> small_t{1};      // 256-thread workgroups for slices of <= 2048 float4 (0: always 1024 threads)
    std::atomic<int> spin{static_cast<int>(kSpinLimit)};
    std::atomic<int> drop{-1};        // test hook: slice that never publishes its partial sums
    std::atomic<unsigned long long *> trace{nullptr};   // phase stamps (deepipr_debug_trace)
};
ResTune g_tune;
inline bool tune_split_full() { return g_tune.split_full.load(std::memory_order_relaxed) != 0; }
inline bool tune_small_t() { return g_tune.small_t.load(std::memory_order_relaxed) != 0; }
#else
inline bool tune_split_full() { return true; }
inline bool tune_small_t() { return true; }
#endif

// Can x[N][C][P] (and dy) be held in registers?  max_f4: float4 units a thread of a 1024-thread workgroup may keep.
// (Two 512-thread workgroups per CU, with or without a delayed second cohort, were measured slower on every
// CIFAR-shape layer -- profiles/r02_res_tune_knob_sweep.log -- and are not planned for.)
bool plan_resident(int N, int C, int P, int max_f4, bool can_sync, ResPlan *out) {
    if (g_resident_mode.load(std::memory_order_relaxed) == 0 || P % 4 != 0) return false;
    const int cus = device_cu_count();
    if (cus <= 0) return false;
    const bool split_full = tune_split_full();
    ResPlan pl{};
    pl.q4 = P / 4;
    int G = 1;
    if ((pl.q4 & (pl.q4 - 1)) == 0 && pl.q4 < 8) {           // planes shorter than a 128-byte line
        G = 8 / pl.q4;
        while (G > 1 && C % G != 0) G >>= 1;
    }
    auto slices = [&](int cb) {                  // split channels over slices until the workgroups fill the chip
        int S = 1;
        if (can_sync && (split_full ? cb < cus : cb * 2 < cus))
            while (S < kXchMaxSlices && cb * (S * 2) <= cus && S * 2 <= N) S *= 2;
        return S;
    };
    int S = slices(C / G);
    if (S > 1 && G > 1) {
        G = 1;
        S = slices(C);
    }
    if (S > 1 && C > kXchChannels) S = 1;
    pl.G = G;
    pl.gq = G * pl.q4;
    pl.S = S;
    pl.nps = (N + S - 1) / S;
    pl.blocks = (C / G) * S;
    const long long units = static_cast<long long>(pl.nps) * pl.gq;
    pl.T = (S == 1 && units <= 8 * 256 && tune_small_t()) ? 256 : 1024;
    if (pl.blocks * 4 < cus) return false;                   // too few workgroups to be worth a single pass
    const long long need = (units + pl.T - 1) / pl.T;
    static const int steps[] = {1, 2, 3, 4, 6, 8, 12, 16};
    pl.F4 = 0;
    for (int f : steps)
        if (f >= need && f <= (pl.T == 256 ? 8 : max_f4)) {
            pl.F4 = f;
            break;
        }
    pl.c_off = 0;
    pl.cpp = C;
    pl.passes = 1;
    bool range_mode = false;                     // the slice count comes from the channel-range search (any integer: its own region)
    if (pl.F4 == 0) {
        range_mode = true;
        // The layer does not fit the register file at once (ImageNet-size maps: [128, 64, 112, 112] is 411 MB against
        // 67 MB of registers a forward / 33 MB per tensor a backward launch can hold).  Channel-range PASSES: a launch
        // takes as many channels as fill the chip once each is split over S workgroups, S the smallest power of two for
        // which a slice fits a workgroup's registers; the layer is ceil(C / channels per pass) launches of the same
        // single-pass kernel -- x is still read once and y written once (8 / 12 B per element instead of the 12 / 20 of
        // the three-launch form), at the price of an in-launch exchange among up to 64 partners.
        if (!can_sync || G != 1) return false;
        const int lim = cus < kXchChannels ? cus : kXchChannels;
        bool found = false;
        // (at most 12 float4 per thread: the 16-unit forward instance has no register to spare for the loop over ranges of
        // k_bn_res_fwd_ranges -- 24 spilled -- and a range is a loop iteration now, not a launch)
        const int pass_f4 = max_f4 > 12 ? 12 : max_f4;
        // The slice count that FILLS a workgroup's registers -- any integer from 2 to 64, not only a power of two (round 6: the
        // exchange costs 2.4 - 4.4 us of a range's ~15 whatever the range holds, tools/res_trace.py on ResNet50's maps,
        // profiles/r06k_res_trace_r50.log: 784-float4 planes at 12 units per thread take 15 samples per slice -- 18 slices, 19
        // ranges of 14 channels -- where the next power of two took 8 samples, 32 slices and 32 ranges of 8).  The exchange's
        // slot regions are per slice count (xch_region).
        // Measured on ResNet50's maps at batch 256 (tools/norm_ranges_bench.py, profiles/r06m_norm_slices.jsonl): forward 5.78 ->
        // 5.28 ms with the filling slice count; backward (8 units per thread: the power of two already fills 77 % of them)
        // 7.38 -> 7.71 ms -- it keeps the power of two.  DEEPIPR_BN_POW2_SLICES = 1 / 0 forces one rule for both directions.
        static const int force_pow2 = getenv("DEEPIPR_BN_POW2_SLICES") ? atoi(getenv("DEEPIPR_BN_POW2_SLICES")) : -1;
        const bool pow2_only = force_pow2 >= 0 ? force_pow2 != 0 : max_f4 <= 8;
        const long long cap = static_cast<long long>(pass_f4) * 1024 / pl.q4;       // samples a workgroup can hold
        int Sp = cap >= 1 ? static_cast<int>((N + cap - 1) / cap) : kXchMaxSlices + 1;
        if (Sp < 2) Sp = 2;
        if (pow2_only)
            for (int p2 = 2; ; p2 *= 2)
                if (p2 >= Sp) { Sp = p2; break; }
        if (Sp <= kXchMaxSlices && Sp <= N && Sp <= lim) {
            const long long nps = (N + Sp - 1) / Sp;
            const long long nd = (nps * pl.q4 + 1023) / 1024;
            int f4 = 0;
            for (int f : steps)
                if (f >= nd && f <= pass_f4) {
                    f4 = f;
                    break;
                }
            if (f4) {
                pl.S = Sp;
                pl.nps = static_cast<int>(nps);
                pl.T = 1024;
                pl.F4 = f4;
                pl.cpp = lim / Sp < C ? lim / Sp : C;
                pl.passes = (C + pl.cpp - 1) / pl.cpp;
                pl.blocks = pl.cpp * Sp;
                found = true;
            }
        }
        if (!found) return false;
    }
    pl.gqdiv = make_fastdiv(static_cast<unsigned>(pl.gq));
    pl.xoff = pl.S > 1 ? xch_region(pl.S, range_mode) : 0;
#ifdef DEEPIPR_TEST_HOOKS
    pl.spin = static_cast<unsigned>(g_tune.spin.load(std::memory_order_relaxed));
    pl.drop = g_tune.drop.load(std::memory_order_relaxed);
    pl.xcd_map = (S > 1 && (C / G) % 8 == 0 && g_tune.xcd_map.load(std::memory_order_relaxed)) ? 1 : 0;
    pl.trace = g_tune.trace.load(std::memory_order_relaxed);
#endif
    *out = pl;
    return true;
}

#define DEEPIPR_RES_CASES(KERNEL, TT, ...)                                                                        \
    switch (pl.F4) {                                                                                              \
        case 1: DEEPIPR_LAUNCH(prof, (KERNEL<TT, 1>), grid, dim3(TT), st, __VA_ARGS__); break;                    \
        case 2: DEEPIPR_LAUNCH(prof, (KERNEL<TT, 2>), grid, dim3(TT), st, __VA_ARGS__); break;                    \
        case 3: DEEPIPR_LAUNCH(prof, (KERNEL<TT, 3>), grid, dim3(TT), st, __VA_ARGS__); break;                    \
        case 4: DEEPIPR_LAUNCH(prof, (KERNEL<TT, 4>), grid, dim3(TT), st, __VA_ARGS__); break;                    \
        case 6: DEEPIPR_LAUNCH(prof, (KERNEL<TT, 6>), grid, dim3(TT), st, __VA_ARGS__); break;                    \
        default: DEEPIPR_LAUNCH(prof, (KERNEL<TT, 8>), grid, dim3(TT), st, __VA_ARGS__); break;                   \
    }

int launch_res_fwd(const float *x, float *y, const float *gamma, const float *beta, int relu, int N, int C,
                   const ResPlan &pl, const BnFinishArgs &f, double *part, unsigned *sync, bool with_sign,
                   const SignArgs &sa, const float *residual, hipStream_t st) {
    ProfScope prof(DEEPIPR_K_BN_RES_FWD, st);
    prof.bytes = (residual ? 12.0 : 8.0) * static_cast<double>(N) * (pl.blocks / pl.S * pl.G) * pl.q4 * 4;
    const float4 *r4 = reinterpret_cast<const float4 *>(residual);
    const dim3 grid(pl.blocks + (with_sign ? 1 : 0));
    const float4 *x4 = reinterpret_cast<const float4 *>(x);
    float4 *y4 = reinterpret_cast<float4 *>(y);
    const int ws = with_sign ? 1 : 0;
    if (pl.T == 256) {
        DEEPIPR_RES_CASES(k_bn_res_fwd, 256, x4, y4, gamma, beta, relu, N, C, pl, f, part, sync, ws, sa, r4)
    } else if (pl.F4 == 12) {
        DEEPIPR_LAUNCH(prof, (k_bn_res_fwd<1024, 12>), grid, dim3(1024), st, x4, y4, gamma, beta, relu, N, C, pl, f, part, sync, ws, sa, r4);
    } else if (pl.F4 == 16) {
        DEEPIPR_LAUNCH(prof, (k_bn_res_fwd<1024, 16>), grid, dim3(1024), st, x4, y4, gamma, beta, relu, N, C, pl, f, part, sync, ws, sa, r4);
    } else {
        DEEPIPR_RES_CASES(k_bn_res_fwd, 1024, x4, y4, gamma, beta, relu, N, C, pl, f, part, sync, ws, sa, r4)
    }
    return check_launch("passport_bn_fwd(resident)");
}

// late start of the odd channel groups in units of s_sleep(127) (~3.9 us at 2.1 GHz); DEEPIPR_BN_STAGGER="fwd,bwd" overrides
int ranges_stagger(int backward) {
    static int v[2] = {-1, -1};
    if (v[0] < 0) {
        v[0] = 2;           // measured on ResNet50's maps at batch 256 (tools/norm_ranges_sweep.sh, profiles/r06e_norm_ranges_sweep.jsonl):
        v[1] = 2;           // forward 6.02 -> 5.84 ms, backward 8.13 -> 7.65 ms against no stagger; 3 and more lose again
        if (const char *e = getenv("DEEPIPR_BN_STAGGER")) {
            int a = 0, b = 0;
            const int n = sscanf(e, "%d,%d", &a, &b);
            if (n >= 1 && a >= 0 && a <= 64) v[0] = a;
            v[1] = (n >= 2 && b >= 0 && b <= 64) ? b : v[0];
        }
    }
    return v[backward];
}

// all `pl.passes` channel ranges of the layer in one launch (k_bn_res_fwd_ranges); false: no instance (the caller loops)
bool launch_res_fwd_ranges(const float *x, float *y, const float *gamma, const float *beta, int relu, int N, int C,
                           const ResPlan &pl, const BnFinishArgs &f, unsigned *sync, const float *residual, hipStream_t st) {
    if (pl.passes < 2 || pl.T != 1024 || pl.G != 1 || pl.S < 2 || 2 * pl.cpp * pl.S > 512 || (pl.F4 != 12 && pl.F4 != 8 && pl.F4 != 6)) return false;
    ProfScope prof(DEEPIPR_K_BN_RES_FWD, st);
    prof.bytes = (residual ? 12.0 : 8.0) * static_cast<double>(N) * C * pl.q4 * 4;
    const dim3 grid(pl.cpp * pl.S);
    const float4 *x4 = reinterpret_cast<const float4 *>(x), *r4 = reinterpret_cast<const float4 *>(residual);
    float4 *y4 = reinterpret_cast<float4 *>(y);
    ResPlan q = pl;
    q.stagger = ranges_stagger(0);
    if (pl.F4 == 12) DEEPIPR_LAUNCH(prof, (k_bn_res_fwd_ranges<1024, 12>), grid, dim3(1024), st, x4, y4, gamma, beta, relu, N, C, q, f, sync, r4);
    else if (pl.F4 == 8) DEEPIPR_LAUNCH(prof, (k_bn_res_fwd_ranges<1024, 8>), grid, dim3(1024), st, x4, y4, gamma, beta, relu, N, C, q, f, sync, r4);
    else DEEPIPR_LAUNCH(prof, (k_bn_res_fwd_ranges<1024, 6>), grid, dim3(1024), st, x4, y4, gamma, beta, relu, N, C, q, f, sync, r4);
    return true;
}

int launch_res_bwd(const float *dy, const float *x, const float *tbl, float *dx, int relu, int N, int C,
                   const ResPlan &pl, double *part, unsigned *sync, const ResBwdArgs &a, hipStream_t st) {
    ProfScope prof(DEEPIPR_K_BN_RES_BWD, st);
    prof.bytes = (a.tail_out ? (a.dy2 ? 24.0 : 20.0) : (a.dy2 ? 16.0 : 12.0)) * static_cast<double>(N) *
                 (pl.blocks / pl.S * pl.G) * pl.q4 * 4;
    const dim3 grid(pl.blocks);
    const float4 *d4 = reinterpret_cast<const float4 *>(dy), *x4 = reinterpret_cast<const float4 *>(x);
    float4 *o4 = reinterpret_cast<float4 *>(dx);
    if (pl.T == 256) {
        DEEPIPR_RES_CASES(k_bn_res_bwd, 256, d4, x4, tbl, o4, relu, N, C, pl, part, sync, a)
    } else {
        DEEPIPR_RES_CASES(k_bn_res_bwd, 1024, d4, x4, tbl, o4, relu, N, C, pl, part, sync, a)
    }
    return check_launch("passport_bn_bwd(resident)");
}

bool launch_res_bwd_ranges(const float *dy, const float *x, const float *tbl, float *dx, int relu, int N, int C,
                           const ResPlan &pl, unsigned *sync, const ResBwdArgs &a, hipStream_t st) {
    if (pl.passes < 2 || pl.T != 1024 || pl.G != 1 || pl.S < 2 || 2 * pl.cpp * pl.S > 512 || (pl.F4 != 6 && pl.F4 != 8 && pl.F4 != 4)) return false;
    ProfScope prof(DEEPIPR_K_BN_RES_BWD, st);
    prof.bytes = (a.tail_out ? (a.dy2 ? 24.0 : 20.0) : (a.dy2 ? 16.0 : 12.0)) * static_cast<double>(N) * C * pl.q4 * 4;
    const dim3 grid(pl.cpp * pl.S);
    const float4 *d4 = reinterpret_cast<const float4 *>(dy), *x4 = reinterpret_cast<const float4 *>(x);
    float4 *o4 = reinterpret_cast<float4 *>(dx);
    ResPlan q = pl;
    q.stagger = ranges_stagger(1);
    if (pl.F4 == 6) DEEPIPR_LAUNCH(prof, (k_bn_res_bwd_ranges<1024, 6>), grid, dim3(1024), st, d4, x4, tbl, o4, relu, N, C, q, sync, a);
    else if (pl.F4 == 8) DEEPIPR_LAUNCH(prof, (k_bn_res_bwd_ranges<1024, 8>), grid, dim3(1024), st, d4, x4, tbl, o4, relu, N, C, q, sync, a);
    else DEEPIPR_LAUNCH(prof, (k_bn_res_bwd_ranges<1024, 4>), grid, dim3(1024), st, d4, x4, tbl, o4, relu, N, C, q, sync, a);
    return true;
}


// ---- GroupNorm / InstanceNorm fused layer: planning and launch ----
bool plan_gn(int N, int C, int P, int groups, GnPlan *out) {
    if (groups <= 0 || C % groups != 0 || P % 4 != 0) return false;
    GnPlan pl{};
    pl.cpg = C / groups;
    pl.q4 = P / 4;
    const long long U = static_cast<long long>(pl.cpg) * pl.q4;
    if (U > 6144) return false;                    // registers (F4 <= 8 at 1024 lanes) and 48 KB of LDS in backward
    pl.U = static_cast<int>(U);
    // lanes per chunk: ~4 float4 in flight per lane (memory-level parallelism, cheap group reductions), but never
    // fewer than 16 lanes (256-byte runs per load instruction) unless the chunk itself is shorter
    auto pow2ceil = [](int v) { int p = 1; while (p < v) p <<= 1; return p; };
    int tg = pow2ceil((pl.U + 3) / 4);
    const int floor_tg = pow2ceil(pl.U < 16 ? pl.U : 16);
    if (tg < floor_tg) tg = floor_tg;
    if (tg > 1024) tg = 1024;
    pl.TG = tg;
    const int need = (pl.U + tg - 1) / tg;
    pl.F4 = need <= 1 ? 1 : need <= 2 ? 2 : need <= 4 ? 4 : 8;
    pl.T = tg < kThreads ? kThreads : tg;
    pl.groups = groups;
    pl.chunks = N * groups;
    pl.q4div = make_fastdiv(static_cast<unsigned>(pl.q4));
    *out = pl;
    return true;
}

#define DEEPIPR_GN_CASES(KERNEL, LDS, ...)                                                                          \
    switch (pl.F4) {                                                                                               \
        case 1: hipLaunchKernelGGL((KERNEL<1>), grid, dim3(pl.T), LDS, st, __VA_ARGS__); break;                     \
        case 2: hipLaunchKernelGGL((KERNEL<2>), grid, dim3(pl.T), LDS, st, __VA_ARGS__); break;                     \
        case 4: hipLaunchKernelGGL((KERNEL<4>), grid, dim3(pl.T), LDS, st, __VA_ARGS__); break;                     \
        default: hipLaunchKernelGGL((KERNEL<8>), grid, dim3(pl.T), LDS, st, __VA_ARGS__); break;                    \
    }

}  // namespace

extern "C" {

size_t deepipr_passport_bn_workspace_bytes(int N, int C, int HW) {
    if (bad_dims(N, C, HW)) return 0;
    const size_t two_pass = bwd_workspace_bytes(N, C, HW);
    const size_t resident = static_cast<size_t>(64) * 2 * C * sizeof(double);      // up to 64 batch slices
    return two_pass > resident ? two_pass : resident;
}

int deepipr_passport_bn_resident(int N, int C, int HW, int have_sync) {
    if (bad_dims(N, C, HW)) return 0;
    ResPlan rp;
    int mask = 0;
    if (plan_resident(N, C, HW, 16, have_sync != 0, &rp)) mask |= 1;
    if (plan_resident(N, C, HW, 8, have_sync != 0, &rp)) mask |= 2;
    return mask;
}

// ---- a projection block's last two norm layers + tail in one launch per direction (k_bn_dual_fwd / _bwd) ----
namespace {
#define DEEPIPR_DUAL_CASES_4(KERNEL, TT, ...)                                                                     \
    switch (pl.F4) {                                                                                              \
        case 1: DEEPIPR_LAUNCH(prof, (KERNEL<TT, 1>), grid, dim3(TT), st, __VA_ARGS__); break;                    \
        case 2: DEEPIPR_LAUNCH(prof, (KERNEL<TT, 2>), grid, dim3(TT), st, __VA_ARGS__); break;                    \
        case 3: DEEPIPR_LAUNCH(prof, (KERNEL<TT, 3>), grid, dim3(TT), st, __VA_ARGS__); break;                    \
        default: DEEPIPR_LAUNCH(prof, (KERNEL<TT, 4>), grid, dim3(TT), st, __VA_ARGS__); break;                   \
    }

// The plan both directions of the dual form share -- and the one the separate single-pass launches of the two layers
// would take (same T, S, F4: that is what makes the dual form bit-identical to them).  Backward keeps three register
// arrays per unit, so a 1024-thread workgroup is limited to 4 float4 per thread (a 256-thread one may use 8: one
// wave per SIMD has the whole register file).
static bool plan_dual(int N, int C, int HW, bool can_sync, ResPlan *out) {
    ResPlan rp;
    if (!plan_resident(N, C, HW, 8, can_sync, &rp) || rp.passes != 1) return false;
    if (rp.T == 1024 && rp.F4 > 4) return false;
    if (rp.S > 1 && 2 * C > kXchChannels) return false;     // the second layer's exchange slots: cb + C
    *out = rp;
    return true;
}
}  // namespace

int deepipr_bn_dual_tail_supported(int N, int C, int HW, int have_sync) {
    if (bad_dims(N, C, HW)) return 0;
    ResPlan rp;
    return plan_dual(N, C, HW, have_sync != 0, &rp) ? 1 : 0;
}

int deepipr_bn_dual_tail_fwd(const float *xa, const float *xb, const float *gamma_a, const float *beta_a,
                             const float *gamma_b, const float *beta_b, float *running_mean_a, float *running_var_a,
                             long long *num_batches_tracked_a, float *running_mean_b, float *running_var_b,
                             long long *num_batches_tracked_b, float momentum_a, float momentum_b, float eps_a,
                             float eps_b, int relu_a, int relu_b, int N, int C, int HW, float *out, float *table_a,
                             float *table_b, unsigned int *sync, void *stream) {
    if (!xa || !xb || !gamma_a || !beta_a || !gamma_b || !beta_b || !out || !table_a || !table_b || bad_dims(N, C, HW))
        return fail(DEEPIPR_EINVAL, "bn_dual_tail_fwd: bad argument");
    if ((running_mean_a == nullptr) != (running_var_a == nullptr) || (running_mean_b == nullptr) != (running_var_b == nullptr))
        return fail(DEEPIPR_EINVAL, "bn_dual_tail_fwd: running_mean and running_var go together");
    if (!aligned16(xa) || !aligned16(xb) || !aligned16(out))
        return fail(DEEPIPR_EINVAL, "bn_dual_tail_fwd: tensors must be 16-byte aligned");
    ResPlan pl;
    if (!plan_dual(N, C, HW, sync != nullptr, &pl))
        return fail(DEEPIPR_EUNSUPPORTED, "bn_dual_tail_fwd: shape outside the dual form (ask deepipr_bn_dual_tail_supported)");
    hipStream_t st = static_cast<hipStream_t>(stream);
    BnFinishArgs f{};
    const double M = static_cast<double>(N) * HW;
    f.inv_m = 1.0 / M;
    f.unbias = M > 1.0 ? M / (M - 1.0) : 1.0;
    f.eps = eps_a;
    f.momentum = momentum_a;
    f.running_mean = running_mean_a;
    f.running_var = running_var_a;
    f.num_batches_tracked = num_batches_tracked_a;
    f.tbl = table_a;
    f.shift_src = xa;
    f.HW = HW;
    DualFwdArgs d{reinterpret_cast<const float4 *>(xb), gamma_b, beta_b, momentum_b, eps_b, running_mean_b,
                  running_var_b, num_batches_tracked_b, table_b};
    ProfScope prof(DEEPIPR_K_BN_RES_FWD, st);
    prof.bytes = 12.0 * static_cast<double>(N) * C * HW;
    const dim3 grid(pl.blocks);
    const float4 *x4 = reinterpret_cast<const float4 *>(xa);
    float4 *y4 = reinterpret_cast<float4 *>(out);
    if (pl.T == 256) {
        DEEPIPR_RES_CASES(k_bn_dual_fwd, 256, x4, y4, gamma_a, beta_a, relu_a, relu_b, N, C, pl, f, d, sync)
    } else {
        DEEPIPR_DUAL_CASES_4(k_bn_dual_fwd, 1024, x4, y4, gamma_a, beta_a, relu_a, relu_b, N, C, pl, f, d, sync)
    }
    return check_launch("bn_dual_tail_fwd");
}

int deepipr_bn_dual_tail_bwd(const float *dy, const float *dy2, const float *out, const float *xa, const float *xb,
                             const float *table_a, const float *table_b, float *dxa, float *dxb, float *dgamma_a,
                             float *dbeta_a, float *dgamma_b, float *dbeta_b, int relu_a, int relu_b, int N, int C,
                             int HW, unsigned int *sync, void *stream) {
    if (!dy || !out || !xa || !xb || !table_a || !table_b || !dxa || !dxb || !dgamma_a || !dbeta_a || !dgamma_b ||
        !dbeta_b || bad_dims(N, C, HW))
        return fail(DEEPIPR_EINVAL, "bn_dual_tail_bwd: bad argument");
    if (!aligned16(dy) || !aligned16(out) || !aligned16(xa) || !aligned16(xb) || !aligned16(dxa) || !aligned16(dxb) ||
        (dy2 && !aligned16(dy2)))
        return fail(DEEPIPR_EINVAL, "bn_dual_tail_bwd: tensors must be 16-byte aligned");
    ResPlan pl;
    if (!plan_dual(N, C, HW, sync != nullptr, &pl))
        return fail(DEEPIPR_EUNSUPPORTED, "bn_dual_tail_bwd: shape outside the dual form (ask deepipr_bn_dual_tail_supported)");
    hipStream_t st = static_cast<hipStream_t>(stream);
    DualBwdArgs a{reinterpret_cast<const float4 *>(dy2), reinterpret_cast<const float4 *>(out),
                  reinterpret_cast<const float4 *>(xb), table_b, reinterpret_cast<float4 *>(dxb), dgamma_a, dbeta_a,
                  dgamma_b, dbeta_b, 1.0 / (static_cast<double>(N) * HW)};
    ProfScope prof(DEEPIPR_K_BN_RES_BWD, st);
    prof.bytes = (dy2 ? 28.0 : 24.0) * static_cast<double>(N) * C * HW;
    const dim3 grid(pl.blocks);
    const float4 *d4 = reinterpret_cast<const float4 *>(dy), *x4 = reinterpret_cast<const float4 *>(xa);
    float4 *o4 = reinterpret_cast<float4 *>(dxa);
    if (pl.T == 256) {
        DEEPIPR_RES_CASES(k_bn_dual_bwd, 256, d4, x4, table_a, o4, relu_a, relu_b, N, C, pl, sync, a)
    } else {
        DEEPIPR_DUAL_CASES_4(k_bn_dual_bwd, 1024, d4, x4, table_a, o4, relu_a, relu_b, N, C, pl, sync, a)
    }
    return check_launch("bn_dual_tail_bwd");
}

int deepipr_passport_bn_passes(int N, int C, int HW, int backward) {
    if (bad_dims(N, C, HW)) return 0;
    ResPlan rp;
    return plan_resident(N, C, HW, backward ? 8 : 16, true, &rp) ? rp.passes : 0;
}

int deepipr_passport_bn_slices(int N, int C, int HW) {
    if (bad_dims(N, C, HW)) return 1;
    ResPlan rp;
    int S = 1;
    if (plan_resident(N, C, HW, 16, true, &rp) && rp.S > S) S = rp.S;
    if (plan_resident(N, C, HW, 8, true, &rp) && rp.S > S) S = rp.S;
    return S;
}

#ifdef DEEPIPR_TEST_HOOKS
int deepipr_debug_tune(const char *key, int value) {
    if (!key) return fail(DEEPIPR_EINVAL, "debug_tune: null key");
    const std::string k(key);
    if (k == "split_full") g_tune.split_full.store(value != 0);
    else if (k == "xcd_map") g_tune.xcd_map.store(value != 0);
    else if (k == "small_t") g_tune.small_t.store(value != 0);
    else if (k == "exchange_spin") g_tune.spin.store(value <= 0 ? static_cast<int>(kSpinLimit) : value);
    else if (k == "exchange_drop") g_tune.drop.store(value);
    else return fail(DEEPIPR_EINVAL, "debug_tune: unknown key '%s'", key);
    return DEEPIPR_OK;
}

int deepipr_debug_trace(unsigned long long *device_buffer) {
    g_tune.trace.store(device_buffer);
    return DEEPIPR_OK;
}
#endif

int deepipr_set_resident(int mode) {
    if (mode != 0 && mode != 1) return fail(DEEPIPR_EINVAL, "set_resident: mode must be 0 or 1");
    g_resident_mode.store(mode);
    return DEEPIPR_OK;
}

int deepipr_passport_bn_fwd(const float *x, const float *W, const double *m, const float *gamma_in,
                            const float *beta_in, const float *b, float alpha, float margin, float l2,
                            float *running_mean, float *running_var, long long *num_batches_tracked,
                            float momentum, float eps, int training, int N, int C, int HW, int K, int relu,
                            float *y, float *table, float *gamma, float *beta, float *loss, float *acc,
                            int8_t *bits, const float *residual, void *workspace, unsigned int *sync,
                            void *stream) {
    if (!x || !y || !table || bad_dims(N, C, HW)) return fail(DEEPIPR_EINVAL, "passport_bn_fwd: bad argument");
    if (residual && !aligned16(residual)) return fail(DEEPIPR_EINVAL, "passport_bn_fwd: residual must be 16-byte aligned");
    if (W && (!m || !gamma || !beta || K <= 0)) return fail(DEEPIPR_EINVAL, "passport_bn_fwd: W needs m, gamma, beta, K");
    if (!W && (!gamma_in || !beta_in)) return fail(DEEPIPR_EINVAL, "passport_bn_fwd: need W or gamma_in/beta_in");
    if (training && !workspace) return fail(DEEPIPR_EINVAL, "passport_bn_fwd: training needs a workspace");
    if (!training && (!running_mean || !running_var))
        return fail(DEEPIPR_EINVAL, "passport_bn_fwd: evaluation needs running statistics");
    if ((running_mean == nullptr) != (running_var == nullptr))
        return fail(DEEPIPR_EINVAL, "passport_bn_fwd: running_mean and running_var go together");
    const bool with_sign = loss != nullptr;
    if (with_sign && (!b || !acc)) return fail(DEEPIPR_EINVAL, "passport_bn_fwd: sign loss needs b and acc");
    const size_t total = static_cast<size_t>(N) * C * HW;
    if (total >= (1ull << 31)) return fail(DEEPIPR_EINVAL, "passport_bn_fwd: tensor has >= 2^31 elements");
    hipStream_t st = static_cast<hipStream_t>(stream);
    double *part = static_cast<double *>(workspace);
    BnFinishArgs f{};
    const double M = static_cast<double>(N) * HW;
    f.inv_m = 1.0 / M;
    f.unbias = M > 1.0 ? M / (M - 1.0) : 1.0;
    f.eps = eps;
    f.momentum = momentum;
    f.running_mean = running_mean;
    f.running_var = running_var;
    f.num_batches_tracked = num_batches_tracked;
    f.tbl = table;
    f.shift_src = x;
    f.HW = HW;
    ResPlan rp;
    if (training && aligned16(x) && aligned16(y) && plan_resident(N, C, HW, 16, sync != nullptr, &rp)) {
        // single pass: x stays in registers between the statistics and the normalise/affine/ReLU phase
        const float *g = gamma_in, *bt = beta_in;
        if (W) {
            int rc = deepipr_gamma_beta_fwd(W, m, C, K, gamma, beta, stream);
            if (rc != DEEPIPR_OK) return rc;
            g = gamma;
            bt = beta;
        }
        f.part = part;
        f.NS = rp.S;
        SignArgs sa{b, alpha, margin, l2, loss, acc, bits};
        static const bool one_launch = !(getenv("DEEPIPR_BN_RANGES") && !atoi(getenv("DEEPIPR_BN_RANGES")));      // A/B knob
        if (one_launch && rp.passes > 1 && !with_sign &&
            launch_res_fwd_ranges(x, y, g, bt, relu, N, C, rp, f, sync, residual, st))
            return check_launch("passport_bn_fwd(resident, ranges)");
        for (int p = 0; p < rp.passes; ++p) {              // one launch, or channel-range passes of a large map
            ResPlan q = rp;
            q.c_off = p * rp.cpp;
            const int cn = C - q.c_off < rp.cpp ? C - q.c_off : rp.cpp;
            q.blocks = (cn / q.G) * q.S;
            BnFinishArgs fp = f;
            if (p) fp.num_batches_tracked = nullptr;       // counted once per call
            int rc = launch_res_fwd(x, y, g, bt, relu, N, C, q, fp, part, sync, with_sign && p == 0, sa, residual, st);
            if (rc != DEEPIPR_OK) return rc;
        }
        return DEEPIPR_OK;
    }
    if (residual)
        return fail(DEEPIPR_EUNSUPPORTED, "passport_bn_fwd: the fused residual tail needs the single-pass form "
                                          "(ask deepipr_passport_bn_resident first)");
    if (training) {
        BwdPlan pl;
        int rc = launch_walk<WALK_STATS>(nullptr, x, nullptr, part, N, C, HW, relu, &pl, st);
        if (rc != DEEPIPR_OK) return rc;
        f.part = part;
        f.NS = pl.NS;
    }
    const float *g_for_sign = gamma_in;
    {
        ProfScope prof(DEEPIPR_K_GAMMA_BETA_FWD, st);
        if (W) {
            const bool vec = K % 4 == 0 && aligned16(W) && aligned16(m);
            if (C >= 2 * kRowPairMinCo) {
                const dim3 grid((C + 1) / 2);
                if (vec) DEEPIPR_LAUNCH(prof, (k_gamma_beta_bn<true, 2>), grid, dim3(kThreads), st, W, m, C, K, gamma, beta, f);
                else DEEPIPR_LAUNCH(prof, (k_gamma_beta_bn<false, 2>), grid, dim3(kThreads), st, W, m, C, K, gamma, beta, f);
            } else {
                if (vec) DEEPIPR_LAUNCH(prof, (k_gamma_beta_bn<true, 1>), dim3(C), dim3(kThreads), st, W, m, C, K, gamma, beta, f);
                else DEEPIPR_LAUNCH(prof, (k_gamma_beta_bn<false, 1>), dim3(C), dim3(kThreads), st, W, m, C, K, gamma, beta, f);
            }
            g_for_sign = gamma;
        } else {
            DEEPIPR_LAUNCH(prof, k_bn_table, dim3((C + 3) / 4), dim3(kThreads), st, gamma_in, beta_in, C, f);
        }
        int rc = check_launch("passport_bn_fwd(finish)");
        if (rc != DEEPIPR_OK) return rc;
    }
    SignArgs sa{b, alpha, margin, l2, loss, acc, bits};
    const FastDiv cdiv = make_fastdiv(static_cast<unsigned>(C));
    ProfScope prof(DEEPIPR_K_BN_AFFINE_FWD, st);
    prof.bytes = 8.0 * static_cast<double>(total);
    if (HW % 4 == 0 && aligned16(x) && aligned16(y)) {
        const unsigned n4 = static_cast<unsigned>(total / 4);
        const FastDiv pdiv = make_fastdiv(static_cast<unsigned>(HW / 4));
        const int grid = grid_for(n4) + (with_sign ? 1 : 0);
        if (relu)
            DEEPIPR_LAUNCH(prof, k_bn_affine_fwd_v4<true>, dim3(grid), dim3(kThreads), st, reinterpret_cast<const float4 *>(x), table, reinterpret_cast<float4 *>(y), n4, pdiv,
                               cdiv, static_cast<unsigned>(C), with_sign ? 1 : 0, sa, g_for_sign);
        else
            DEEPIPR_LAUNCH(prof, k_bn_affine_fwd_v4<false>, dim3(grid), dim3(kThreads), st, reinterpret_cast<const float4 *>(x), table, reinterpret_cast<float4 *>(y), n4, pdiv,
                               cdiv, static_cast<unsigned>(C), with_sign ? 1 : 0, sa, g_for_sign);
    } else if (total % 4 == 0 && aligned16(x) && aligned16(y)) {
        const unsigned n4 = static_cast<unsigned>(total / 4);
        const FastDiv pdiv = make_fastdiv(static_cast<unsigned>(HW));
        const int grid = grid_for(n4) + (with_sign ? 1 : 0);
        if (relu)
            DEEPIPR_LAUNCH(prof, k_bn_affine_fwd_v4g<true>, dim3(grid), dim3(kThreads), st,
                           reinterpret_cast<const float4 *>(x), table, reinterpret_cast<float4 *>(y), n4, pdiv, cdiv,
                           static_cast<unsigned>(C), with_sign ? 1 : 0, sa, g_for_sign);
        else
            DEEPIPR_LAUNCH(prof, k_bn_affine_fwd_v4g<false>, dim3(grid), dim3(kThreads), st,
                           reinterpret_cast<const float4 *>(x), table, reinterpret_cast<float4 *>(y), n4, pdiv, cdiv,
                           static_cast<unsigned>(C), with_sign ? 1 : 0, sa, g_for_sign);
    } else {
        const FastDiv pdiv = make_fastdiv(static_cast<unsigned>(HW));
        const int grid = grid_for(total) + (with_sign ? 1 : 0);
        if (relu)
            DEEPIPR_LAUNCH(prof, k_bn_affine_fwd_s<true>, dim3(grid), dim3(kThreads), st, x, table, y,
                               static_cast<unsigned>(total), pdiv, cdiv, static_cast<unsigned>(C),
                               with_sign ? 1 : 0, sa, g_for_sign);
        else
            DEEPIPR_LAUNCH(prof, k_bn_affine_fwd_s<false>, dim3(grid), dim3(kThreads), st, x, table, y,
                               static_cast<unsigned>(total), pdiv, cdiv, static_cast<unsigned>(C),
                               with_sign ? 1 : 0, sa, g_for_sign);
    }
    return check_launch("passport_bn_fwd(apply)");
}

int deepipr_passport_bn_bwd(const float *dy, const float *x, const float *table, const double *m, const float *b,
                            float alpha, float margin, float l2, const float *dloss, const float *dgamma_extra,
                            const float *dbeta_extra, int training, int N, int C, int HW, int K, int relu,
                            float *dx, float *dW, float *dgamma, float *dbeta, float *table_out, void *workspace,
                            unsigned int *sync, const float *dy2, const float *tail_out, float *dres,
                            void *stream) {
    if ((tail_out != nullptr) != (dres != nullptr))
        return fail(DEEPIPR_EINVAL, "passport_bn_bwd: tail_out and dres go together");
    if ((tail_out && !aligned16(tail_out)) || (dres && !aligned16(dres)) || (dy2 && !aligned16(dy2)))
        return fail(DEEPIPR_EINVAL, "passport_bn_bwd: tail pointers must be 16-byte aligned");
    if (!dy || !x || !table || !dx || !dgamma || !dbeta || !table_out || !workspace || bad_dims(N, C, HW))
        return fail(DEEPIPR_EINVAL, "passport_bn_bwd: bad argument");
    if (dW && (!m || K <= 0)) return fail(DEEPIPR_EINVAL, "passport_bn_bwd: dW needs m and K");
    if (dloss && !b) return fail(DEEPIPR_EINVAL, "passport_bn_bwd: sign loss needs b");
    const size_t total = static_cast<size_t>(N) * C * HW;
    if (total >= (1ull << 31)) return fail(DEEPIPR_EINVAL, "passport_bn_bwd: tensor has >= 2^31 elements");
    hipStream_t st = static_cast<hipStream_t>(stream);
    double *part = static_cast<double *>(workspace);
    ResPlan rp;
    if (aligned16(x) && aligned16(dy) && aligned16(dx) && plan_resident(N, C, HW, 8, sync != nullptr, &rp)) {
        // single pass: dz and xhat stay in registers between the two channel sums and the dx phase
        ResBwdArgs a{b, alpha, margin, l2, dloss, dgamma_extra, dbeta_extra, dgamma, dbeta,
                     training ? 1.0 / (static_cast<double>(N) * HW) : 0.0,
                     reinterpret_cast<const float4 *>(dy2), reinterpret_cast<const float4 *>(tail_out),
                     reinterpret_cast<float4 *>(dres)};
        static const bool one_launch = !(getenv("DEEPIPR_BN_RANGES") && !atoi(getenv("DEEPIPR_BN_RANGES")));
        if (one_launch && rp.passes > 1 && launch_res_bwd_ranges(dy, x, table, dx, relu, N, C, rp, sync, a, st)) {
            const int rc = check_launch("passport_bn_bwd(resident, ranges)");
            if (rc != DEEPIPR_OK) return rc;
            return dW ? deepipr_gamma_beta_bwd(dgamma, dbeta, m, C, K, dW, stream) : DEEPIPR_OK;
        }
        for (int p = 0; p < rp.passes; ++p) {
            ResPlan q = rp;
            q.c_off = p * rp.cpp;
            const int cn = C - q.c_off < rp.cpp ? C - q.c_off : rp.cpp;
            q.blocks = (cn / q.G) * q.S;
            int rc = launch_res_bwd(dy, x, table, dx, relu, N, C, q, part, sync, a, st);
            if (rc != DEEPIPR_OK) return rc;
        }
        return dW ? deepipr_gamma_beta_bwd(dgamma, dbeta, m, C, K, dW, stream) : DEEPIPR_OK;
    }
    if (tail_out || dy2)
        return fail(DEEPIPR_EUNSUPPORTED, "passport_bn_bwd: the fused residual tail / a second gradient (dy2) needs the "
                                          "single-pass form (ask deepipr_passport_bn_resident first)");
    BwdPlan pl;
    int rc = launch_walk<WALK_BN_BWD>(dy, x, table, part, N, C, HW, relu, &pl, st);
    if (rc != DEEPIPR_OK) return rc;
    BnBwdFinishArgs f{table, table_out, training ? 1.0 / (static_cast<double>(N) * HW) : 0.0};
    {
        ProfScope prof(DEEPIPR_K_PASSPORT_BWD_FINISH, st);
        const bool vec = dW && K % 4 == 0 && aligned16(dW) && aligned16(m);
#define DEEPIPR_BNFIN_ARGS part, pl.NS, C, b, alpha, margin, l2, dloss, dgamma_extra, dbeta_extra, m, K, dgamma, dbeta, dW, f
        if (C >= 2 * kRowPairMinCo) {
            const dim3 grid((C + 1) / 2);
            if (vec) DEEPIPR_LAUNCH(prof, (k_passport_bn_bwd_finish<true, 2>), grid, dim3(kThreads), st, DEEPIPR_BNFIN_ARGS);
            else DEEPIPR_LAUNCH(prof, (k_passport_bn_bwd_finish<false, 2>), grid, dim3(kThreads), st, DEEPIPR_BNFIN_ARGS);
        } else {
            if (vec) DEEPIPR_LAUNCH(prof, (k_passport_bn_bwd_finish<true, 1>), dim3(C), dim3(kThreads), st, DEEPIPR_BNFIN_ARGS);
            else DEEPIPR_LAUNCH(prof, (k_passport_bn_bwd_finish<false, 1>), dim3(C), dim3(kThreads), st, DEEPIPR_BNFIN_ARGS);
        }
#undef DEEPIPR_BNFIN_ARGS
        rc = check_launch("passport_bn_bwd(finish)");
        if (rc != DEEPIPR_OK) return rc;
    }
    const FastDiv cdiv = make_fastdiv(static_cast<unsigned>(C));
    ProfScope prof(DEEPIPR_K_BN_AFFINE_BWD, st);
    prof.bytes = 12.0 * static_cast<double>(total);
    if (HW % 4 == 0 && aligned16(x) && aligned16(dy) && aligned16(dx)) {
        const unsigned n4 = static_cast<unsigned>(total / 4);
        const FastDiv pdiv = make_fastdiv(static_cast<unsigned>(HW / 4));
        const int grid = grid_for(n4);
        if (relu)
            DEEPIPR_LAUNCH(prof, k_bn_affine_bwd_v4<true>, dim3(grid), dim3(kThreads), st, reinterpret_cast<const float4 *>(dy), reinterpret_cast<const float4 *>(x), table_out,
                               reinterpret_cast<float4 *>(dx), n4, pdiv, cdiv, static_cast<unsigned>(C));
        else
            DEEPIPR_LAUNCH(prof, k_bn_affine_bwd_v4<false>, dim3(grid), dim3(kThreads), st, reinterpret_cast<const float4 *>(dy), reinterpret_cast<const float4 *>(x), table_out,
                               reinterpret_cast<float4 *>(dx), n4, pdiv, cdiv, static_cast<unsigned>(C));
    } else if (total % 4 == 0 && aligned16(x) && aligned16(dy) && aligned16(dx)) {
        const unsigned n4 = static_cast<unsigned>(total / 4);
        const FastDiv pdiv = make_fastdiv(static_cast<unsigned>(HW));
        const int grid = grid_for(n4);
        if (relu)
            DEEPIPR_LAUNCH(prof, k_bn_affine_bwd_v4g<true>, dim3(grid), dim3(kThreads), st,
                           reinterpret_cast<const float4 *>(dy), reinterpret_cast<const float4 *>(x), table_out,
                           reinterpret_cast<float4 *>(dx), n4, pdiv, cdiv, static_cast<unsigned>(C));
        else
            DEEPIPR_LAUNCH(prof, k_bn_affine_bwd_v4g<false>, dim3(grid), dim3(kThreads), st,
                           reinterpret_cast<const float4 *>(dy), reinterpret_cast<const float4 *>(x), table_out,
                           reinterpret_cast<float4 *>(dx), n4, pdiv, cdiv, static_cast<unsigned>(C));
    } else {
        const FastDiv pdiv = make_fastdiv(static_cast<unsigned>(HW));
        const int grid = grid_for(total);
        if (relu)
            DEEPIPR_LAUNCH(prof, k_bn_affine_bwd_s<true>, dim3(grid), dim3(kThreads), st, dy, x, table_out, dx,
                               static_cast<unsigned>(total), pdiv, cdiv, static_cast<unsigned>(C));
        else
            DEEPIPR_LAUNCH(prof, k_bn_affine_bwd_s<false>, dim3(grid), dim3(kThreads), st, dy, x, table_out, dx,
                               static_cast<unsigned>(total), pdiv, cdiv, static_cast<unsigned>(C));
    }
    return check_launch("passport_bn_bwd(apply)");
}


int deepipr_passport_gn_supported(int N, int C, int HW, int groups) {
    GnPlan pl;
    return (!bad_dims(N, C, HW) && plan_gn(N, C, HW, groups, &pl)) ? 1 : 0;
}

size_t deepipr_passport_gn_workspace_bytes(int N, int C, int HW) {
    if (bad_dims(N, C, HW)) return 0;
    return static_cast<size_t>(N) * 2 * C * sizeof(double);
}

int deepipr_passport_gn_fwd(const float *x, const float *W, const double *m, const float *gamma_in,
                            const float *beta_in, const float *b, float alpha, float margin, float l2, int groups,
                            float eps, int N, int C, int HW, int K, int relu, float *y, float *stats, float *gamma,
                            float *beta, float *loss, float *acc, int8_t *bits, void *stream) {
    if (!x || !y || !stats || bad_dims(N, C, HW)) return fail(DEEPIPR_EINVAL, "passport_gn_fwd: bad argument");
    if (W && (!m || !gamma || !beta || K <= 0)) return fail(DEEPIPR_EINVAL, "passport_gn_fwd: W needs m, gamma, beta, K");
    if (loss && (!b || !acc)) return fail(DEEPIPR_EINVAL, "passport_gn_fwd: sign loss needs b and acc");
    if (loss && !W && !gamma_in) return fail(DEEPIPR_EINVAL, "passport_gn_fwd: sign loss needs a gamma");
    GnPlan pl;
    if (!aligned16(x) || !aligned16(y) || !plan_gn(N, C, HW, groups, &pl))
        return fail(DEEPIPR_EUNSUPPORTED, "passport_gn_fwd: group of %d channels x %d does not fit the fused form",
                    groups > 0 ? C / groups : 0, HW);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const float *g = gamma_in, *bt = beta_in;
    if (W) {
        int rc = deepipr_gamma_beta_fwd(W, m, C, K, gamma, beta, stream);
        if (rc != DEEPIPR_OK) return rc;
        g = gamma;
        bt = beta;
    }
    {
        ProfScope prof(DEEPIPR_K_GN_FWD, st);
        prof.bytes = 8.0 * static_cast<double>(N) * C * HW;
        const dim3 grid((pl.chunks + pl.T / pl.TG - 1) / (pl.T / pl.TG));
        const float4 *x4 = reinterpret_cast<const float4 *>(x);
        float4 *y4 = reinterpret_cast<float4 *>(y);
        if (prof.a) {
            prof.used = true;
            switch (pl.F4) {
                case 1: hipExtLaunchKernelGGL((k_gn_fwd<1>), grid, dim3(pl.T), 0, st, prof.a, prof.b, 0, x4, y4, g, bt, stats, relu, pl, eps); break;
                case 2: hipExtLaunchKernelGGL((k_gn_fwd<2>), grid, dim3(pl.T), 0, st, prof.a, prof.b, 0, x4, y4, g, bt, stats, relu, pl, eps); break;
                case 4: hipExtLaunchKernelGGL((k_gn_fwd<4>), grid, dim3(pl.T), 0, st, prof.a, prof.b, 0, x4, y4, g, bt, stats, relu, pl, eps); break;
                default: hipExtLaunchKernelGGL((k_gn_fwd<8>), grid, dim3(pl.T), 0, st, prof.a, prof.b, 0, x4, y4, g, bt, stats, relu, pl, eps); break;
            }
        } else {
            DEEPIPR_GN_CASES(k_gn_fwd, 0, x4, y4, g, bt, stats, relu, pl, eps)
        }
        int rc = check_launch("passport_gn_fwd");
        if (rc != DEEPIPR_OK) return rc;
    }
    if (loss) return deepipr_sign_loss_fwd(g, b, alpha, margin, l2, C, loss, acc, bits, stream);
    return DEEPIPR_OK;
}

int deepipr_passport_gn_bwd(const float *dy, const float *x, const float *stats, const float *gamma,
                            const float *beta, const double *m, const float *b, float alpha, float margin, float l2,
                            const float *dloss, const float *dgamma_extra, const float *dbeta_extra, int groups,
                            int N, int C, int HW, int K, int relu, float *dx, float *dW, float *dgamma, float *dbeta,
                            void *workspace, void *stream) {
    if (!dy || !x || !stats || !dx || !dgamma || !dbeta || !workspace || bad_dims(N, C, HW))
        return fail(DEEPIPR_EINVAL, "passport_gn_bwd: bad argument");
    if (dW && (!m || !gamma || K <= 0)) return fail(DEEPIPR_EINVAL, "passport_gn_bwd: dW needs m, gamma and K");
    if (dloss && (!b || !gamma)) return fail(DEEPIPR_EINVAL, "passport_gn_bwd: sign loss needs b and gamma");
    if (!dW && (dgamma_extra || dbeta_extra) && !gamma) return fail(DEEPIPR_EINVAL, "passport_gn_bwd: gamma missing");
    GnPlan pl;
    if (!aligned16(x) || !aligned16(dy) || !aligned16(dx) || !plan_gn(N, C, HW, groups, &pl))
        return fail(DEEPIPR_EUNSUPPORTED, "passport_gn_bwd: group does not fit the fused form");
    hipStream_t st = static_cast<hipStream_t>(stream);
    double *part = static_cast<double *>(workspace);
    {
        ProfScope prof(DEEPIPR_K_GN_BWD, st);
        prof.bytes = 12.0 * static_cast<double>(N) * C * HW;
        const int cpb = pl.T / pl.TG;
        const dim3 grid((pl.chunks + cpb - 1) / cpb);
        const unsigned lds = static_cast<unsigned>(static_cast<size_t>(cpb) * pl.U * 2 * sizeof(float));
        const float4 *d4 = reinterpret_cast<const float4 *>(dy), *x4 = reinterpret_cast<const float4 *>(x);
        float4 *o4 = reinterpret_cast<float4 *>(dx);
        if (prof.a) {
            prof.used = true;
            switch (pl.F4) {
                case 1: hipExtLaunchKernelGGL((k_gn_bwd<1>), grid, dim3(pl.T), lds, st, prof.a, prof.b, 0, d4, x4, stats, gamma, beta, o4, part, relu, C, pl); break;
                case 2: hipExtLaunchKernelGGL((k_gn_bwd<2>), grid, dim3(pl.T), lds, st, prof.a, prof.b, 0, d4, x4, stats, gamma, beta, o4, part, relu, C, pl); break;
                case 4: hipExtLaunchKernelGGL((k_gn_bwd<4>), grid, dim3(pl.T), lds, st, prof.a, prof.b, 0, d4, x4, stats, gamma, beta, o4, part, relu, C, pl); break;
                default: hipExtLaunchKernelGGL((k_gn_bwd<8>), grid, dim3(pl.T), lds, st, prof.a, prof.b, 0, d4, x4, stats, gamma, beta, o4, part, relu, C, pl); break;
            }
        } else {
            DEEPIPR_GN_CASES(k_gn_bwd, lds, d4, x4, stats, gamma, beta, o4, part, relu, C, pl)
        }
        int rc = check_launch("passport_gn_bwd");
        if (rc != DEEPIPR_OK) return rc;
    }
    if (!dW && !dloss && !dgamma_extra && !dbeta_extra) {
        ProfScope prof(DEEPIPR_K_REDUCE_PARTIALS, st);
        DEEPIPR_LAUNCH(prof, k_reduce_partials, dim3((C + 3) / 4), dim3(kThreads), st, part, N, C,
                       dgamma, dbeta);
        return check_launch("passport_gn_bwd(finish)");
    }
    // (dW == nullptr with a sign loss / external dgamma, dbeta: the passport branch whose rank-2 update the caller adds
    // to the data convolution's wgrad afterwards, deepipr_gamma_beta_bwd_acc)
    return launch_passport_finish(part, N, C, gamma, b, alpha, margin, l2, dloss, dgamma_extra, dbeta_extra, m, K,
                                  dgamma, dbeta, dW, st);
}


}  // extern "C"

namespace {
int launch_sgd(float *param, const float *grad, float *momentum_buf, size_t n, float lr, float momentum,
               float weight_decay, float grad_scale, const float *hyper, hipStream_t st) {
    ProfScope prof(DEEPIPR_K_SGD, st);
    prof.bytes = 20.0 * static_cast<double>(n);
    if (n % 4 == 0 && aligned16(param) && aligned16(grad) && aligned16(momentum_buf)) {
        const size_t n4 = n / 4;
        DEEPIPR_LAUNCH(prof, k_sgd_momentum_v4, dim3(grid_for(n4)), dim3(kThreads), st, reinterpret_cast<float4 *>(param),
                       reinterpret_cast<const float4 *>(grad), reinterpret_cast<float4 *>(momentum_buf), n4, lr,
                       momentum, weight_decay, grad_scale, hyper);
    } else {
        DEEPIPR_LAUNCH(prof, k_sgd_momentum_s, dim3(grid_for(n)), dim3(kThreads), st, param, grad, momentum_buf, n, lr,
                       momentum, weight_decay, grad_scale, hyper);
    }
    return check_launch("sgd_momentum_step");
}
}  // namespace

extern "C" {

int deepipr_sgd_momentum_step(float *param, const float *grad, float *momentum_buf, size_t n, float lr,
                              float momentum, float weight_decay, float grad_scale, void *stream) {
    if (!param || !grad || !momentum_buf || n == 0) return fail(DEEPIPR_EINVAL, "sgd_momentum_step: bad argument");
    return launch_sgd(param, grad, momentum_buf, n, lr, momentum, weight_decay, grad_scale, nullptr,
                      static_cast<hipStream_t>(stream));
}

int deepipr_sgd_momentum_chunk(void) { return kSgdChunk; }

int deepipr_sgd_momentum_step_multi(float *param, float *momentum_buf, const long long *table, int entries,
                                    size_t total_elements, const float *hyper, void *stream) {
    if (!param || !momentum_buf || !table || !hyper || entries <= 0)
        return fail(DEEPIPR_EINVAL, "sgd_momentum_step_multi: bad argument");
    hipStream_t st = static_cast<hipStream_t>(stream);
    ProfScope prof(DEEPIPR_K_SGD, st);
    prof.bytes = 20.0 * static_cast<double>(total_elements);
    DEEPIPR_LAUNCH(prof, k_sgd_momentum_multi, dim3(entries), dim3(kThreads), st, param, momentum_buf, table, hyper);
    return check_launch("sgd_momentum_step_multi");
}

int deepipr_sgd_momentum_step_dev(float *param, const float *grad, float *momentum_buf, size_t n,
                                  const float *hyper, void *stream) {
    if (!param || !grad || !momentum_buf || !hyper || n == 0)
        return fail(DEEPIPR_EINVAL, "sgd_momentum_step_dev: bad argument");
    return launch_sgd(param, grad, momentum_buf, n, 0.0f, 0.0f, 0.0f, 1.0f, hyper, static_cast<hipStream_t>(stream));
}


int deepipr_ce_top1_supported(int N, int C) {
    return (N > 0 && C > 0 && static_cast<long long>(N) * C <= (1ll << 20)) ? 1 : 0;
}

size_t deepipr_ce_top1_workspace_bytes(int N) { return N > 0 ? static_cast<size_t>(N) * 2 * sizeof(double) : 0; }

int deepipr_ce_top1_fwd(const float *logits, const long long *target, int N, int C, float *loss, float *top1_pct,
                        float *lse, void *workspace, void *stream) {
    if (!logits || !target || !loss || !top1_pct || !lse || !workspace || N <= 0 || C <= 0)
        return fail(DEEPIPR_EINVAL, "ce_top1_fwd: bad argument");
    if (!deepipr_ce_top1_supported(N, C))
        return fail(DEEPIPR_EUNSUPPORTED, "ce_top1_fwd: more than 2^20 logits (use the library ops)");
    hipStream_t st = static_cast<hipStream_t>(stream);
    double *part = static_cast<double *>(workspace);
    const int rows_per_wg = kThreads / kWave;
    hipLaunchKernelGGL(k_ce_rows, dim3((N + rows_per_wg - 1) / rows_per_wg), dim3(kThreads), 0, st, logits, target, N, C,
                       lse, part);
    hipLaunchKernelGGL(k_ce_finish, dim3(1), dim3(kThreads), 0, st, part, N, loss, top1_pct);
    return check_launch("ce_top1_fwd");
}

int deepipr_pooled_linear_supported(int N, int C, int HW, int K) {
    return (N > 0 && C > 0 && C % 64 == 0 && C <= kHeadMaxC && HW > 0 && HW % 4 == 0 && HW <= 256 && K > 0 && K <= kHeadMaxK
            && static_cast<long long>(N) * C * HW < (1ll << 31)) ? 1 : 0;
}

int deepipr_pooled_linear_fwd(const float *x, const float *W, const float *b, float *pooled, float *logits, int N, int C, int HW,
                              int K, void *stream) {
    if (!x || !W || !pooled || !logits) return fail(DEEPIPR_EINVAL, "pooled_linear_fwd: null pointer");
    if (!deepipr_pooled_linear_supported(N, C, HW, K))
        return fail(DEEPIPR_EUNSUPPORTED, "pooled_linear_fwd: C a multiple of 64 up to %d, HW a multiple of 4, at most %d classes "
                    "(use the library's pooling + Linear)", kHeadMaxC, kHeadMaxK);
    if (!aligned16(x)) return fail(DEEPIPR_EINVAL, "pooled_linear_fwd: x must be 16-byte aligned");
    hipStream_t st = static_cast<hipStream_t>(stream);
    ProfScope prof(DEEPIPR_K_HEAD, st);
    prof.bytes = 4.0 * (static_cast<double>(N) * C * HW + static_cast<double>(N) * C + static_cast<double>(K) * C);
    const int kblocks = (K + 7) / 8;
    DEEPIPR_LAUNCH(prof, k_pooled_linear_fwd, dim3(N * kblocks), dim3(kThreads), st, x, W, b, pooled, logits, C, HW, K, kblocks);
    return check_launch("pooled_linear_fwd");
}

int deepipr_pooled_linear_bwd(const float *dlogits, const float *W, const float *pooled, float *dx, float *dW, float *db, int N,
                              int C, int HW, int K, void *stream) {
    if (!dlogits || !W || !pooled || !dx || !dW) return fail(DEEPIPR_EINVAL, "pooled_linear_bwd: null pointer");
    if (!deepipr_pooled_linear_supported(N, C, HW, K)) return fail(DEEPIPR_EUNSUPPORTED, "pooled_linear_bwd: shape outside the kernel");
    if (!aligned16(dx)) return fail(DEEPIPR_EINVAL, "pooled_linear_bwd: dx must be 16-byte aligned");
    hipStream_t st = static_cast<hipStream_t>(stream);
    ProfScope prof(DEEPIPR_K_HEAD, st);
    prof.bytes = 4.0 * (static_cast<double>(N) * C * HW + static_cast<double>(N) * C + 2.0 * K * C);
    const int grid = N * ((C + kThreads - 1) / kThreads) + (C / 64) * ((K + 7) / 8);
    DEEPIPR_LAUNCH(prof, k_pooled_linear_bwd, dim3(grid), dim3(kThreads), st, dlogits, W, pooled, dx, dW, db, N, C, HW, K);
    return check_launch("pooled_linear_bwd");
}

int deepipr_ce_bwd(const float *dloss, const float *logits, const long long *target, const float *lse, int N, int C,
                   float *dlogits, void *stream) {
    if (!dloss || !logits || !target || !lse || !dlogits || N <= 0 || C <= 0)
        return fail(DEEPIPR_EINVAL, "ce_bwd: bad argument");
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(k_ce_bwd, dim3(grid_for(static_cast<size_t>(N) * C)), dim3(kThreads), 0, st, dloss, logits,
                       target, lse, N, C, dlogits);
    return check_launch("ce_bwd");
}

int deepipr_scalar_sums(const float *const *terms, int n_a, int n_b, float *out, void *stream) {
    if (!terms || !out || n_a < 0 || n_b < 0 || n_a + n_b <= 0 || n_a + n_b > kScalarTermsMax)
        return fail(DEEPIPR_EINVAL, "scalar_sums: 1..%d terms", kScalarTermsMax);
    ScalarTerms T{};
    T.na = n_a;
    T.nb = n_b;
    for (int i = 0; i < n_a + n_b; ++i) {
        if (!terms[i]) return fail(DEEPIPR_EINVAL, "scalar_sums: null term %d", i);
        T.p[i] = terms[i];
    }
    hipLaunchKernelGGL(k_scalar_sums, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), T, out);
    return check_launch("scalar_sums");
}

int deepipr_add_relu_fwd(const float *a, const float *b, float *out, size_t n, void *stream) {
    if (!a || !b || !out || n == 0) return fail(DEEPIPR_EINVAL, "add_relu_fwd: bad argument");
    if (!aligned16(a) || !aligned16(b) || !aligned16(out)) return fail(DEEPIPR_EINVAL, "add_relu_fwd: pointers must be 16-byte aligned");
    hipStream_t st = static_cast<hipStream_t>(stream);
    ProfScope prof(DEEPIPR_K_ADD_RELU, st);
    prof.bytes = 12.0 * static_cast<double>(n);
    DEEPIPR_LAUNCH(prof, k_add_relu_fwd, dim3(grid_for((n + 3) / 4)), dim3(kThreads), st, a, b, out, n);
    return check_launch("add_relu_fwd");
}

int deepipr_relu_bwd(const float *dy, const float *out, float *dx, size_t n, void *stream) {
    return deepipr_relu_bwd2(dy, nullptr, out, dx, n, stream);
}

int deepipr_relu_bwd2(const float *dy, const float *dy2, const float *out, float *dx, size_t n, void *stream) {
    if (!dy || !out || !dx || n == 0) return fail(DEEPIPR_EINVAL, "relu_bwd: bad argument");
    if (!aligned16(dy) || !aligned16(out) || !aligned16(dx) || (dy2 && !aligned16(dy2)))
        return fail(DEEPIPR_EINVAL, "relu_bwd: pointers must be 16-byte aligned");
    hipStream_t st = static_cast<hipStream_t>(stream);
    ProfScope prof(DEEPIPR_K_ADD_RELU, st);
    prof.bytes = (dy2 ? 16.0 : 12.0) * static_cast<double>(n);
    if (dy2) DEEPIPR_LAUNCH(prof, k_relu_bwd<true>, dim3(grid_for((n + 3) / 4)), dim3(kThreads), st, dy, dy2, out, dx, n);
    else DEEPIPR_LAUNCH(prof, k_relu_bwd<false>, dim3(grid_for((n + 3) / 4)), dim3(kThreads), st, dy, dy2, out, dx, n);
    return check_launch("relu_bwd");
}

}  // extern "C"

// =============================================================================================
// 3x3 stride-2 pad-1 max-pool of the ImageNet stem (models/resnet_passport.py:94-98: nn.MaxPool2d(3, 2, 1) behind the 7x7
// convolution) -- HBM-bound.  ATen keeps the argmax as int64 flat indices (8 B per OUTPUT element, 411 MB at batch 256) and
// its backward is one thread per input element walking them (1.55 ms for the 822 MB map, profiles/r05b_steady_state_r50.md).
// Here the argmax is the window SLOT (0 .. 8, one byte), forward is one pass, backward a gather in ATen's own order (windows by
// ascending row, then column: the up to four contributions of an input pixel are added in the same order, so dx is bit for bit
// ATen's) with no atomics.  Algorithmic bytes: forward 4 |x| + 5 |y|, backward 4 |dx| + 5 |dy|.
// =============================================================================================
namespace {
// Plane index from a 2-D grid (y, z): planes may exceed 65 535.  Everything inside a plane is 32-bit arithmetic (a flat 64-bit
// index costs a 64-bit division per thread: the first version of these kernels ran at ATen's speed because of it).
__device__ __forceinline__ size_t pool_plane() { return static_cast<size_t>(blockIdx.z) * 65535u + blockIdx.y; }

__device__ __forceinline__ void pool_take(float v, int s, float &best, int &bs) {
    if (v > best || v != v) {             // ties keep the first maximum in scan order; NaN wins and the LAST NaN is kept
        best = v;                         // (at::native::max_pool_forward_nchw: `if ((val > maxval) || isnan(val))`)
        bs = s;
    }
}

// V2: one thread per PAIR of outputs (ow = 2 q, 2 q + 1; W a multiple of 4): per input row one aligned float4 (columns
// 4 q .. 4 q + 3) and one scalar (column 4 q - 1).  Otherwise one thread per output.
template <bool V2>
__global__ __launch_bounds__(256) void k_maxpool3x3s2_fwd(const float *__restrict__ x, float *__restrict__ y,
                                                          unsigned char *__restrict__ slot, size_t planes, int H, int W, int OH, int OW) {
    const size_t plane = pool_plane();
    if (plane >= planes) return;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const float *xp = x + plane * H * W;
    if (V2) {
        const int OW2 = OW / 2;
        if (i >= OH * OW2) return;
        const int q = i % OW2, oh = i / OW2;
        const int h0 = 2 * oh - 1;
        float b0 = -INFINITY, b1 = -INFINITY;
        int s0 = (h0 < 0 ? 3 : 0) + (q == 0 ? 1 : 0), s1 = (h0 < 0 ? 3 : 0);      // the windows' first elements inside the map
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const int h = h0 + a;
            if (h < 0 || h >= H) continue;
            const float4 v = *reinterpret_cast<const float4 *>(xp + h * W + 4 * q);
            if (q > 0) pool_take(xp[h * W + 4 * q - 1], 3 * a, b0, s0);
            pool_take(v.x, 3 * a + 1, b0, s0);
            pool_take(v.y, 3 * a + 2, b0, s0);
            pool_take(v.y, 3 * a, b1, s1);
            pool_take(v.z, 3 * a + 1, b1, s1);
            pool_take(v.w, 3 * a + 2, b1, s1);
        }
        const size_t o = (plane * OH + oh) * OW + 2 * q;
        *reinterpret_cast<float2 *>(y + o) = make_float2(b0, b1);
        *reinterpret_cast<uchar2 *>(slot + o) = make_uchar2(static_cast<unsigned char>(s0), static_cast<unsigned char>(s1));
    } else {
        if (i >= OH * OW) return;
        const int ow = i % OW, oh = i / OW;
        const int h0 = 2 * oh - 1, w0 = 2 * ow - 1;
        float best = -INFINITY;
        int bs = (h0 < 0 ? 3 : 0) + (w0 < 0 ? 1 : 0);
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const int h = h0 + a;
            if (h < 0 || h >= H) continue;
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                const int w = w0 + b;
                if (w < 0 || w >= W) continue;
                pool_take(xp[h * W + w], 3 * a + b, best, bs);
            }
        }
        const size_t o = plane * OH * OW + i;
        y[o] = best;
        slot[o] = static_cast<unsigned char>(bs);
    }
}

// the windows (ph, pw) that contain input pixel (h, w), ph then pw ascending (ATen's order of additions): 2 p - 1 <= h <= 2 p + 1
__device__ __forceinline__ float pool_gather(const float *__restrict__ dp, const unsigned char *__restrict__ sp, int h, int w,
                                             int OH, int OW) {
    const int ph_lo = h / 2, ph_hi = min((h + 1) / 2, OH - 1);
    const int pw_lo = w / 2, pw_hi = min((w + 1) / 2, OW - 1);
    float g = 0.f;
    for (int ph = ph_lo; ph <= ph_hi; ++ph)
        for (int pw = pw_lo; pw <= pw_hi; ++pw) {
            const int s = 3 * (h - (2 * ph - 1)) + (w - (2 * pw - 1));
            if (sp[ph * OW + pw] == s) g += dp[ph * OW + pw];
        }
    return g;
}

// V4: one thread per float4 of dx (W a multiple of 4); otherwise one thread per input element
template <bool V4>
__global__ __launch_bounds__(256) void k_maxpool3x3s2_bwd(const float *__restrict__ dy, const unsigned char *__restrict__ slot,
                                                          float *__restrict__ dx, size_t planes, int H, int W, int OH, int OW) {
    const size_t plane = pool_plane();
    if (plane >= planes) return;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const float *dp = dy + plane * OH * OW;
    const unsigned char *sp = slot + plane * OH * OW;
    if (V4) {
        const int W4 = W / 4;
        if (i >= H * W4) return;
        const int q = i % W4, h = i / W4;
        // columns 4 q .. 4 q + 3 lie in the windows pw = 2 q, 2 q + 1 (and 2 q + 2 for the last column), row h in ph = h / 2 (and
        // (h + 1) / 2 for odd h).  Per window row: dy and the slots of the three windows (8 + 4 and 2 + 1 bytes); a selected
        // add per (pixel, window) in pool_gather's order -- adding +0 where the slot does not match changes no bit.
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        const int c0 = 2 * q;
        const bool third = c0 + 2 < OW;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int ph = h / 2 + r;
            if (r == 1 && (!(h & 1) || ph >= OH)) break;
            const int dh = h - 2 * ph + 1;                     // the pixel's row inside window row ph
            const float2 d01 = *reinterpret_cast<const float2 *>(dp + ph * OW + c0);
            const uchar2 s01 = *reinterpret_cast<const uchar2 *>(sp + ph * OW + c0);
            const float d2 = third ? dp[ph * OW + c0 + 2] : 0.f;
            const int s2 = third ? sp[ph * OW + c0 + 2] : 255;
            o.x += s01.x == 3 * dh + 1 ? d01.x : 0.f;
            o.y += s01.x == 3 * dh + 2 ? d01.x : 0.f;
            o.y += s01.y == 3 * dh ? d01.y : 0.f;
            o.z += s01.y == 3 * dh + 1 ? d01.y : 0.f;
            o.w += s01.y == 3 * dh + 2 ? d01.y : 0.f;
            o.w += s2 == 3 * dh ? d2 : 0.f;
        }
        *reinterpret_cast<float4 *>(dx + (plane * H + h) * W + 4 * q) = o;
    } else {
        if (i >= H * W) return;
        dx[plane * H * W + i] = pool_gather(dp, sp, i / W, i % W, OH, OW);
    }
}
// 2x2 / 2: one thread per PAIR of windows (input columns 4 q .. 4 q + 3 of rows 2 oh, 2 oh + 1)
__global__ __launch_bounds__(256) void k_maxpool2x2s2_fwd(const float *__restrict__ x, float *__restrict__ y,
                                                          unsigned char *__restrict__ slot, size_t planes, int H, int W) {
    const size_t plane = pool_plane();
    if (plane >= planes) return;
    const int i = blockIdx.x * 256 + threadIdx.x, OH = H / 2, OW = W / 2, OW2 = W / 4;
    if (i >= OH * OW2) return;
    const int q = i % OW2, oh = i / OW2;
    const float *xp = x + plane * H * W + (2 * oh) * W + 4 * q;
    const float4 r0 = *reinterpret_cast<const float4 *>(xp), r1 = *reinterpret_cast<const float4 *>(xp + W);
    float b0 = -INFINITY, b1 = -INFINITY;
    int s0 = 0, s1 = 0;
    pool_take(r0.x, 0, b0, s0);
    pool_take(r0.y, 1, b0, s0);
    pool_take(r1.x, 2, b0, s0);
    pool_take(r1.y, 3, b0, s0);
    pool_take(r0.z, 0, b1, s1);
    pool_take(r0.w, 1, b1, s1);
    pool_take(r1.z, 2, b1, s1);
    pool_take(r1.w, 3, b1, s1);
    const size_t o = (plane * OH + oh) * OW + 2 * q;
    *reinterpret_cast<float2 *>(y + o) = make_float2(b0, b1);
    *reinterpret_cast<uchar2 *>(slot + o) = make_uchar2(static_cast<unsigned char>(s0), static_cast<unsigned char>(s1));
}

// one thread per float4 of dx: two windows' gradients, selected by their slots
__global__ __launch_bounds__(256) void k_maxpool2x2s2_bwd(const float *__restrict__ dy, const unsigned char *__restrict__ slot,
                                                          float *__restrict__ dx, size_t planes, int H, int W) {
    const size_t plane = pool_plane();
    if (plane >= planes) return;
    const int i = blockIdx.x * 256 + threadIdx.x, OW = W / 2, W4 = W / 4;
    if (i >= H * W4) return;
    const int q = i % W4, h = i / W4;
    const size_t o = (plane * (H / 2) + h / 2) * OW + 2 * q;
    const float2 d = *reinterpret_cast<const float2 *>(dy + o);
    const uchar2 s = *reinterpret_cast<const uchar2 *>(slot + o);
    const int base = 2 * (h & 1);
    const float4 v = make_float4(s.x == base ? d.x : 0.f, s.x == base + 1 ? d.x : 0.f, s.y == base ? d.y : 0.f, s.y == base + 1 ? d.y : 0.f);
    *reinterpret_cast<float4 *>(dx + (plane * H + h) * W + 4 * q) = v;
}
}  // namespace

extern "C" {

int deepipr_maxpool3x3s2_fwd(const float *x, float *y, unsigned char *slot, size_t planes, int H, int W, void *stream) {
    if (!x || !y || !slot || planes == 0 || H <= 0 || W <= 0) return fail(DEEPIPR_EINVAL, "maxpool3x3s2_fwd: bad argument");
    const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
    if (static_cast<long long>(H) * W >= (1ll << 30) || planes >= 65535ull * 65535ull) return fail(DEEPIPR_EINVAL, "maxpool3x3s2_fwd: tensor too large");
    const bool v2 = W % 4 == 0 && aligned16(x) && (reinterpret_cast<uintptr_t>(y) & 7u) == 0 && (reinterpret_cast<uintptr_t>(slot) & 1u) == 0;
    hipStream_t st = static_cast<hipStream_t>(stream);
    ProfScope prof(DEEPIPR_K_MAXPOOL, st);
    prof.bytes = 4.0 * static_cast<double>(planes) * H * W + 5.0 * static_cast<double>(planes) * OH * OW;
    const unsigned py = static_cast<unsigned>(planes < 65535 ? planes : 65535), pz = static_cast<unsigned>((planes + 65534) / 65535);
    const dim3 grid(static_cast<unsigned>(((v2 ? OH * (OW / 2) : OH * OW) + 255) / 256), py, pz);
    if (v2) DEEPIPR_LAUNCH(prof, k_maxpool3x3s2_fwd<true>, grid, dim3(256), st, x, y, slot, planes, H, W, OH, OW);
    else DEEPIPR_LAUNCH(prof, k_maxpool3x3s2_fwd<false>, grid, dim3(256), st, x, y, slot, planes, H, W, OH, OW);
    return check_launch("maxpool3x3s2_fwd");
}

int deepipr_maxpool3x3s2_bwd(const float *dy, const unsigned char *slot, float *dx, size_t planes, int H, int W, void *stream) {
    if (!dy || !dx || !slot || planes == 0 || H <= 0 || W <= 0) return fail(DEEPIPR_EINVAL, "maxpool3x3s2_bwd: bad argument");
    const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
    if (static_cast<long long>(H) * W >= (1ll << 30) || planes >= 65535ull * 65535ull) return fail(DEEPIPR_EINVAL, "maxpool3x3s2_bwd: tensor too large");
    const bool v4 = W % 4 == 0 && aligned16(dx) && (reinterpret_cast<uintptr_t>(dy) & 7u) == 0 && (reinterpret_cast<uintptr_t>(slot) & 1u) == 0;
    hipStream_t st = static_cast<hipStream_t>(stream);
    ProfScope prof(DEEPIPR_K_MAXPOOL, st);
    prof.bytes = 4.0 * static_cast<double>(planes) * H * W + 5.0 * static_cast<double>(planes) * OH * OW;
    const unsigned py = static_cast<unsigned>(planes < 65535 ? planes : 65535), pz = static_cast<unsigned>((planes + 65534) / 65535);
    const dim3 grid(static_cast<unsigned>(((v4 ? H * (W / 4) : H * W) + 255) / 256), py, pz);
    if (v4) DEEPIPR_LAUNCH(prof, k_maxpool3x3s2_bwd<true>, grid, dim3(256), st, dy, slot, dx, planes, H, W, OH, OW);
    else DEEPIPR_LAUNCH(prof, k_maxpool3x3s2_bwd<false>, grid, dim3(256), st, dy, slot, dx, planes, H, W, OH, OW);
    return check_launch("maxpool3x3s2_bwd");
}

// ---- 2x2 stride-2 max-pool (the CIFAR AlexNet's pools, models/alexnet_passport.py:30-38 of the reference: nn.MaxPool2d(2, 2)):
// non-overlapping windows, so the backward is a select -- dx = dy where the window's slot (0 .. 3, row-major, one byte) names
// this pixel, else 0 -- bit for bit ATen's.  Even H, W a multiple of 4 (two windows per thread, one float4 per input row).
int deepipr_maxpool2x2s2_fwd(const float *x, float *y, unsigned char *slot, size_t planes, int H, int W, void *stream) {
    if (!x || !y || !slot || planes == 0 || H <= 0 || W <= 0) return fail(DEEPIPR_EINVAL, "maxpool2x2s2_fwd: bad argument");
    if (H % 2 || W % 4 || !aligned16(x) || (reinterpret_cast<uintptr_t>(y) & 7u) || (reinterpret_cast<uintptr_t>(slot) & 1u))
        return fail(DEEPIPR_EUNSUPPORTED, "maxpool2x2s2_fwd: even H, W a multiple of 4, aligned pointers (use the library's pool)");
    if (static_cast<long long>(H) * W >= (1ll << 30) || planes >= 65535ull * 65535ull) return fail(DEEPIPR_EINVAL, "maxpool2x2s2_fwd: tensor too large");
    hipStream_t st = static_cast<hipStream_t>(stream);
    ProfScope prof(DEEPIPR_K_MAXPOOL, st);
    prof.bytes = 4.0 * static_cast<double>(planes) * H * W + 5.0 * static_cast<double>(planes) * (H / 2) * (W / 2);
    const unsigned py = static_cast<unsigned>(planes < 65535 ? planes : 65535), pz = static_cast<unsigned>((planes + 65534) / 65535);
    const dim3 grid(static_cast<unsigned>(((H / 2) * (W / 4) + 255) / 256), py, pz);
    DEEPIPR_LAUNCH(prof, k_maxpool2x2s2_fwd, grid, dim3(256), st, x, y, slot, planes, H, W);
    return check_launch("maxpool2x2s2_fwd");
}

int deepipr_maxpool2x2s2_bwd(const float *dy, const unsigned char *slot, float *dx, size_t planes, int H, int W, void *stream) {
    if (!dy || !dx || !slot || planes == 0 || H <= 0 || W <= 0) return fail(DEEPIPR_EINVAL, "maxpool2x2s2_bwd: bad argument");
    if (H % 2 || W % 4 || !aligned16(dx) || (reinterpret_cast<uintptr_t>(dy) & 7u) || (reinterpret_cast<uintptr_t>(slot) & 1u))
        return fail(DEEPIPR_EUNSUPPORTED, "maxpool2x2s2_bwd: even H, W a multiple of 4, aligned pointers");
    if (static_cast<long long>(H) * W >= (1ll << 30) || planes >= 65535ull * 65535ull) return fail(DEEPIPR_EINVAL, "maxpool2x2s2_bwd: tensor too large");
    hipStream_t st = static_cast<hipStream_t>(stream);
    ProfScope prof(DEEPIPR_K_MAXPOOL, st);
    prof.bytes = 4.0 * static_cast<double>(planes) * H * W + 5.0 * static_cast<double>(planes) * (H / 2) * (W / 2);
    const unsigned py = static_cast<unsigned>(planes < 65535 ? planes : 65535), pz = static_cast<unsigned>((planes + 65534) / 65535);
    const dim3 grid(static_cast<unsigned>((H * (W / 4) + 255) / 256), py, pz);
    DEEPIPR_LAUNCH(prof, k_maxpool2x2s2_bwd, grid, dim3(256), st, dy, slot, dx, planes, H, W);
    return check_launch("maxpool2x2s2_bwd");
}

}  // extern "C"

// =============================================================================================
// The 1x1 stride-2 projection shortcuts at ImageNet geometry (models/resnet_normal.py:41-42, resnet_passport.py:33-36 with the
// Bottleneck's channel counts): conv1x1(x, stride 2) = conv1x1(x[:, :, ::2, ::2]) -- a pixel gather in front of a plain
// stride-1 GEMM, and its backward-data a scatter of the GEMM's result between zeros.  Two streaming kernels instead of the
// vendor library's NHWC implicit GEMM with three layout transposes and a zero fill per call.  Bytes: 4 (|y| + |touched x|) /
// 4 (|dx| + |dy|).
// =============================================================================================
namespace {
template <bool V4>
__global__ __launch_bounds__(256) void k_subsample2(const float *__restrict__ x, float *__restrict__ y, size_t total, int W, int OH, int OW) {
    const size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x;      // V4: one thread per output PAIR
    if (i >= total) return;
    if (V4) {
        const int OW2 = OW / 2;
        const int q = static_cast<int>(i % OW2);
        const size_t r = i / OW2;                                               // plane * OH + oh
        const int oh = static_cast<int>(r % OH);
        const size_t plane = r / OH;
        const float4 v = *reinterpret_cast<const float4 *>(x + (plane * 2 * OH + 2 * oh) * W + 4 * q);
        *reinterpret_cast<float2 *>(y + r * OW + 2 * q) = make_float2(v.x, v.z);
    } else {
        const int ow = static_cast<int>(i % OW);
        const size_t r = i / OW;
        const int oh = static_cast<int>(r % OH);
        const size_t plane = r / OH;
        y[i] = x[(plane * 2 * OH + 2 * oh) * W + 2 * ow];
    }
}

template <bool V4>
__global__ __launch_bounds__(256) void k_upsample2_zero(const float *__restrict__ dy, float *__restrict__ dx, size_t total, int W, int OH, int OW) {
    const size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x;      // V4: one thread per float4 of dx
    if (i >= total) return;
    if (V4) {
        const int W4 = W / 4;
        const int q = static_cast<int>(i % W4);
        const size_t r = i / W4;                                                // plane * H + h
        const int h = static_cast<int>(r % (2 * OH));
        const size_t plane = r / (2 * OH);
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        if ((h & 1) == 0) {
            const float2 v = *reinterpret_cast<const float2 *>(dy + (plane * OH + (h >> 1)) * OW + 2 * q);
            o.x = v.x;
            o.z = v.y;
        }
        *reinterpret_cast<float4 *>(dx + r * W + 4 * q) = o;
    } else {
        const int w = static_cast<int>(i % W);
        const size_t r = i / W;
        const int h = static_cast<int>(r % (2 * OH));
        const size_t plane = r / (2 * OH);
        dx[i] = ((h | w) & 1) ? 0.f : dy[(plane * OH + (h >> 1)) * OW + (w >> 1)];
    }
}
}  // namespace

extern "C" {

int deepipr_subsample2(const float *x, float *y, size_t planes, int H, int W, void *stream) {
    if (!x || !y || planes == 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1)) return fail(DEEPIPR_EINVAL, "subsample2: bad argument (even H, W)");
    const int OH = H / 2, OW = W / 2;
    const bool v4 = W % 4 == 0 && aligned16(x) && (reinterpret_cast<uintptr_t>(y) & 7u) == 0;
    const size_t total = planes * OH * (v4 ? OW / 2 : OW);
    if ((total + 255) / 256 >= (1ull << 31)) return fail(DEEPIPR_EINVAL, "subsample2: tensor too large");
    hipStream_t st = static_cast<hipStream_t>(stream);
    ProfScope prof(DEEPIPR_K_RESAMPLE2, st);
    prof.bytes = 4.0 * static_cast<double>(planes) * OH * (OW + W);
    const dim3 grid(static_cast<unsigned>((total + 255) / 256));
    if (v4) DEEPIPR_LAUNCH(prof, k_subsample2<true>, grid, dim3(256), st, x, y, total, W, OH, OW);
    else DEEPIPR_LAUNCH(prof, k_subsample2<false>, grid, dim3(256), st, x, y, total, W, OH, OW);
    return check_launch("subsample2");
}

int deepipr_upsample2_zero(const float *dy, float *dx, size_t planes, int H, int W, void *stream) {
    if (!dy || !dx || planes == 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1)) return fail(DEEPIPR_EINVAL, "upsample2_zero: bad argument (even H, W)");
    const int OH = H / 2, OW = W / 2;
    const bool v4 = W % 4 == 0 && aligned16(dx) && (reinterpret_cast<uintptr_t>(dy) & 7u) == 0;
    const size_t total = planes * H * (v4 ? W / 4 : W);
    if ((total + 255) / 256 >= (1ull << 31)) return fail(DEEPIPR_EINVAL, "upsample2_zero: tensor too large");
    hipStream_t st = static_cast<hipStream_t>(stream);
    ProfScope prof(DEEPIPR_K_RESAMPLE2, st);
    prof.bytes = 4.0 * static_cast<double>(planes) * (static_cast<double>(H) * W + static_cast<double>(OH) * OW);
    const dim3 grid(static_cast<unsigned>((total + 255) / 256));
    if (v4) DEEPIPR_LAUNCH(prof, k_upsample2_zero<true>, grid, dim3(256), st, dy, dx, total, W, OH, OW);
    else DEEPIPR_LAUNCH(prof, k_upsample2_zero<false>, grid, dim3(256), st, dy, dx, total, W, OH, OW);
    return check_launch("upsample2_zero");
}

}  // extern "C"

// =============================================================================================
// data convolution: weight gradient on the fp32 matrix cores (kernels and planner: deepipr_conv.inc)
// =============================================================================================
namespace {
// Algorithm of the 3x3 stride-1 convolutions, all three directions: 1 = Winograd (F(2x2, 3x3) forward / backward-data,
// deepipr_conv_wino.inc; F(3x3, 2x2) weight gradient, deepipr_conv_wino_wgrad.inc; default), 0 = the direct implicit GEMMs.  DEEPIPR_CONV_ALGO=direct|winograd at load time, deepipr_conv_set_algo afterwards.
int g_conv_algo = -1;
int conv_algo() {
    if (g_conv_algo < 0) {
        const char *e = getenv("DEEPIPR_CONV_ALGO");
        g_conv_algo = (e && !strcmp(e, "direct")) ? 0 : 1;
    }
    return g_conv_algo;
}

#include "deepipr_conv.inc"
#include "deepipr_conv_fwd.inc"
#include "deepipr_conv_1x1.inc"
#include "deepipr_conv_stem7.inc"

template <class C>
void launch_wgrad(ProfScope &prof, const WgradPlan &p, const float *x, const float *dy, float *part, int Ci, int Co, int H,
                  hipStream_t st) {
    const int grid = p.splits * p.tiles_co * p.tiles_ci;
    DEEPIPR_LAUNCH(prof, (k_conv3x3_wgrad<C>), dim3(grid), dim3(kWgThreads), st, x, dy, part, Ci, Co, H, p.tiles_co, p.tiles_ci,
                   p.chunks, p.chunks_per_split);
}

template <class C>
void launch_wgrad_b3(ProfScope &prof, const WgradPlan &p, const float *x, const float *dy, float *part, int Ci, int Co, int H,
                     hipStream_t st) {
    const int grid = p.splits * p.tiles_co * p.tiles_ci;
    DEEPIPR_LAUNCH(prof, (k_conv3x3_wgrad_b3<C>), dim3(grid), dim3(kWgThreads), st, x, dy, part, Ci, Co, H, p.tiles_co,
                   p.tiles_ci, p.chunks, p.chunks_per_split);
}
}  // namespace

extern "C" {

int deepipr_conv_set_arith(int mode) {
    if (mode != 0 && mode != 1) return fail(DEEPIPR_EINVAL, "conv_set_arith: 0 (fp32 MFMA) or 1 (bf16x3)");
    g_conv_arith = mode;
    return DEEPIPR_OK;
}

int deepipr_conv_get_arith(void) { return conv_arith(); }

size_t deepipr_conv_wgrad_workspace_bytes(int N, int Ci, int Co, int H, int W, int kh, int kw, int stride, int pad) {
    return plan_wgrad(N, Ci, Co, H, W, kh, kw, stride, pad).workspace;
}

int deepipr_conv_wgrad(const float *x, const float *dy, float *dW, int N, int Ci, int Co, int H, int W, int kh, int kw,
                       int stride, int pad, const float *dgamma, const float *dbeta, const double *m, void *workspace,
                       size_t workspace_bytes, void *stream) {
    if (!x || !dy || !dW) return fail(DEEPIPR_EINVAL, "conv_wgrad: null pointer");
    const WgradPlan p = plan_wgrad(N, Ci, Co, H, W, kh, kw, stride, pad);
    if (!p.cfg)
        return fail(DEEPIPR_EUNSUPPORTED, "conv_wgrad: only 3x3 pad-1 convolutions of stride 1 / 2 and 1x1 pad-0 stride-2 ones, Ci, Co "
                    "multiples of 64, output maps 4/8/16/32 (stride 2: 4/8/16) wide (use the library's weight gradient)");
    if (!workspace || workspace_bytes < p.workspace) return fail(DEEPIPR_EINVAL, "conv_wgrad: workspace too small");
    if (!aligned16(x) || !aligned16(dy) || !aligned16(workspace)) return fail(DEEPIPR_EINVAL, "conv_wgrad: pointers must be 16-byte aligned");
    const bool rank2 = dgamma || dbeta || m;
    if (rank2 && !(dgamma && dbeta && m)) return fail(DEEPIPR_EINVAL, "conv_wgrad: dgamma, dbeta and m go together");
    hipStream_t st = static_cast<hipStream_t>(stream);
    float *part = static_cast<float *>(workspace);
    const int grid = p.splits * p.tiles_co * p.tiles_ci;
    if (p.cfg == 3000) {                                                 // the stem (Ci = 3)
        if (rank2) return fail(DEEPIPR_EUNSUPPORTED, "conv_wgrad: no fused rank-2 term for a 3-channel input");
        {
            ProfScope prof(DEEPIPR_K_CONV_WGRAD, st);
            prof.bytes = 2.0 * Co * 27.0 * static_cast<double>(N) * H * W;
            DEEPIPR_LAUNCH(prof, k_conv_stem_wgrad, dim3(grid), dim3(256), st, x, dy, part, Co, H, p.tiles_co, p.chunks,
                           p.chunks_per_split);
        }
        ProfScope prof(DEEPIPR_K_CONV_WGRAD_REDUCE, st);
        DEEPIPR_LAUNCH(prof, k_conv_stem_wgrad_reduce, dim3(Co), dim3(256), st, part, dW, Co, p.tiles_co,
                       p.splits);
        return check_launch("conv_wgrad");
    }
    if (p.cfg == 3700) {                                                 // the ImageNet stem (deepipr_conv_stem7.inc)
        if (rank2) return fail(DEEPIPR_EUNSUPPORTED, "conv_wgrad: no fused rank-2 term for a 3-channel input");
        {
            ProfScope prof(DEEPIPR_K_CONV_WGRAD, st);
            prof.bytes = 2.0 * Co * 147.0 * static_cast<double>(N) * (H / 2) * (W / 2);
            DEEPIPR_LAUNCH(prof, k_conv_stem7_wgrad, dim3(p.splits), dim3(512), st, x, dy, part, N, H, p.chunks_per_split);
        }
        ProfScope prof(DEEPIPR_K_CONV_WGRAD_REDUCE, st);
        DEEPIPR_LAUNCH(prof, k_conv_stem7_wgrad_reduce, dim3(64), dim3(256), st, part, dW, p.splits);
        return check_launch("conv_wgrad");
    }
    if (p.cfg >= 6000) {                                               // 1x1 stride 1 (deepipr_conv_1x1.inc)
        ProfScope prof(DEEPIPR_K_CONV1X1_WGRAD, st);
        prof.bytes = 2.0 * Co * Ci * static_cast<double>(N) * H * W;    // FLOPs
        const int g1 = p.splits * p.wg_tiles_co * p.wg_tiles_ci;
#define DEEPIPR_G1(BM, BN, HWP, NI, VEC)                                                                              \
    DEEPIPR_LAUNCH(prof, (k_conv1x1_wgrad<G1Cfg<BM, BN, HWP, NI, VEC>>), dim3(g1), dim3(512), st, x, dy, part, N, Ci, Co, \
                   H * W, p.wg_tiles_co, p.wg_tiles_ci, p.chunks, p.chunks_per_split)
#define DEEPIPR_G1_GEO(CODE, HWP, NI, VEC)                                                                            \
    case CODE + 11: DEEPIPR_G1(1, 1, HWP, NI, VEC); break;                                                            \
    case CODE + 12: DEEPIPR_G1(1, 2, HWP, NI, VEC); break;                                                            \
    case CODE + 21: DEEPIPR_G1(2, 1, HWP, NI, VEC); break;                                                            \
    case CODE + 22: DEEPIPR_G1(2, 2, HWP, NI, VEC); break;
        switch (p.cfg) {
            DEEPIPR_G1_GEO(6100, 64, 1, 4)
            DEEPIPR_G1_GEO(6200, 56, 1, 4)
            DEEPIPR_G1_GEO(6300, 28, 2, 4)
            DEEPIPR_G1_GEO(6400, 49, 1, 1)
            default: return fail(DEEPIPR_EUNSUPPORTED, "conv_wgrad: no instance");
        }
#undef DEEPIPR_G1_GEO
#undef DEEPIPR_G1
    } else if (p.cfg >= 5000) {                                        // Winograd F(3x3, 2x2)
        ProfScope prof(DEEPIPR_K_CONV_WINO_WGRAD, st);
        prof.bytes = 2.0 * Co * Ci * 16.0 * static_cast<double>(N) * ((H + 1) / 2) * ((W + 1) / 2);          // EXECUTED FLOPs (direct: x 2.25)
        const bool timed = prof.a && !prof.used;
        if (!dipr_launch_wgrad_wino(p.cfg - 5000 + (p.pairs ? 10000 : 0), x, dy, part, N, Ci, Co, H, p.tiles_co, p.tiles_ci, p.chunks, p.chunks_per_split,
                                    grid, st, timed ? prof.a : nullptr, timed ? prof.b : nullptr))
            return fail(DEEPIPR_EUNSUPPORTED, "conv_wgrad: no instance");
        if (timed) prof.used = true;
    } else {
        ProfScope prof(p.cfg >= 4000 ? DEEPIPR_K_CONV_WGRAD_B3 : DEEPIPR_K_CONV_WGRAD, st);
        prof.bytes = 2.0 * Co * Ci * p.taps * static_cast<double>(N) * (H / stride) * (W / stride);     // FLOPs, not bytes: this kernel's roofline is the MFMA peak
#define DEEPIPR_WGRAD_1X1(...)                                                                                        \
    DEEPIPR_LAUNCH(prof, (k_conv1x1s2_wgrad<W1Cfg<__VA_ARGS__>>), dim3(grid), dim3(256), st, x, dy, part, Ci, Co, H,   \
                   p.tiles_co, p.tiles_ci, p.chunks, p.chunks_per_split)
        switch (p.cfg) {
            case 132: launch_wgrad<WgCfg<1, 32, 2, 1>>(prof, p, x, dy, part, Ci, Co, H, st); break;
            case 116: launch_wgrad<WgCfg<1, 16, 4, 1>>(prof, p, x, dy, part, Ci, Co, H, st); break;
            case 108: launch_wgrad<WgCfg<1, 8, 8, 1>>(prof, p, x, dy, part, Ci, Co, H, st); break;
            case 104: launch_wgrad<WgCfg<1, 4, 4, 2>>(prof, p, x, dy, part, Ci, Co, H, st); break;
            case 2132: launch_wgrad<WgCfg<1, 32, 4, 1, 32>>(prof, p, x, dy, part, Ci, Co, H, st); break;
            case 2116: launch_wgrad<WgCfg<1, 16, 8, 1, 32>>(prof, p, x, dy, part, Ci, Co, H, st); break;
            case 2108: launch_wgrad<WgCfg<1, 8, 8, 2, 32>>(prof, p, x, dy, part, Ci, Co, H, st); break;
            case 2104: launch_wgrad<WgCfg<1, 4, 4, 4, 32>>(prof, p, x, dy, part, Ci, Co, H, st); break;
case 4032: launch_wgrad_b3<WbCfg<32, 2, 1>>(prof, p, x, dy, part, Ci, Co, H, st); break;
            case 4016: launch_wgrad_b3<WbCfg<16, 4, 1>>(prof, p, x, dy, part, Ci, Co, H, st); break;
            case 4008: launch_wgrad_b3<WbCfg<8, 8, 1>>(prof, p, x, dy, part, Ci, Co, H, st); break;
            case 216: launch_wgrad<WgCfg<2, 16, 2, 1>>(prof, p, x, dy, part, Ci, Co, H, st); break;
            case 208: launch_wgrad<WgCfg<2, 8, 4, 1>>(prof, p, x, dy, part, Ci, Co, H, st); break;
            case 204: launch_wgrad<WgCfg<2, 4, 4, 1>>(prof, p, x, dy, part, Ci, Co, H, st); break;
            case 1216: DEEPIPR_WGRAD_1X1(16, 4, 1); break;
            case 1208: DEEPIPR_WGRAD_1X1(8, 8, 1); break;
            case 1204: DEEPIPR_WGRAD_1X1(4, 4, 2); break;
            default: return fail(DEEPIPR_EUNSUPPORTED, "conv_wgrad: no instance");
        }
#undef DEEPIPR_WGRAD_1X1
    }
    const int tiles = p.tiles_co * p.tiles_ci;
    ProfScope prof(DEEPIPR_K_CONV_WGRAD_REDUCE, st);
    prof.bytes = 4.0 * (static_cast<double>(p.splits) + 1.0) * tiles * 64 * p.cit * p.taps;
#define DEEPIPR_WGRAD_REDUCE(SG, T, TW, GRID, BLOCK)                                                                   \
    do {                                                                                                              \
        if (rank2) DEEPIPR_LAUNCH(prof, (k_conv_wgrad_reduce<SG, true, T, TW>), dim3(GRID), dim3(BLOCK), st, part, dW, \
                                  Ci, p.tiles_co, tiles, p.splits, dgamma, dbeta, m);                                 \
        else DEEPIPR_LAUNCH(prof, (k_conv_wgrad_reduce<SG, false, T, TW>), dim3(GRID), dim3(BLOCK), st, part, dW, Ci,  \
                            p.tiles_co, tiles, p.splits, dgamma, dbeta, m);                                           \
    } while (0)
    // rows (64 lanes x 16 bytes) of a partial tile: 4 * tile waves * taps
    if (p.taps == 1) {
        if (p.splits >= 2) DEEPIPR_WGRAD_REDUCE(4, 1, 4, tiles * 16, 256);
        else DEEPIPR_WGRAD_REDUCE(1, 1, 4, tiles * 4, 256);
    } else if (p.cit == 32) {
        if (p.splits >= 64) DEEPIPR_WGRAD_REDUCE(16, 9, 2, tiles * 72, 1024);
        else if (p.splits >= 2) DEEPIPR_WGRAD_REDUCE(4, 9, 2, tiles * 72, 256);
        else DEEPIPR_WGRAD_REDUCE(1, 9, 2, tiles * 18, 256);
    } else if (p.splits >= 64) DEEPIPR_WGRAD_REDUCE(16, 9, 4, tiles * 144, 1024);
    else if (p.splits >= 2) DEEPIPR_WGRAD_REDUCE(4, 9, 4, tiles * 144, 256);
    else DEEPIPR_WGRAD_REDUCE(1, 9, 4, tiles * 36, 256);
#undef DEEPIPR_WGRAD_REDUCE
    return check_launch("conv_wgrad");
}


}  // extern "C"

// =============================================================================================
// data convolution: forward and backward-data on the fp32 matrix cores (deepipr_conv_fwd.inc)
// =============================================================================================
namespace {
template <class C, bool DGRAD>
void launch_gemm(ProfScope &prof, const FwPlan &p, const float *wgt, const float *in, float *out, int Cin, int M, int H,
                 hipStream_t st, float *ws) {
    if (ws) DEEPIPR_LAUNCH(prof, (k_conv_gemm<C, DGRAD>), dim3(p.grid * p.splits), dim3(256), st, wgt, in, out, Cin, M, H, p.bands,
                           ws, p.grid, p.cps, p.slab);
    else DEEPIPR_LAUNCH(prof, (k_conv_gemm<C, DGRAD>), dim3(p.grid), dim3(256), st, wgt, in, out, Cin, M, H, p.bands,
                        static_cast<float *>(nullptr), p.grid, Cin, p.slab);
}

// the plan a 3x3 / 1x1 forward-shaped GEMM call takes: Winograd where it applies and is switched on, else the direct form
FwPlan plan_conv_any(int N, int C, int M, int H, int W, int k, int stride, int pad) {
    if (conv_algo() == 1) {
        const FwPlan p = dipr_plan_conv_wino(N, C, M, H, W, k, stride, pad);
        if (p.cfg) return p;
    }
    return plan_conv_gemm(N, C, M, H, W, k, stride, pad);
}

template <bool DGRAD>
int conv_wino(const FwPlan &p, const float *wgt, const float *in, float *out, int N, int Cin, int M, int H, int W,
              hipStream_t st, const char *what, void *workspace, size_t workspace_bytes, bool pre = false) {
    if (!aligned16(wgt) || !aligned16(in) || !aligned16(out) || !aligned16(workspace))
        return fail(DEEPIPR_EINVAL, "%s: pointers must be 16-byte aligned", what);
    // split K needs its slabs: without a workspace one workgroup walks all of K
    float *ws = (p.splits > 1 && workspace && workspace_bytes >= p.splits * p.slab * sizeof(float)) ? static_cast<float *>(workspace) : nullptr;
    {
        ProfScope prof(DGRAD ? DEEPIPR_K_CONV_WINO_DGRAD : DEEPIPR_K_CONV_WINO_FWD, st);
        prof.bytes = 2.0 * M * Cin * 16.0 * static_cast<double>(N) * (H / 2) * (W / 2);          // EXECUTED FLOPs (direct: x 2.25)
        const bool timed = prof.a && !prof.used;
        const bool ok = pre ? dipr_launch_conv_wino_pre(p, wgt, in, out, N, Cin, M, H, ws, st, timed ? prof.a : nullptr, timed ? prof.b : nullptr)
                            : dipr_launch_conv_wino(p, DGRAD, wgt, in, out, N, Cin, M, H, ws, st, timed ? prof.a : nullptr, timed ? prof.b : nullptr);
        if (!ok) return fail(DEEPIPR_EUNSUPPORTED, "%s: no instance", what);
        if (timed) prof.used = true;
    }
    if (ws) {
        ProfScope prof(DEEPIPR_K_CONV_SPLIT_SUM, st);
        prof.bytes = 4.0 * (p.splits + 1.0) * p.slab;
        const size_t n4 = p.slab / 4;
        DEEPIPR_LAUNCH(prof, k_conv_sum_slabs, dim3(static_cast<unsigned>((n4 + 255) / 256)), dim3(256), st,
                       reinterpret_cast<const f32x4 *>(ws), reinterpret_cast<f32x4 *>(out), n4, p.splits);
    }
    return check_launch(what);
}

template <bool DGRAD>
int conv_gemm(int slot, const float *wgt, const float *in, float *out, int N, int Cin, int M, int H, int W, int k, int stride,
              int pad, hipStream_t st, const char *what, void *workspace, size_t workspace_bytes) {
    const FwPlan p = plan_conv_any(N, Cin, M, H, W, k, stride, pad);
    if (p.cfg >= 1000) return conv_wino<DGRAD>(p, wgt, in, out, N, Cin, M, H, W, st, what, workspace, workspace_bytes);
    if (!p.cfg) return fail(DEEPIPR_EUNSUPPORTED, "%s: shape outside the kernel (use the library's convolution)", what);
    if (p.cfg == 700 || p.cfg == 600) {                                  // the stems: 7x7 stride 2 (ImageNet), 3x3 stride 1 (CIFAR); deepipr_conv_stem7.inc
        if (DGRAD) return fail(DEEPIPR_EUNSUPPORTED, "%s: the stem has no backward-data instance (its input is the image)", what);
        if (!aligned16(wgt) || !aligned16(in) || !aligned16(out)) return fail(DEEPIPR_EINVAL, "%s: pointers must be 16-byte aligned", what);
        ProfScope prof(slot, st);
        prof.bytes = 2.0 * M * Cin * k * k * static_cast<double>(N) * (H / stride) * (W / stride);     // useful FLOPs (the padded taps / idle lanes not counted)
        if (p.cfg == 700) DEEPIPR_LAUNCH(prof, k_conv_stem7_fwd, dim3(p.grid), dim3(256), st, wgt, in, out, N, H);
        else DEEPIPR_LAUNCH(prof, k_conv_stem3_fwd, dim3(p.grid), dim3(256), st, wgt, in, out, N, H);
        return check_launch(what);
    }
    if (p.cfg >= 900) {                                                  // 1x1 stride 1: one GEMM over NCHW (deepipr_conv_1x1.inc)
        if (!aligned16(wgt) || !aligned16(in) || !aligned16(out) || !aligned16(workspace))
            return fail(DEEPIPR_EINVAL, "%s: pointers must be 16-byte aligned", what);
        const int total = N * H * W, bm = (p.cfg - 900) / 10, tiles_m = M / (64 * bm);
        // the stream-K tail needs its workspace: without one (the workspace-free entry points) every tile is a whole tile
        const bool tail = p.splits > 1 && workspace && workspace_bytes >= 2 * p.slab * sizeof(float);
        const int tail_tiles = tail ? p.bands : 0, full_tiles = p.grid - tail_tiles;
        const int tail_wgs = tail ? static_cast<int>(p.slab / (64 * bm * 128)) : 0;
        float *ws1 = static_cast<float *>(workspace);
        {
            ProfScope prof(DGRAD ? DEEPIPR_K_CONV1X1_DGRAD : DEEPIPR_K_CONV1X1_FWD, st);
            prof.bytes = 2.0 * M * Cin * static_cast<double>(N) * H * W;     // FLOPs
#define DEEPIPR_F1(BM, VEC)                                                                                           \
    DEEPIPR_LAUNCH(prof, (k_conv1x1_gemm<F1Cfg<BM, 2, DGRAD, VEC>>), dim3(full_tiles + tail_wgs), dim3(256), st, wgt, in, out, M, Cin, \
                   H * W, total, tiles_m, full_tiles, tail_tiles, tail_wgs, ws1)
            switch (p.cfg) {
                case 910: DEEPIPR_F1(1, 4); break;
                case 911: DEEPIPR_F1(1, 1); break;
                case 920: DEEPIPR_F1(2, 4); break;
                case 921: DEEPIPR_F1(2, 1); break;
                default: return fail(DEEPIPR_EUNSUPPORTED, "%s: no instance", what);
            }
#undef DEEPIPR_F1
        }
        if (tail) {
            ProfScope prof(DEEPIPR_K_CONV_SPLIT_SUM, st);
            prof.bytes = 4.0 * (64.0 * bm * 128) * (tail_wgs + 2.0 * tail_tiles);
            const dim3 grid(tail_tiles * (64 * bm * 128 / 4 / 256));
            const bool v4 = (H * W) % 4 == 0;
#define DEEPIPR_F1S(TM, V4)                                                                                           \
    DEEPIPR_LAUNCH(prof, (k_conv1x1_tail_sum<TM, 128, V4>), grid, dim3(256), st, ws1, out, M, H * W, total, tiles_m, full_tiles,  \
                   tail_tiles, tail_wgs, p.cps)
            if (bm == 2) { if (v4) DEEPIPR_F1S(128, true); else DEEPIPR_F1S(128, false); }
            else { if (v4) DEEPIPR_F1S(64, true); else DEEPIPR_F1S(64, false); }
#undef DEEPIPR_F1S
        }
        return check_launch(what);
    }
    if (!aligned16(wgt) || !aligned16(in) || !aligned16(out) || !aligned16(workspace))
        return fail(DEEPIPR_EINVAL, "%s: pointers must be 16-byte aligned", what);
    // split K needs its slabs: without a workspace (the entry points of ABI v7) the plain form runs
    float *ws = (p.splits > 1 && workspace && workspace_bytes >= p.splits * p.slab * sizeof(float)) ? static_cast<float *>(workspace) : nullptr;
    {
    ProfScope prof(slot, st);
    prof.bytes = 2.0 * M * Cin * k * k * static_cast<double>(N) * (H / stride) * (W / stride);       // FLOPs
    switch (p.cfg) {
        case 11: launch_gemm<FwCfg<1, 9, 32, 8, 1, 4, 1, 8>, DGRAD>(prof, p, wgt, in, out, Cin, M, H, st, ws); break;
        case 21: launch_gemm<FwCfg<1, 9, 32, 2, 1, 1, 4, 8>, DGRAD>(prof, p, wgt, in, out, Cin, M, H, st, ws); break;
        case 31: launch_gemm<FwCfg<1, 9, 16, 8, 1, 2, 2, 8>, DGRAD>(prof, p, wgt, in, out, Cin, M, H, st, ws); break;
        case 41: launch_gemm<FwCfg<1, 9, 16, 4, 1, 1, 4, 8>, DGRAD>(prof, p, wgt, in, out, Cin, M, H, st, ws); break;
        case 51: launch_gemm<FwCfg<1, 9, 8, 8, 1, 1, 4, 8>, DGRAD>(prof, p, wgt, in, out, Cin, M, H, st, ws); break;
        case 61: launch_gemm<FwCfg<1, 9, 4, 4, 4, 1, 4, 8>, DGRAD>(prof, p, wgt, in, out, Cin, M, H, st, ws); break;
        default:
            if (DGRAD) return fail(DEEPIPR_EUNSUPPORTED, "%s: backward-data of a stride-2 convolution is not a gather", what);
            switch (p.cfg) {
                case 42: launch_gemm<FwCfg<2, 9, 16, 4, 1, 1, 4, 8>, false>(prof, p, wgt, in, out, Cin, M, H, st, ws); break;
                case 52: launch_gemm<FwCfg<2, 9, 8, 8, 1, 1, 4, 8>, false>(prof, p, wgt, in, out, Cin, M, H, st, ws); break;
                case 62: launch_gemm<FwCfg<2, 9, 4, 4, 4, 1, 4, 8>, false>(prof, p, wgt, in, out, Cin, M, H, st, ws); break;
                case 142: launch_gemm<FwCfg<2, 1, 16, 4, 1, 1, 4, 16>, false>(prof, p, wgt, in, out, Cin, M, H, st, ws); break;
                case 152: launch_gemm<FwCfg<2, 1, 8, 8, 1, 1, 4, 16>, false>(prof, p, wgt, in, out, Cin, M, H, st, ws); break;
                case 162: launch_gemm<FwCfg<2, 1, 4, 4, 4, 1, 4, 16>, false>(prof, p, wgt, in, out, Cin, M, H, st, ws); break;
                default: return fail(DEEPIPR_EUNSUPPORTED, "%s: no instance", what);
            }
    }
    }
    if (ws) {
        ProfScope prof(DEEPIPR_K_CONV_SPLIT_SUM, st);
        prof.bytes = 4.0 * (p.splits + 1.0) * p.slab;
        const size_t n4 = p.slab / 4;
        DEEPIPR_LAUNCH(prof, k_conv_sum_slabs, dim3(static_cast<unsigned>((n4 + 255) / 256)), dim3(256), st,
                       reinterpret_cast<const f32x4 *>(ws), reinterpret_cast<f32x4 *>(out), n4, p.splits);
    }
    return check_launch(what);
}

template <class C>
void launch_dgrad_s2(ProfScope &prof, const FwPlan &p, const float *wgt, const float *dy, float *dx, int Co, int Ci, int oh,
                     hipStream_t st) {
    DEEPIPR_LAUNCH(prof, (k_conv_dgrad_s2<C>), dim3(p.grid), dim3(256), st, wgt, dy, dx, Co, Ci, oh, p.bands);
}

int conv_dgrad_s2(const float *dy, const float *w, float *dx, int N, int Ci, int Co, int H, int W, int k, int pad, hipStream_t st) {
    if (H % 2 || W % 2) return fail(DEEPIPR_EUNSUPPORTED, "conv_dgrad: odd map");
    const int oh = H / 2, ow = W / 2;
    const FwPlan p = plan_conv_dgrad_s2(N, Co, Ci, oh, ow, k, pad);
    if (!p.cfg) return fail(DEEPIPR_EUNSUPPORTED, "conv_dgrad: shape outside the kernel (use the library's backward-data)");
    if (!aligned16(w) || !aligned16(dy) || !aligned16(dx)) return fail(DEEPIPR_EINVAL, "conv_dgrad: pointers must be 16-byte aligned");
    ProfScope prof(DEEPIPR_K_CONV_DGRAD, st);
    prof.bytes = 2.0 * Ci * Co * k * k * static_cast<double>(N) * oh * ow;       // FLOPs
#define DEEPIPR_DGRAD_X4(...)                                                                                         \
    DEEPIPR_LAUNCH(prof, (k_conv_dgrad_s2x4<FwCfg<__VA_ARGS__>>), dim3(p.grid), dim3(256), st, w, dy, dx, Co, Ci, oh, p.bands)
    // chunks of 16 channels (round 6): a chunk of 8 is 18-36 MFMAs per wavefront between two barriers -- the barrier and the operand
    // latencies behind it cost as much as the MFMAs; 16 halves the barriers per MFMA at one workgroup per CU (config R: the family
    // 282 -> 246 us per step).  DEEPIPR_DGRAD_S2_CK16=0: the 8-channel instances (measurement)
    static const int ck16 = getenv("DEEPIPR_DGRAD_S2_CK16") ? atoi(getenv("DEEPIPR_DGRAD_S2_CK16")) : 1;
    if (ck16 && Co % 16 == 0) {
        switch (p.cfg) {
            case 11: DEEPIPR_DGRAD_X4(1, 9, 16, 8, 1, 4, 1, 16, 32); return check_launch("conv_dgrad");
            case 41: DEEPIPR_DGRAD_X4(1, 9, 8, 8, 1, 2, 2, 16, 32); return check_launch("conv_dgrad");
            case 61: DEEPIPR_DGRAD_X4(1, 9, 4, 4, 2, 1, 4, 16, 32); return check_launch("conv_dgrad");
            default: break;
        }
    }
    switch (p.cfg) {                                               // W, RB, NIB, POSW, KG, CK, 32 positions per wave
        case 11: DEEPIPR_DGRAD_X4(1, 9, 16, 8, 1, 4, 1, 8, 32); break;      // 128 positions per workgroup
        case 21: DEEPIPR_DGRAD_X4(1, 9, 16, 4, 1, 2, 2, 8, 32); break;      // 64
        case 31: DEEPIPR_DGRAD_X4(1, 9, 16, 2, 1, 1, 4, 8, 32); break;      // 32
        case 41: DEEPIPR_DGRAD_X4(1, 9, 8, 8, 1, 2, 2, 8, 32); break;
        case 51: DEEPIPR_DGRAD_X4(1, 9, 8, 4, 1, 1, 4, 8, 32); break;
        case 61: DEEPIPR_DGRAD_X4(1, 9, 4, 4, 2, 1, 4, 8, 32); break;
        case 141: launch_dgrad_s2<FwCfg<1, 1, 16, 4, 1, 1, 4, 16>>(prof, p, w, dy, dx, Co, Ci, oh, st); break;
        case 151: launch_dgrad_s2<FwCfg<1, 1, 8, 8, 1, 1, 4, 16>>(prof, p, w, dy, dx, Co, Ci, oh, st); break;
        case 161: launch_dgrad_s2<FwCfg<1, 1, 4, 4, 4, 1, 4, 16>>(prof, p, w, dy, dx, Co, Ci, oh, st); break;
        default: return fail(DEEPIPR_EUNSUPPORTED, "conv_dgrad: no instance");
    }
#undef DEEPIPR_DGRAD_X4
    return check_launch("conv_dgrad");
}
}  // namespace

extern "C" {

int deepipr_conv_supported(int N, int Ci, int Co, int H, int W, int k, int stride, int pad, int direction) {
    if (direction == 0) return plan_conv_any(N, Ci, Co, H, W, k, stride, pad).cfg ? 1 : 0;
    if (direction != 1) return 0;
    if (stride == 1) return plan_conv_any(N, Co, Ci, H, W, k, stride, pad).cfg ? 1 : 0;
    if (stride == 2 && H % 2 == 0 && W % 2 == 0) return plan_conv_dgrad_s2(N, Co, Ci, H / 2, W / 2, k, pad).cfg ? 1 : 0;
    return 0;
}

size_t deepipr_conv_workspace_bytes(int N, int Ci, int Co, int H, int W, int k, int stride, int pad, int direction) {
    if (direction != 0 && !(direction == 1 && stride == 1)) return 0;
    const FwPlan p = direction == 0 ? plan_conv_any(N, Ci, Co, H, W, k, stride, pad) : plan_conv_any(N, Co, Ci, H, W, k, stride, pad);
    return (p.cfg && p.splits > 1) ? p.splits * p.slab * sizeof(float) : 0;
}

int deepipr_conv_set_algo(int algo) {
    if (algo != 0 && algo != 1) return fail(DEEPIPR_EINVAL, "conv_set_algo: 0 (direct implicit GEMM) or 1 (Winograd F(2x2, 3x3))");
    g_conv_algo = algo;
    return DEEPIPR_OK;
}

int deepipr_conv_get_algo(void) { return conv_algo(); }

#ifdef DEEPIPR_TRACE
// measurement build only: device buffer of 64 u64 per workgroup for the Winograd kernels' phase stamps (nullptr: off)
int deepipr_debug_wino_trace(unsigned long long *device_buffer) {
    return dipr_wino_set_trace(device_buffer) ? DEEPIPR_OK : fail(DEEPIPR_ELAUNCH, "debug_wino_trace: hipMemcpyToSymbol failed");
}
#endif

int deepipr_conv_algo_of(int N, int Ci, int Co, int H, int W, int k, int stride, int pad, int direction) {
    if (direction != 0 && !(direction == 1 && stride == 1)) return 0;
    const FwPlan p = direction == 0 ? plan_conv_any(N, Ci, Co, H, W, k, stride, pad) : plan_conv_any(N, Co, Ci, H, W, k, stride, pad);
    return p.cfg >= 1000 ? 1 : 0;
}

int deepipr_conv_fwd_ws(const float *x, const float *w, float *y, int N, int Ci, int Co, int H, int W, int k, int stride, int pad,
                        void *workspace, size_t workspace_bytes, void *stream) {
    if (!x || !w || !y) return fail(DEEPIPR_EINVAL, "conv_fwd: null pointer");
    return conv_gemm<false>(DEEPIPR_K_CONV_FWD, w, x, y, N, Ci, Co, H, W, k, stride, pad, static_cast<hipStream_t>(stream),
                            "conv_fwd", workspace, workspace_bytes);
}

int deepipr_conv_fwd(const float *x, const float *w, float *y, int N, int Ci, int Co, int H, int W, int k, int stride, int pad,
                     void *stream) {
    return deepipr_conv_fwd_ws(x, w, y, N, Ci, Co, H, W, k, stride, pad, nullptr, 0, stream);
}

int deepipr_conv_dgrad_ws(const float *dy, const float *w, float *dx, int N, int Ci, int Co, int H, int W, int k, int stride,
                          int pad, void *workspace, size_t workspace_bytes, void *stream) {
    if (!dy || !w || !dx) return fail(DEEPIPR_EINVAL, "conv_dgrad: null pointer");
    if (stride == 2) return conv_dgrad_s2(dy, w, dx, N, Ci, Co, H, W, k, pad, static_cast<hipStream_t>(stream));
    if (stride != 1) return fail(DEEPIPR_EUNSUPPORTED, "conv_dgrad: stride 1 or 2 (use the library's backward-data)");
    return conv_gemm<true>(DEEPIPR_K_CONV_DGRAD, w, dy, dx, N, Co, Ci, H, W, k, stride, pad, static_cast<hipStream_t>(stream),
                           "conv_dgrad", workspace, workspace_bytes);
}

int deepipr_conv_dgrad(const float *dy, const float *w, float *dx, int N, int Ci, int Co, int H, int W, int k, int stride,
                       int pad, void *stream) {
    return deepipr_conv_dgrad_ws(dy, w, dx, N, Ci, Co, H, W, k, stride, pad, nullptr, 0, stream);
}

size_t deepipr_conv_wino_image_bytes(int Co, int Ci) { return dipr_wino_image_floats(Co, Ci) * sizeof(float); }

int deepipr_conv_wino_max_layers(void) { return dipr_wino_max_layers(); }

int deepipr_conv_wino_transform_multi(const DeepiprWinoLayer *layers, int n, void *stream) {
    if (!layers || n <= 0 || n > dipr_wino_max_layers())
        return fail(DEEPIPR_EINVAL, "conv_wino_transform_multi: 1..%d layers per call", dipr_wino_max_layers());
    DiprWinoLayer L[DEEPIPR_WINO_MAX_LAYERS];
    double bytes = 0.0;
    for (int i = 0; i < n; ++i) {
        const DeepiprWinoLayer &l = layers[i];
        if (!l.W || !(l.Uf || l.Ud) || !dipr_wino_image_floats(l.Co, l.Ci))
            return fail(DEEPIPR_EINVAL, "conv_wino_transform_multi: bad layer %d (Co, Ci multiples of 32, W and an image)", i);
        if (!aligned16(l.W) || !aligned16(l.Uf) || !aligned16(l.Ud))
            return fail(DEEPIPR_EINVAL, "conv_wino_transform_multi: pointers must be 16-byte aligned");
        L[i] = DiprWinoLayer{l.W, l.Uf, l.Ud, l.Co, l.Ci};
        bytes += static_cast<double>(l.Co) * l.Ci * (36.0 + 66.0 * ((l.Uf ? 1 : 0) + (l.Ud ? 1 : 0)));
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    ProfScope prof(DEEPIPR_K_CONV_WINO_WEIGHTS, st);
    prof.bytes = bytes;
    const bool timed = prof.a && !prof.used;
    if (!dipr_launch_wino_weights(L, n, st, timed ? prof.a : nullptr, timed ? prof.b : nullptr))
        return fail(DEEPIPR_EUNSUPPORTED, "conv_wino_transform_multi: no instance");
    if (timed) prof.used = true;
    return check_launch("conv_wino_transform_multi");
}

int deepipr_conv_fwd_pre(const float *x, const float *image, float *y, int N, int Ci, int Co, int H, int W, void *workspace,
                         size_t workspace_bytes, void *stream) {
    if (!x || !image || !y) return fail(DEEPIPR_EINVAL, "conv_fwd_pre: null pointer");
    const FwPlan p = plan_conv_any(N, Ci, Co, H, W, 3, 1, 1);
    if (p.cfg < 1000 || !dipr_wino_image_floats(Co, Ci))
        return fail(DEEPIPR_EUNSUPPORTED, "conv_fwd_pre: this call does not take the Winograd kernel, or the weight has no image");
    return conv_wino<false>(p, image, x, y, N, Ci, Co, H, W, static_cast<hipStream_t>(stream), "conv_fwd_pre", workspace,
                            workspace_bytes, true);
}

int deepipr_conv_dgrad_pre(const float *dy, const float *image, float *dx, int N, int Ci, int Co, int H, int W, void *workspace,
                           size_t workspace_bytes, void *stream) {
    if (!dy || !image || !dx) return fail(DEEPIPR_EINVAL, "conv_dgrad_pre: null pointer");
    const FwPlan p = plan_conv_any(N, Co, Ci, H, W, 3, 1, 1);
    if (p.cfg < 1000 || !dipr_wino_image_floats(Co, Ci))
        return fail(DEEPIPR_EUNSUPPORTED, "conv_dgrad_pre: this call does not take the Winograd kernel, or the weight has no image");
    return conv_wino<true>(p, image, dy, dx, N, Co, Ci, H, W, static_cast<hipStream_t>(stream), "conv_dgrad_pre", workspace,
                           workspace_bytes, true);
}

}  // extern "C"


int dipr_device_cu_count() { return device_cu_count(); }
