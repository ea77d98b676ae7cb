// deepipr_wino.hip -- second translation unit of libdeepipr_hip.so: the Winograd F(2x2, 3x3) forward / backward-data kernels
// (deepipr_conv_wino.inc), their planner and launcher.  Built with -fno-slp-vectorize (Makefile; deepipr_conv_plan.h says why).
// The C ABI entry points that reach this code are in deepipr_hip.hip (deepipr_conv_fwd_ws / deepipr_conv_dgrad_ws).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "deepipr_conv_plan.h"

namespace {
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
inline int device_cu_count() { return dipr_device_cu_count(); }
#include "deepipr_conv_wino.inc"
#include "deepipr_conv_wino_wgrad.inc"

template <class C, bool DGRAD, bool PRE>
void launch(const FwPlan &p, const float *wgt, const float *in, float *out, int N, int Cin, int M, int H, float *ws, hipStream_t st,
            hipEvent_t a, hipEvent_t b) {
    const dim3 grid(ws ? p.grid * p.splits : p.grid), block(C::NTH);
    const int cps = ws ? p.cps : Cin / C::CK;
    const int xcd = dipr_xcd_remap();
    if (a) hipExtLaunchKernelGGL((k_conv_wino<C, DGRAD, PRE>), grid, block, 0, st, a, b, 0, wgt, in, out, N, Cin, M, H, p.bands, ws, p.grid,
                                 cps, p.slab, xcd);
    else hipLaunchKernelGGL((k_conv_wino<C, DGRAD, PRE>), grid, block, 0, st, wgt, in, out, N, Cin, M, H, p.bands, ws, p.grid, cps, p.slab, xcd);
}

template <bool DGRAD, bool PRE>
bool dispatch(const FwPlan &p, const float *wgt, const float *in, float *out, int N, int Cin, int M, int H, float *ws, hipStream_t st,
              hipEvent_t a, hipEvent_t b) {
#define DIPR_WINO(WW, TBR, NIB)                                                                                       \
    switch (p.cfg % 100) {                                                                                            \
        case 21: launch<WnCfg<WW, TBR, NIB, 2, 1>, DGRAD, PRE>(p, wgt, in, out, N, Cin, M, H, ws, st, a, b); return true;  \
        case 22: launch<WnCfg<WW, TBR, NIB, 2, 2>, DGRAD, PRE>(p, wgt, in, out, N, Cin, M, H, ws, st, a, b); return true;  \
        case 11: launch<WnCfg<WW, TBR, NIB, 1, 1>, DGRAD, PRE>(p, wgt, in, out, N, Cin, M, H, ws, st, a, b); return true;  \
        case 12: launch<WnCfg<WW, TBR, NIB, 1, 2>, DGRAD, PRE>(p, wgt, in, out, N, Cin, M, H, ws, st, a, b); return true;  \
        default: return false;                                                                                        \
    }
    switch ((p.cfg / 100) % 10) {
        case 3: DIPR_WINO(32, 2, 1)
        case 2: DIPR_WINO(16, 4, 1)
        case 1: DIPR_WINO(8, 4, 2)
        case 0: DIPR_WINO(4, 2, 8)
        case 7: DIPR_WINO(56, 1, 1)
        case 6: DIPR_WINO(28, 2, 1)
        case 5: DIPR_WINO(14, 4, 1)
        case 4: DIPR_WINO(7, 4, 2)
        default: return false;
    }
#undef DIPR_WINO
}
}  // namespace

bool dipr_launch_wgrad_wino(int width, const float *x, const float *dy, float *part, int N, int Ci, int Co, int H, int tiles_co,
                            int tiles_ci, int chunks, int chunks_per_split, int grid, hipStream_t st, hipEvent_t a, hipEvent_t b) {
    if (width >= 10000) {                                       // two K ranges per workgroup (k_conv_wino_wgrad2): width + 10000
#define DIPR_WW2(WW, TBR, NIB)                                                                                         \
    do {                                                                                                              \
        if (a) hipExtLaunchKernelGGL((k_conv_wino_wgrad2<WwCfg<WW, TBR, NIB>>), dim3(grid), dim3(512), 0, st, a, b, 0, x, dy, part, N, \
                                     Ci, Co, H, tiles_co, tiles_ci, chunks, chunks_per_split, dipr_xcd_remap());      \
        else hipLaunchKernelGGL((k_conv_wino_wgrad2<WwCfg<WW, TBR, NIB>>), dim3(grid), dim3(512), 0, st, x, dy, part, N, Ci, Co, H,  \
                                tiles_co, tiles_ci, chunks, chunks_per_split, dipr_xcd_remap());                      \
        return true;                                                                                                  \
    } while (0)
        switch (width - 10000) {
            case 32: DIPR_WW2(32, 1, 1);
            case 16: DIPR_WW2(16, 2, 1);
            case 8: DIPR_WW2(8, 4, 1);
            case 4: DIPR_WW2(4, 2, 4);
            default: return false;
        }
#undef DIPR_WW2
    }
#define DIPR_WW(WW, TBR, NIB)                                                                                          \
    do {                                                                                                              \
        if (a) hipExtLaunchKernelGGL((k_conv_wino_wgrad<WwCfg<WW, TBR, NIB>>), dim3(grid), dim3(256), 0, st, a, b, 0, x, dy, part, N, \
                                     Ci, Co, H, tiles_co, tiles_ci, chunks, chunks_per_split, dipr_xcd_remap());      \
        else hipLaunchKernelGGL((k_conv_wino_wgrad<WwCfg<WW, TBR, NIB>>), dim3(grid), dim3(256), 0, st, x, dy, part, N, Ci, Co, H,   \
                                tiles_co, tiles_ci, chunks, chunks_per_split, dipr_xcd_remap());                      \
        return true;                                                                                                  \
    } while (0)
    switch (width) {
        case 32: DIPR_WW(32, 1, 1);
        case 16: DIPR_WW(16, 2, 1);
        case 8: DIPR_WW(8, 4, 1);
        case 4: DIPR_WW(4, 2, 4);
        default: break;
    }
#undef DIPR_WW
    // ImageNet-geometry maps: WxCfg<W, dy row width in LDS, tiles per (segment) row, tile rows, images, segments, floats per item>
#define DIPR_WX(...)                                                                                                  \
    do {                                                                                                              \
        if (a) hipExtLaunchKernelGGL((k_conv_wino_wgrad_x<WxCfg<__VA_ARGS__>>), dim3(grid), dim3(256), 0, st, a, b, 0, x, dy, part, N, \
                                     Ci, Co, H, tiles_co, tiles_ci, chunks, chunks_per_split, dipr_xcd_remap());      \
        else hipLaunchKernelGGL((k_conv_wino_wgrad_x<WxCfg<__VA_ARGS__>>), dim3(grid), dim3(256), 0, st, x, dy, part, N, Ci, Co, H,   \
                                tiles_co, tiles_ci, chunks, chunks_per_split, dipr_xcd_remap());                      \
        return true;                                                                                                  \
    } while (0)
    switch (width) {
        case 56: DIPR_WX(56, 28, 14, 1, 1, 2, 4);
        case 28: DIPR_WX(28, 28, 14, 1, 1, 1, 4);
        case 14: DIPR_WX(14, 16, 8, 1, 2, 1, 2);
        case 7: DIPR_WX(7, 8, 4, 4, 1, 1, 1);
        default: return false;
    }
#undef DIPR_WX
}

int dipr_xcd_remap() {
    static const int v = (getenv("DEEPIPR_XCD_REMAP") && !atoi(getenv("DEEPIPR_XCD_REMAP"))) ? 0 : 1;
    return v;
}

FwPlan dipr_plan_conv_wino(int N, int C, int M, int H, int W, int k, int stride, int pad) {
    return plan_conv_wino(N, C, M, H, W, k, stride, pad);
}

bool dipr_launch_conv_wino(const FwPlan &p, bool dgrad, const float *wgt, const float *in, float *out, int N, int Cin, int M, int H,
                           float *ws, hipStream_t st, hipEvent_t ev_a, hipEvent_t ev_b) {
    if (p.cfg < 1000) return false;
    return dgrad ? dispatch<true, false>(p, wgt, in, out, N, Cin, M, H, ws, st, ev_a, ev_b)
                 : dispatch<false, false>(p, wgt, in, out, N, Cin, M, H, ws, st, ev_a, ev_b);
}

bool dipr_launch_conv_wino_pre(const FwPlan &p, const float *image, const float *in, float *out, int N, int Cin, int M, int H,
                               float *ws, hipStream_t st, hipEvent_t ev_a, hipEvent_t ev_b) {
    if (p.cfg < 1000 || M % 32 || Cin % 8) return false;
    return dispatch<false, true>(p, image, in, out, N, Cin, M, H, ws, st, ev_a, ev_b);
}

size_t dipr_wino_image_floats(int Co, int Ci) {
    if (Co <= 0 || Ci <= 0 || Co % 32 || Ci % 32) return 0;
    return static_cast<size_t>(Ci / 8) * (Co / 32) * kWnImgBlock;          // = (Co / 8) * (Ci / 32) blocks in the other direction
}

int dipr_wino_max_layers() { return kWnXfMaxLayers; }

bool dipr_launch_wino_weights(const DiprWinoLayer *layers, int n, hipStream_t st, hipEvent_t ev_a, hipEvent_t ev_b) {
    if (n <= 0 || n > kWnXfMaxLayers) return false;
    WnXfBatch B{};
    B.n = n;
    int blocks = 0;
    for (int i = 0; i < n; ++i) {
        const DiprWinoLayer &l = layers[i];
        if (!dipr_wino_image_floats(l.Co, l.Ci) || !l.W) return false;
        B.L[i] = WnXfLayer{l.W, l.Uf, l.Ud, l.Co, l.Ci, blocks};
        blocks += (l.Co / 32) * (l.Ci / 32);
    }
    if (ev_a) hipExtLaunchKernelGGL(k_wino_weights, dim3(blocks), dim3(256), 0, st, ev_a, ev_b, 0, B);
    else hipLaunchKernelGGL(k_wino_weights, dim3(blocks), dim3(256), 0, st, B);
    return true;
}

#ifdef DEEPIPR_TRACE
bool dipr_wino_set_trace(unsigned long long *device_buffer) {
    return hipMemcpyToSymbol(HIP_SYMBOL(g_wino_trace), &device_buffer, sizeof(device_buffer)) == hipSuccess;
}
#endif
