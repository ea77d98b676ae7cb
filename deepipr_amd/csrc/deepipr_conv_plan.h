// deepipr_conv_plan.h -- internal (not installed): what the two translation units of libdeepipr_hip.so share about the
// forward / backward-data convolution kernels.  deepipr_hip.hip holds the C ABI and every other kernel; deepipr_wino.hip
// holds the Winograd kernels and is compiled with -fno-slp-vectorize (the SLP vectoriser packs the transforms' scalar fp32
// adds into v_pk_* instructions that need register shuffles to feed: +30 vector instructions per chunk beside the MFMAs).
#ifndef DEEPIPR_CONV_PLAN_H
#define DEEPIPR_CONV_PLAN_H
#include <hip/hip_runtime.h>
#include <cstddef>

#define DIPR_HIDDEN __attribute__((visibility("hidden")))

// Workgroup b of a launch runs on XCD b % 8 (observed dispatch order; used for speed only), each XCD with an L2 of its own.  Tiles
// that share an operand are neighbours in the tile index: taken as they come they land on eight DIFFERENT XCDs and each fetches
// the shared operand through the fabric.  The remap makes the workgroups of ONE XCD walk a contiguous eighth of the index space:
// neighbours meet in one L2.  (n not a multiple of 8: the last n % 8 indices stay put.)
__device__ __forceinline__ int dipr_xcd_contiguous(int b, int n) {
    const int n8 = n & ~7;
    return b < n8 ? (b & 7) * (n8 >> 3) + (b >> 3) : b;
}
// DEEPIPR_XCD_REMAP=0 switches the remap of the Winograd kernels off (A/B)
DIPR_HIDDEN int dipr_xcd_remap();

struct FwPlan {
    int cfg;        // 0: unsupported; >= 1000: a Winograd instance (1000 + width code * 100 + m blocks * 10 + k groups)
    int bands;      // row bands per image group
    int grid;       // output tiles (workgroups of the plain form)
    int splits;     // K splits (1: plain)
    int cps;        // chunks of CK channels per split
    size_t slab;    // floats of one partial output
};

DIPR_HIDDEN int dipr_device_cu_count();
// H, W: the map (stride 1, pad 1: input and output alike); C: input channels of the GEMM, M: output channels
DIPR_HIDDEN FwPlan dipr_plan_conv_wino(int N, int C, int M, int H, int W, int k, int stride, int pad);
// Enqueues k_conv_wino for plan `p` (ws != nullptr: split K, slab `split` of ws takes the partial output).  ev_a / ev_b: the
// dispatch's own start / stop events when the caller times it (both or neither).  -> false: no instance for p.cfg.
DIPR_HIDDEN bool dipr_launch_conv_wino(const FwPlan &p, bool dgrad, const float *wgt, const float *in, float *out, int N, int Cin,
                                       int M, int H, float *ws, hipStream_t st, hipEvent_t ev_a, hipEvent_t ev_b);
// The pre-transformed form (deepipr_conv_wino.inc): `image` = U = G g G^T of this direction's weights as k_wino_weights lays it out
// (dipr_wino_image_floats(Co, Ci) floats per direction; 0: no such form, Co and Ci must be multiples of 32).  M, Cin: the GEMM's
// output / input channels (backward-data: M = Ci, Cin = Co, image = Ud).
DIPR_HIDDEN bool dipr_launch_conv_wino_pre(const FwPlan &p, const float *image, const float *in, float *out, int N, int Cin, int M,
                                           int H, float *ws, hipStream_t st, hipEvent_t ev_a, hipEvent_t ev_b);
DIPR_HIDDEN size_t dipr_wino_image_floats(int Co, int Ci);
DIPR_HIDDEN int dipr_wino_max_layers();
struct DiprWinoLayer {
    const float *W;      // [Co][Ci][3][3]
    float *Uf, *Ud;      // forward / backward-data image (either may be null)
    int Co, Ci;
};
DIPR_HIDDEN bool dipr_launch_wino_weights(const DiprWinoLayer *layers, int n, hipStream_t st, hipEvent_t ev_a, hipEvent_t ev_b);
// Winograd F(3x3, 2x2) weight gradient (deepipr_conv_wino_wgrad.inc) on a `width`-wide map (4 / 8 / 16 / 32): 64 co x 32 ci
// partial tiles in k_conv3x3_wgrad's CIT = 32 layout, chunks of 16 tiles (images per chunk: 4 / 1 / 1 / 1, tile rows 2 / 4 / 2 / 1).
DIPR_HIDDEN bool dipr_launch_wgrad_wino(int width, const float *x, const float *dy, float *part, int N, int Ci, int Co, int H,
                                        int tiles_co, int tiles_ci, int chunks, int chunks_per_split, int grid, hipStream_t st,
                                        hipEvent_t ev_a, hipEvent_t ev_b);
#ifdef DEEPIPR_TRACE
DIPR_HIDDEN bool dipr_wino_set_trace(unsigned long long *device_buffer);
#endif
#endif
