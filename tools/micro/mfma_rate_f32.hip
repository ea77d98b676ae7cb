// Issue rate of v_mfma_f32_32x32x2_f32 (the instruction every convolution kernel of this library runs on): W waves per SIMD,
// NACC independent accumulators per wave, short (one "launch" of ~50 us, a convolution's length) and long runs -- what the chip
// clocks under fp32 matrix load.  Peak at the nominal 2.4 GHz: 256 CUs x 4 SIMDs x 1024 FLOP / 16 passes of 4 clk = 157.3 TFLOP/s.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_rate_f32.out tools/micro/mfma_rate_f32.hip && /tmp/mfma_rate_f32.out
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(512) void k(float *out, int iters, long long *cycles) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    const float a = 1.0f + threadIdx.x * 1e-3f, b = 1.0f - threadIdx.x * 1e-3f;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < NACC; ++i)
        for (int j = 0; j < 16; ++j) s += acc[i][j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = t1 - t0;
}

template <int NACC>
void run(int threads, int iters, int reps, const char *what) {
    float *out; long long *cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<NACC><<<256, threads>>>(out, iters, cyc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) k<NACC><<<256, threads>>>(out, iters, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    const double n = (double)iters * NACC;                       // MFMAs per wave
    const int wps = threads / 256;
    const double tf = 2.0 * 32 * 32 * 2 * n * wps * 4 * 256 / (ms * 1e-3) / 1e12;
    printf("%-40s waves/SIMD %d  NACC %d  iters %7d x %3d launches  %9.1f us/launch  %6.2f ns/MFMA/SIMD  %6.1f TFLOP/s = %.3f of 157.3\n",
           what, wps, NACC, iters, reps, ms * 1e3, ms * 1e6 / (n * wps), tf, tf / 157.3);
    hipFree(out); hipFree(cyc);
}

int main() {
    run<8>(256, 200, 50, "short launches, back to back");
    run<8>(512, 100, 50, "short launches, back to back");
    run<4>(512, 200, 50, "short launches, back to back");
    run<8>(256, 2000, 20, "0.5 ms launches");
    run<8>(512, 1000, 20, "0.5 ms launches");
    run<8>(256, 40000, 4, "10 ms launches");
    run<8>(512, 20000, 4, "10 ms launches");
    run<16>(256, 20000, 4, "10 ms launches");
    return 0;
}
