// Issue rate of v_mfma_f32_32x32x16_bf16 as this repo's kernels use it: W waves per SIMD, NACC independent accumulators
// per wave, FILL v_perm fillers per MFMA; short (one "launch" of ~40 us) and long runs, to see clocks under load.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_rate.out mfma_rate.hip && ./mfma_rate.out
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int NACC, int FILL>
__global__ __launch_bounds__(512) void k(float *out, int iters, long long *cycles) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    u32x4 a = {threadIdx.x, 2u, 3u, 4u}, b = {5u, threadIdx.x, 7u, 8u};
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc[i], 0, 0, 0);
#pragma unroll
            for (int f = 0; f < FILL; ++f) b[f & 3] = __builtin_amdgcn_alignbit(b[(f + 1) & 3], a[f & 3], 16);
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < NACC; ++i)
        for (int j = 0; j < 16; ++j) s += acc[i][j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = t1 - t0;
}

template <int NACC, int FILL>
void run(int threads, int iters, const char *what) {
    float *out; long long *cyc, h;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<NACC, FILL><<<256, threads>>>(out, iters, cyc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<NACC, FILL><<<256, threads>>>(out, iters, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    const double n = (double)iters * NACC;                       // MFMAs per wave
    const int wps = threads / 256;
    printf("%-46s waves/SIMD %d  NACC %d FILL %d  iters %7d  %8.1f us  %6.1f ns/MFMA/SIMD  s_memtime ticks/MFMA/SIMD %.1f  TF %.0f\n", what, wps, NACC, FILL,
           iters, ms * 1e3, ms * 1e6 / (n * wps), (double)h / (n * wps), 2.0 * 32 * 32 * 16 * n * wps * 4 * 256 / (ms * 1e-3) / 1e12);
    hipFree(out); hipFree(cyc);
}

// KIND: which vector instruction fills the gaps, N per MFMA (9 accumulators, one wave per SIMD)
template <int KIND, int N>
__global__ __launch_bounds__(256) void kf(float *out, int iters, long long *cycles) {
    f32x16 acc[9];
    for (int i = 0; i < 9; ++i)
        for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    u32x4 a = {threadIdx.x, 2u, 3u, 4u}, b = {5u, threadIdx.x, 7u, 8u};
    unsigned v[8];
    float fv[8];
    for (int i = 0; i < 8; ++i) { v[i] = threadIdx.x * (i + 3); fv[i] = threadIdx.x * 0.5f + i; }
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc[i], 0, 0, 0);
#pragma unroll
            for (int f = 0; f < N; ++f) {
                const int k = (i * N + f) & 7;
                if (KIND == 0) v[k] = __builtin_amdgcn_alignbit(v[k], v[(k + 1) & 7], 16);
                if (KIND == 1) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(v[k]) : "v"(v[k]), "v"(v[(k + 1) & 7]));
                if (KIND == 2) asm volatile("v_sub_f32 %0, %1, %2" : "=v"(fv[k]) : "v"(fv[k]), "v"(fv[(k + 1) & 7]));
                if (KIND == 3) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(*(double *)&fv[k & 6]) : "v"(*(double *)&fv[k & 6]), "v"(*(double *)&fv[(k + 2) & 6]));
                if (KIND == 4) asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(v[k]) : "v"(v[(k + 1) & 7]));
                if (KIND == 5) asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(v[k]) : "v"(v[(k + 1) & 7]));
                if (KIND == 6) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(v[k]) : "v"(fv[k]), "v"(fv[(k + 1) & 7]));
                if (KIND == 7) asm volatile("v_mov_b32 %0, %1" : "=v"(v[k]) : "v"(v[(k + 1) & 7]));
            }
        }
        asm volatile("" : "+v"(a), "+v"(b));
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 9; ++i)
        for (int j = 0; j < 16; ++j) s += acc[i][j];
    for (int i = 0; i < 8; ++i) s += v[i] + fv[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = t1 - t0;
}

template <int KIND, int N>
void runf(const char *what) {
    float *out; long long *cyc, h;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 8);
    const int iters = 200;
    kf<KIND, N><<<256, 256>>>(out, iters, cyc);
    hipDeviceSynchronize();
    kf<KIND, N><<<256, 256>>>(out, iters, cyc);
    hipDeviceSynchronize();
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-18s x %d per MFMA: %.1f ticks per MFMA\n", what, N, (double)h / (iters * 9.0));
    hipFree(out); hipFree(cyc);
}

#define ALLN(K, name) runf<K, 2>(name); runf<K, 4>(name); runf<K, 6>(name);

int main() {
    ALLN(0, "v_perm/alignbit") ALLN(1, "v_cndmask") ALLN(2, "v_sub_f32") ALLN(3, "v_pk_add_f32") ALLN(4, "v_and")
    ALLN(5, "v_lshlrev") ALLN(6, "v_cvt_pk_bf16_f32") ALLN(7, "v_mov")
    run<3, 0>(256, 600, "1 wave/SIMD, 3 accumulators, short");
    run<3, 0>(256, 60000, "1 wave/SIMD, 3 accumulators, long");
    run<9, 0>(256, 200, "1 wave/SIMD, 9 accumulators, short");
    run<9, 1>(256, 200, "1 wave/SIMD, 9 acc, 1 filler, short");
    run<9, 2>(256, 200, "1 wave/SIMD, 9 acc, 2 fillers, short");
    run<9, 0>(512, 100, "2 waves/SIMD, 9 accumulators, short");
    run<9, 0>(512, 10000, "2 waves/SIMD, 9 accumulators, long");
    run<1, 0>(256, 1800, "1 wave/SIMD, 1 accumulator (dependent chain)");
    return 0;
}
