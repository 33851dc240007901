// Probe (gfx950): sustained rate of v_mfma_f32_32x32x2_f32 (the contraction's matrix instruction) with nothing else going
// on -- the clock the matrix pipe really runs at under load (nominal: 256 flop/clk/CU x 2.4 GHz x 256 CUs = 157 TFLOP/s).
// build: hipcc --offload-arch=gfx950 -O3 -w -o mfma_f32_probe mfma_f32_probe.hip ; run: ./mfma_f32_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int WAVES>
__global__ void __launch_bounds__(64 * WAVES) probe(float* out, int iters, long long* cyc) {
    f32x16 acc[4];
    for (int r = 0; r < 4; ++r)
        for (int j = 0; j < 16; ++j) acc[r][j] = 0.f;
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    const long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[3], 0, 0, 0);
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int r = 0; r < 4; ++r)
        for (int j = 0; j < 16; ++j) s += acc[r][j];
    out[blockIdx.x * 64 * WAVES + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int WAVES>
static void run(float* out, long long* cyc, int cus, int wgs_per_cu, int iters) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    float ms = 0.f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(a);
        hipLaunchKernelGGL((probe<WAVES>), dim3(cus * wgs_per_cu), dim3(64 * WAVES), 0, 0, out, iters, cyc);
        hipEventRecord(b);
        hipEventSynchronize(b);
        hipEventElapsedTime(&ms, a, b);
    }
    long long c0 = 0;
    hipMemcpy(&c0, cyc, sizeof(c0), hipMemcpyDeviceToHost);
    const double flop = (double)cus * wgs_per_cu * WAVES * iters * 32.0 * (2.0 * 32 * 32 * 2);
    printf("%2d waves per workgroup x %d workgroups per CU, %6d iterations: %8.3f ms  %7.1f TFLOP/s;  cycle counter of workgroup 0: %lld ticks = %.0f MHz\n",
           WAVES, wgs_per_cu, iters, ms, flop / ms / 1e9, c0, c0 / ms / 1e3);
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    float* out;
    long long* cyc;
    hipMalloc(&out, (size_t)cus * 16 * 64 * 16 * sizeof(float));
    hipMalloc(&cyc, (size_t)cus * 16 * sizeof(long long));
    printf("%s, %d CUs, clockRate %d kHz\n", prop.name, cus, prop.clockRate);
    run<4>(out, cyc, cus, 1, 2000);     // one wave per SIMD
    run<4>(out, cyc, cus, 1, 20000);
    run<8>(out, cyc, cus, 1, 10000);    // two waves per SIMD
    run<4>(out, cyc, cus, 3, 10000);    // three waves per SIMD (the contraction's occupancy)
    run<4>(out, cyc, cus, 3, 1000);     // ~0.1 ms: the length of one contraction launch
    run<4>(out, cyc, cus, 3, 300);
    return 0;
}
