// Probe (gfx950): what does the memory system deliver to the access patterns of the contraction kernels, with no
// arithmetic at all?  Calibrates the "64 % of 8 TB/s with the MFMAs compiled out" of EXPERIMENTS.md section 3.
//   read<U, NT>  : every wave streams 1 KB per load instruction (16 B per lane), U loads in flight, grid-stride over a
//                  buffer of `bytes`; nt = non-temporal loads (what ld_stream() uses)
//   rows<U>      : the contraction's pattern: a wave owns 128 consecutive vertices and walks 160 planes that lie
//                  `plane` bytes apart (half-wave = 512 contiguous bytes of one plane, two planes per instruction)
//   copy<U>      : read + write of the same amount
// build: hipcc --offload-arch=gfx950 -O3 -w -o hbm_stream_probe hbm_stream_probe.hip ; run: ./hbm_stream_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int U, bool NT>
__global__ void __launch_bounds__(256) read_kernel(const f32x4* __restrict__ p, size_t n16, float* out) {
    const size_t stride = (size_t)gridDim.x * 256 * U;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (size_t i = (size_t)blockIdx.x * 256 * U + threadIdx.x; i + (U - 1) * 256 < n16; i += stride) {
        f32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(p + i + u * 256) : p[i + u * 256];
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u];
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;
}

// one wave = 128 vertices x `rows` planes; planes of one window `plane` floats apart, windows rows*plane apart
template <int U>
__global__ void __launch_bounds__(256) rows_kernel(const float* __restrict__ p, int Mp, int rows, int tiles, float* out) {
    extern __shared__ float occupancy_limiter[];          // dynamic LDS only to hold the launch to 160 KB / size workgroups per CU
    if (Mp < 0) occupancy_limiter[threadIdx.x] = 0.f;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 31, h = lane >> 5;
    const int tile = blockIdx.x * 4 + wave;
    if (tile >= tiles) return;
    const int b = blockIdx.y;
    const float* base = p + (size_t)b * rows * Mp + tile * 128 + 4 * c;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int r0 = 0; r0 < rows; r0 += 2 * U) {
        f32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u)
            v[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(base + (size_t)(r0 + 2 * u + h) * Mp));
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u];
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;
}

// the same with the stack's real layout [K][B][Fin][Mp]: row r = fin * K + k lies at k * slab + fin * Mp (slab = B * Fin * Mp
// floats, + pad): consecutive rows of a wave are a whole slab apart -- 2^21 * 41 bytes at the bench shape
template <int U>
__global__ void __launch_bounds__(256) rowsk_kernel(const float* __restrict__ p, int Mp, int Fin, int K, size_t slab, int tiles, float* out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 31, h = lane >> 5;
    const int tile = blockIdx.x * 4 + wave;
    if (tile >= tiles) return;
    const int b = blockIdx.y;
    const float* base = p + (size_t)b * Fin * Mp + tile * 128 + 4 * c;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int rows = Fin * K;
    for (int r0 = 0; r0 < rows; r0 += 2 * U) {
        f32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int r = r0 + 2 * u + h;
            v[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(base + (size_t)(r % K) * slab + (size_t)(r / K) * Mp));
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u];
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;
}

template <int U>
__global__ void __launch_bounds__(256) copy_kernel(const f32x4* __restrict__ p, f32x4* __restrict__ q, size_t n16) {
    const size_t stride = (size_t)gridDim.x * 256 * U;
    for (size_t i = (size_t)blockIdx.x * 256 * U + threadIdx.x; i + (U - 1) * 256 < n16; i += stride) {
        f32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(p + i + u * 256);
#pragma unroll
        for (int u = 0; u < U; ++u) __builtin_nontemporal_store(v[u], q + i + u * 256);
    }
}

// the pattern of contract_bwd_x's stores: a wave owns 128 vertices and writes `rows` planes (two per instruction)
template <int U, bool NT>
__global__ void __launch_bounds__(256) wrows_kernel(float* __restrict__ p, int Mp, int rows, int tiles) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 31, h = lane >> 5;
    const int tile = blockIdx.x * 4 + wave;
    if (tile >= tiles) return;
    const int b = blockIdx.y;
    float* base = p + (size_t)b * rows * Mp + tile * 128 + 4 * c;
    const f32x4 v = {1.f, 2.f, 3.f, (float)lane};
    for (int r0 = 0; r0 < rows; r0 += 2 * U) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            f32x4* q = reinterpret_cast<f32x4*>(base + (size_t)(r0 + 2 * u + h) * Mp);
            if (NT) __builtin_nontemporal_store(v, q); else *q = v;
        }
    }
}

template <bool NT>
__global__ void __launch_bounds__(256) write_kernel(f32x4* __restrict__ q, size_t n16) {
    const size_t stride = (size_t)gridDim.x * 256 * 4;
    const f32x4 v = {1.f, 2.f, 3.f, 4.f};
    for (size_t i = (size_t)blockIdx.x * 256 * 4 + threadIdx.x; i + 3 * 256 < n16; i += stride) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (NT) __builtin_nontemporal_store(v, q + i + u * 256); else q[i + u * 256] = v;
        }
    }
}

template <typename F>
static float time_ms(F launch, int reps) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    launch();
    launch();
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < reps; ++i) launch();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0.f;
    hipEventElapsedTime(&ms, a, b);
    return ms / reps;
}

int main(int argc, char** argv) {
    const int Mp = 10496, rows = 160, tiles = 82;
    const int B = argc > 1 ? atoi(argv[1]) : 64;     // windows: 64 = the stack of the bench shape (430 MB); 16 / 32 fit the 256 MB Infinity Cache
    const size_t bytes = (size_t)B * rows * Mp * 4;           // 430 MB: the stack of the bench shape
    float *p, *q, *out;
    hipMalloc(&p, bytes);
    hipMalloc(&q, bytes);
    hipMalloc(&out, 64);
    hipMemset(p, 0, bytes);
    hipMemset(q, 0, bytes);
    const size_t n16 = bytes / 16;
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    printf("%s, %d CUs, buffer %.1f MB\n", prop.name, cus, bytes / 1e6);
#define READ(U, NT, WGS)                                                                                              \
    {                                                                                                                 \
        const float ms = time_ms([&] { hipLaunchKernelGGL((read_kernel<U, NT>), dim3(WGS), dim3(256), 0, 0, (const f32x4*)p, n16, out); }, 20); \
        printf("read  U=%d nt=%d workgroups=%5d : %7.3f ms  %6.0f GB/s\n", U, (int)NT, WGS, ms, bytes / ms / 1e6);   \
    }
    READ(1, false, cus * 8) READ(2, false, cus * 8) READ(4, false, cus * 8) READ(8, false, cus * 8)
    READ(4, true, cus * 8) READ(8, true, cus * 8)
    READ(4, true, cus * 4) READ(8, true, cus * 4) READ(8, true, cus * 3) READ(8, true, cus * 2) READ(16, true, cus * 2)
    READ(8, true, (int)(n16 / (256 * 8)))                      // one pass per workgroup, no grid-stride loop
#define ROWS(U, WGPC)                                                                                                 \
    {                                                                                                                 \
        const int lds = WGPC ? (160 * 1024 / WGPC) & ~255 : 0;                                                        \
        hipFuncSetAttribute((const void*)rows_kernel<U>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);      \
        const float ms = time_ms([&] { hipLaunchKernelGGL((rows_kernel<U>), dim3((tiles + 3) / 4, B), dim3(256), lds, 0, p, Mp, rows, tiles, out); }, 20); \
        printf("rows  U=%d (128 vertices x 160 planes per wave, %d waves, %d workgroups per CU) : %7.3f ms  %6.0f GB/s\n", U, tiles * B, WGPC, ms,   \
               (double)tiles * 128 * rows * B * 4 / ms / 1e6);                                                       \
    }
    ROWS(2, 0) ROWS(4, 0) ROWS(8, 0) ROWS(16, 0)
    ROWS(8, 6) ROWS(8, 4) ROWS(8, 3) ROWS(8, 2) ROWS(16, 3) ROWS(16, 2) ROWS(4, 3) ROWS(2, 3)
#define ROWSK(U, PAD)                                                                                                 \
    {                                                                                                                 \
        const size_t slab = (size_t)B * 32 * Mp + PAD;                                                                \
        if (5 * slab * 4 <= bytes + 5 * 4 * (size_t)PAD && B * 160 == B * rows) {                                      \
            const float ms = time_ms([&] { hipLaunchKernelGGL((rowsk_kernel<U>), dim3((tiles + 3) / 4, B), dim3(256), 0, 0, p2, Mp, 32, 5, slab, tiles, out); }, 20); \
            printf("rowsk U=%d slab pad %6d floats (stack layout [K][B][Fin][Mp], slab stride %zu B) : %7.3f ms  %6.0f GB/s\n", U, PAD, slab * 4, ms, \
                   (double)tiles * 128 * rows * B * 4 / ms / 1e6);                                                   \
        }                                                                                                             \
    }
    float* p2;
    hipMalloc(&p2, bytes + 5 * 4 * 65536);
    hipMemset(p2, 0, bytes + 5 * 4 * 65536);
    ROWSK(8, 0) ROWSK(8, 64) ROWSK(8, 1024) ROWSK(8, 2080) ROWSK(8, 16416) ROWSK(4, 0) ROWSK(4, 2080)
#define COPY(U, WGS)                                                                                                  \
    {                                                                                                                 \
        const float ms = time_ms([&] { hipLaunchKernelGGL((copy_kernel<U>), dim3(WGS), dim3(256), 0, 0, (const f32x4*)p, (f32x4*)q, n16); }, 20); \
        printf("copy  U=%d workgroups=%5d : %7.3f ms  %6.0f GB/s (read + write)\n", U, WGS, ms, 2.0 * bytes / ms / 1e6); \
    }
    COPY(4, cus * 8) COPY(8, cus * 4) COPY(8, cus * 8)
#define WRITE(NT, WGS)                                                                                                \
    {                                                                                                                 \
        const float ms = time_ms([&] { hipLaunchKernelGGL((write_kernel<NT>), dim3(WGS), dim3(256), 0, 0, (f32x4*)q, n16); }, 20); \
        printf("write nt=%d workgroups=%5d : %7.3f ms  %6.0f GB/s\n", (int)NT, WGS, ms, bytes / ms / 1e6);                \
    }
    WRITE(false, cus * 8) WRITE(true, cus * 8) WRITE(true, cus * 4) WRITE(true, cus * 16)
#define WROWS(U, NT)                                                                                                  \
    {                                                                                                                 \
        const float ms = time_ms([&] { hipLaunchKernelGGL((wrows_kernel<U, NT>), dim3((tiles + 3) / 4, B), dim3(256), 0, 0, q, Mp, rows, tiles); }, 20); \
        printf("wrows U=%d nt=%d (128 vertices x 160 planes per wave) : %7.3f ms  %6.0f GB/s\n", U, (int)NT, ms,          \
               (double)tiles * 128 * rows * B * 4 / ms / 1e6);                                                       \
    }
    WROWS(8, false) WROWS(8, true) WROWS(16, true) WROWS(2, true)
    return 0;
}
