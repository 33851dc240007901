// Probe (gfx950): do vector-memory loads (the L1/TA -> VGPR return path) and LDS reads overlap inside one CU, or do
// they add up?  The recurrence gather issues, per row group and wave, ~3.8 buffer_load_dwordx4 of operator records
// (L2-resident, every workgroup streams the same ~600 KB) and ~10 ds_read_b128 of plane entries; its measured time is the
// SUM of what either stream takes alone (EXPERIMENTS.md).  This probe times the two streams alone and together:
//   one workgroup of 512 threads per CU (two waves per SIMD, 160 KB of LDS, like the kernel), every wave loops over
//   "groups": NLOAD 16-byte loads per lane from a 640 KB table (L2-resident after the first pass) and NREAD ds_read_b128 at
//   per-lane pseudo-random 16-byte entries, results folded into a checksum so nothing is optimised away.
// build: hipcc --offload-arch=gfx950 -O3 -o ta_lds_probe ta_lds_probe.hip ; run: ./ta_lds_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int NLOAD, int NREAD>
__global__ void __launch_bounds__(512) probe(const float4* __restrict__ table, int table_records, float* out, long long* cycles,
                                             int groups) {
    __shared__ float4 T[10240];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 10240; i += 512) T[i] = make_float4(i, 1.f, 2.f, 3.f);
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)table, 0, (unsigned)table_records * 1024u, 0x00020000);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    unsigned seed = tid * 2654435761u + 12345u;
    const long long t0 = __builtin_readcyclecounter();
    unsigned rec = wave * 4;
    for (int g = 0; g < groups; ++g) {
        f32x4 v[NLOAD > 0 ? NLOAD : 1];
#pragma unroll
        for (int i = 0; i < NLOAD; ++i) {
            v[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, (rec + i) * 1024u, 0);
        }
        rec += 8 * 4;
        if (rec + 4 >= (unsigned)table_records) rec = wave * 4;
        float4 t[NREAD > 0 ? NREAD : 1];
#pragma unroll
        for (int i = 0; i < NREAD; ++i) {
            seed = seed * 1664525u + 1013904223u;
            t[i] = T[(seed >> 8) % 10240u];
        }
#pragma unroll
        for (int i = 0; i < NREAD; ++i) {
            acc.x += t[i].x; acc.y += t[i].y; acc.z += t[i].z; acc.w += t[i].w;
        }
#pragma unroll
        for (int i = 0; i < NLOAD; ++i) {
            acc.x += v[i].x; acc.y += v[i].y; acc.z += v[i].z; acc.w += v[i].w;
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 512 + tid] = acc.x + acc.y + acc.z + acc.w;
    if (tid == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int NLOAD, int NREAD>
static void run(const float4* table, int records, float* out, long long* cyc, int groups, int cus) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(a);
        hipLaunchKernelGGL((probe<NLOAD, NREAD>), dim3(cus), dim3(512), 0, 0, table, records, out, cyc, groups);
        hipEventRecord(b);
        hipEventSynchronize(b);
    }
    float ms = 0.f;
    hipEventElapsedTime(&ms, a, b);
    std::vector<long long> h(cus);
    hipMemcpy(h.data(), cyc, cus * sizeof(long long), hipMemcpyDeviceToHost);
    double mean = 0;
    for (long long c : h) mean += (double)c / cus;
    const double per_group = mean / groups;          // cycles per (group of every wave) = per wave
    printf("loads/group %d  reads/group %d : %8.3f ms, %9.0f cycles per workgroup, %7.1f cycles per group and wave, "
           "per CU: %5.1f cycles per wave-load, %5.1f per wave-read\n",
           NLOAD, NREAD, ms, mean, per_group, NLOAD ? per_group / (8.0 * NLOAD) * 8 / 8 : 0.0, NREAD ? per_group / (8.0 * NREAD) * 8 / 8 : 0.0);
}

int main() {
    int dev = 0;
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, dev);
    const int cus = prop.multiProcessorCount, records = 640, groups = 2000;
    float4* table;
    float* out;
    long long* cyc;
    hipMalloc(&table, (size_t)records * 1024);
    hipMemset(table, 0, (size_t)records * 1024);
    hipMalloc(&out, (size_t)cus * 512 * sizeof(float));
    hipMalloc(&cyc, cus * sizeof(long long));
    printf("%s, %d CUs; per group and wave; 8 waves per CU share the paths: cycles per wave-instruction at CU level = per-group cycles / (8 waves * n) * 8\n",
           prop.name, cus);
    run<4, 0>(table, records, out, cyc, groups, cus);
    run<0, 10>(table, records, out, cyc, groups, cus);
    run<4, 10>(table, records, out, cyc, groups, cus);
    run<2, 0>(table, records, out, cyc, groups, cus);
    run<0, 5>(table, records, out, cyc, groups, cus);
    run<2, 5>(table, records, out, cyc, groups, cus);
    run<8, 0>(table, records, out, cyc, groups, cus);
    run<0, 20>(table, records, out, cyc, groups, cus);
    run<8, 20>(table, records, out, cyc, groups, cus);
    run<4, 20>(table, records, out, cyc, groups, cus);
    run<8, 10>(table, records, out, cyc, groups, cus);
    return 0;
}
