// Probe (gfx950): is the SGPR offset of a raw buffer load part of the range check, and what does an
// out-of-range 16-byte buffer load cost compared with an L2-resident one?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void probe(const float* p, float* out, unsigned nrec, unsigned soff, unsigned vsel) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, nrec, 0x00020000);
    const unsigned lane = threadIdx.x;
    f32x4 a = __builtin_amdgcn_raw_buffer_load_b128(r, lane * 16 + vsel, soff, 0);
    out[lane * 4 + 0] = a.x; out[lane * 4 + 1] = a.y; out[lane * 4 + 2] = a.z; out[lane * 4 + 3] = a.w;
}

// timing: each wave issues `n` loads of 1 KB from a 512 KB L2-resident table (mode 0), or the same number
// of out-of-range loads via voffset (mode 1) / via soffset (mode 2)
__global__ void __launch_bounds__(256) timing(const float* p, float* out, unsigned nrec, int n, int mode) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, nrec, 0x00020000);
    const unsigned lane = threadIdx.x & 63;
    const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    f32x4 acc = {0, 0, 0, 0};
    unsigned so = (blockIdx.x * 4 + wave) * 1024u;
    for (int i = 0; i < n; i += 4) {
        f32x4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned s = (so + (i + k) * 4096u) & (512u * 1024u - 1u);
            if (mode == 0) v[k] = __builtin_amdgcn_raw_buffer_load_b128(r, lane * 16, s, 0);
            else if (mode == 1) v[k] = __builtin_amdgcn_raw_buffer_load_b128(r, lane * 16 + 0x7F000000u, s, 0);
            else v[k] = __builtin_amdgcn_raw_buffer_load_b128(r, lane * 16, s + 0x7F000000u, 0);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) acc += v[k];
    }
    if (acc.x == 12345.f) out[0] = acc.y + acc.z + acc.w;
}

int main() {
    const size_t N = 1 << 20;
    std::vector<float> h(N);
    for (size_t i = 0; i < N; ++i) h[i] = (float)(i % 1000) + 1.f;
    float *d, *o;
    hipMalloc(&d, N * 4); hipMalloc(&o, 4096);
    hipMemcpy(d, h.data(), N * 4, hipMemcpyHostToDevice);
    float ho[256];
    struct { unsigned nrec, soff, vsel; const char* what; } cases[] = {
        {4096, 0, 0, "in range"}, {4096, 8192, 0, "soffset beyond num_records (memory is allocated)"},
        {4096, 0, 8192, "voffset beyond num_records"}, {4096, 4096 - 512, 0, "soffset+voffset straddles the end (lanes >= 32 out)"}};
    for (auto& c : cases) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, o, c.nrec, c.soff, c.vsel);
        hipMemcpy(ho, o, 1024, hipMemcpyDeviceToHost);
        printf("%-60s lane0: %g %g  lane40: %g %g   (memory there: %g, %g)\n", c.what, ho[0], ho[1], ho[160], ho[161],
               h[(c.soff + c.vsel) / 4], h[(c.soff + c.vsel + 640) / 4]);
    }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 3; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(timing, dim3(256 * 3), dim3(256), 0, 0, d, o, 512u * 1024u, 4096, mode);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double bytes = 256.0 * 3 * 4 * 4096 * 1024;
            printf("mode %d (%s): %.3f ms  -> %.1f B/clk/CU at 2.4 GHz if all were delivered\n", mode,
                   mode == 0 ? "L2-resident loads" : mode == 1 ? "out of range via voffset" : "out of range via soffset", ms,
                   bytes / (ms * 1e-3) / 256 / 2.4e9);
        }
    }
    return 0;
}
