// calib_fetch.hip -- known-byte-count kernels in the engine's own access pattern, to calibrate rocprofv3's FETCH_SIZE / WRITE_SIZE
// on gfx950 (the guide's x2 rule was measured on 16 B/lane streams and says: calibrate other widths yourself).
//   soa_copy<K>:  every thread reads K dwords from an SoA array [K][N] (one dword per lane and field, fully coalesced 256-B wave
//                 transactions -- what load_sim / store_sim do) and writes K dwords to a second array: 4 K N bytes each way.
//   soa_read<K>:  reads only (one 4-byte result per wave to keep the loads alive).
// Build: hipcc --offload-arch=gfx950 -O3 tools/calib/calib_fetch.hip -o tools/calib/calib_fetch ; run under
//   rocprofv3 --pmc FETCH_SIZE -- tools/calib/calib_fetch   and   rocprofv3 --pmc WRITE_SIZE -- tools/calib/calib_fetch
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int K>
__global__ void soa_copy(const float* __restrict__ in, float* __restrict__ out, int N) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= N) return;
    float v[K];
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] = in[(size_t)k * N + e];
#pragma unroll
    for (int k = 0; k < K; ++k) out[(size_t)k * N + e] = v[k] * 1.0001f;
}
template <int K>
__global__ void soa_read(const float* __restrict__ in, float* __restrict__ out, int N) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= N) return;
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) acc += in[(size_t)k * N + e];
    if (acc == 12345.678f) out[e] = acc;     // never true: the loads stay, nothing is written
}

int main() {
    constexpr int K = 32;
    // three sizes: the engine's own (Ant@4096 state is ~0.5 MB: L2-resident between launches), 64 MB and 512 MB (past the 256 MB L3)
    const int sizes[3] = {4096, 512 * 1024, 4 * 1024 * 1024};
    for (int N : sizes) {
        float *in, *out;
        const size_t bytes = (size_t)K * N * sizeof(float);
        hipMalloc(&in, bytes); hipMalloc(&out, bytes);
        hipMemset(in, 0, bytes); hipMemset(out, 0, bytes);
        for (int rep = 0; rep < 5; ++rep) {
            hipLaunchKernelGGL(soa_copy<K>, dim3((N + 63) / 64), dim3(64), 0, 0, in, out, N);
            hipLaunchKernelGGL(soa_read<K>, dim3((N + 63) / 64), dim3(64), 0, 0, in, out, N);
        }
        hipDeviceSynchronize();
        printf("N=%d K=%d: soa_copy reads %zu B and writes %zu B per launch; soa_read reads %zu B\n", N, K, bytes, bytes, bytes);
        hipFree(in); hipFree(out);
    }
    return 0;
}
