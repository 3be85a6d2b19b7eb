// GPU debug harness: runs Sim<ModelAnt>::step on AoS states from a file, writes AoS results (+ debug taps).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../isaacgymenvs_amd/csrc/core/engine.hpp"
#include "../../isaacgymenvs_amd/csrc/gen/model_ant.h"
using namespace mi;
using M = ModelAnt;
constexpr int ND = M::ND, NSPH = M::NSPH, NSENS = M::NSENS;
constexpr int SS = 13 + 2 * ND + 3 * NSPH + ND;
#ifndef NDBG
#define NDBG 64
#endif
__global__ __launch_bounds__(64) void k(int n, SimParams P, float* state, const float* tau, float* dbg) {
    int e = blockIdx.x * 64 + threadIdx.x;
    if (e >= n) return;
    float* s = state + (size_t)e * SS;
    extern __shared__ float lds_rows[];
    Sim<M> sim;
    for (int i = 0; i < 13; ++i) sim.root[i] = s[i];
    for (int i = 0; i < ND; ++i) { sim.q[i] = s[13 + i]; sim.qd[i] = s[13 + ND + i]; }
    float t[ND];
    for (int i = 0; i < ND; ++i) t[i] = tau[(size_t)e * ND + i];
    float outb[6 * NSENS + ND];
    const float h = P.dt / (float)P.substeps;
    for (int ss = 0; ss < P.substeps; ++ss)
        sim.substep(P, t, h, RowStore<64>(lds_rows + threadIdx.x), Strided{s + 13 + 2 * ND, 1}, Strided{s + 13 + 2 * ND + 3 * NSPH, 1},
                    Strided{outb, 1}, Strided{outb + 6 * NSENS, 1}, PlaneGround{}, -1.f, Strided{nullptr, 1});
    for (int i = 0; i < 13; ++i) s[i] = sim.root[i];
    for (int i = 0; i < ND; ++i) { s[13 + i] = sim.q[i]; s[13 + ND + i] = sim.qd[i]; }
}
int main(int argc, char** argv) {
    const size_t LDSB = (size_t)Sim<M>::ROW_SLOTS * 64 * 4;
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDSB);
    const char* in = argv[1]; const char* out = argv[2];
    FILE* f = fopen(in, "rb");
    int n; SimParams P;
    fread(&n, 4, 1, f); fread(&P, sizeof(P), 1, f);
    std::vector<float> st((size_t)n * SS), tau((size_t)n * ND), dbg((size_t)n * NDBG, 0.f);
    fread(st.data(), 4, st.size(), f); fread(tau.data(), 4, tau.size(), f); fclose(f);
    float *ds, *dt, *dd;
    hipMalloc(&ds, st.size() * 4); hipMalloc(&dt, tau.size() * 4); hipMalloc(&dd, dbg.size() * 4);
    hipMemcpy(ds, st.data(), st.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dt, tau.data(), tau.size() * 4, hipMemcpyHostToDevice);
    hipMemset(dd, 0, dbg.size() * 4);
    // determinism probe: same inputs, two launches into separate buffers
    {
        float* ds2; hipMalloc(&ds2, st.size() * 4);
        std::vector<float> r1(st.size()), r2(st.size());
        for (int rep = 0; rep < 2; ++rep) {
            hipMemcpy(ds2, st.data(), st.size() * 4, hipMemcpyHostToDevice);
            hipLaunchKernelGGL(k, dim3((n + 63) / 64), dim3(64), LDSB, 0, n, P, ds2, dt, dd);
            hipDeviceSynchronize();
            hipMemcpy(rep ? r2.data() : r1.data(), ds2, st.size() * 4, hipMemcpyDeviceToHost);
        }
        int diff = 0;
        for (size_t i = 0; i < r1.size(); ++i) if (memcmp(&r1[i], &r2[i], 4)) diff++;
        printf("determinism: %d of %zu words differ between two identical launches\n", diff, r1.size());
        // replicate env 20 into every lane: identical inputs per lane must give identical outputs
        std::vector<float> rs(st.size()), rt(tau.size());
        for (int e = 0; e < n; ++e) { memcpy(&rs[(size_t)e * SS], &st[(size_t)20 * SS], SS * 4); memcpy(&rt[(size_t)e * ND], &tau[(size_t)20 * ND], ND * 4); }
        float* dt2; hipMalloc(&dt2, tau.size() * 4);
        hipMemcpy(ds2, rs.data(), rs.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dt2, rt.data(), rt.size() * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3((n + 63) / 64), dim3(64), LDSB, 0, n, P, ds2, dt2, dd);
        hipDeviceSynchronize();
        hipMemcpy(r1.data(), ds2, st.size() * 4, hipMemcpyDeviceToHost);
        int lanes_diff = 0;
        for (int e = 1; e < n; ++e) if (memcmp(&r1[(size_t)e * SS], &r1[0], SS * 4)) lanes_diff++;
        printf("replicated env 20: %d of %d lanes differ from lane 0; lane0 qd0=%g\n", lanes_diff, n, r1[13 + ND]);
    }
    hipLaunchKernelGGL(k, dim3((n + 63) / 64), dim3(64), LDSB, 0, n, P, ds, dt, dd);
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) { printf("hip error %s\n", hipGetErrorString(e)); return 1; }
    hipMemcpy(st.data(), ds, st.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(dbg.data(), dd, dbg.size() * 4, hipMemcpyDeviceToHost);
    f = fopen(out, "wb"); fwrite(st.data(), 4, st.size(), f); fwrite(dbg.data(), 4, dbg.size(), f); fclose(f);
    printf("ok %s n=%d\n", out, n);
    return 0;
}
