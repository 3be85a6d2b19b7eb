// GPU debug harness: per-phase s_memtime stamps of one sub-step (Ant or Humanoid), 64 waves like the bench.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -DMI_TIMING [-DHUM] phase_timing.hip -o phase_timing
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../../isaacgymenvs_amd/csrc/core/engine.hpp"
#ifdef HUM
#include "../../isaacgymenvs_amd/csrc/gen/model_humanoid.h"
using M = ModelHumanoid;
#else
#include "../../isaacgymenvs_amd/csrc/gen/model_ant.h"
using M = ModelAnt;
#endif
using namespace mi;
constexpr int ND = M::ND, NSPH = M::NSPH, NSENS = M::NSENS;
constexpr int LANES = Sim<M>::LANES;
constexpr bool LDS_ROWS = (size_t)Sim<M>::ROW_SLOTS * LANES * 4 <= 160 * 1024;
__global__ __launch_bounds__(64) void k(int N, SimParams P, float* root, float* dof, float* lamc, float* laml, float* sens, float* dff,
                                        unsigned long long* stamps) {
    extern __shared__ float lds_rows[];
    const int e = blockIdx.x * LANES + threadIdx.x;
    Sim<M> sim;
    for (int i = 0; i < 13; ++i) sim.root[i] = root[i * N + e];
    for (int i = 0; i < ND; ++i) { sim.q[i] = dof[i * N + e]; sim.qd[i] = dof[(ND + i) * N + e]; }
    float t[M::NDA];
    for (int i = 0; i < ND; ++i) t[i] = 0.3f * (float)((e + i) % 7 - 3);
#if defined(MI_TIMING)
    sim.tstamp = (threadIdx.x == 0) ? stamps + blockIdx.x * 16 : nullptr;
#endif
    const float h = P.dt / (float)P.substeps;
    const Strided a{lamc + e, N}, b{laml + e, N}, c{sens + e, N}, d{dff + e, N};
    constexpr bool PRE = LDS_ROWS && Sim<M>::STAGES_LAM;   // LDS-direct staging exactly like step_kernels.hpp
    if constexpr (PRE) {
        typedef const __attribute__((address_space(1))) void* gptr_t;
        typedef __attribute__((address_space(3))) void* lptr_t;
        sfor<ND>([&](auto D) { constexpr int dd = D; if constexpr (M::dof_limited[dd])
            __builtin_amdgcn_global_load_lds((gptr_t)(laml + (size_t)dd * N + e), (lptr_t)(lds_rows + Sim<M>::stage_slot_lim(dd) * LANES), 4, 0, 0); });
        sfor<3 * NSPH>([&](auto K) {
            __builtin_amdgcn_global_load_lds((gptr_t)(lamc + (size_t)K * N + e), (lptr_t)(lds_rows + Sim<M>::stage_slot_con(K) * LANES), 4, 0, 0); });
    }
    if constexpr (LDS_ROWS && LANES == 64) sim.substep(P, t, h, RowStore<64>(lds_rows + threadIdx.x), a, b, c, d, PlaneGround{}, -1.f, Strided{nullptr, 1}, nullptr, PRE);
    else if constexpr (LDS_ROWS) sim.substep(P, t, h, RowStore<LANES>{lds_rows + threadIdx.x}, a, b, c, d, PlaneGround{}, -1.f, Strided{nullptr, 1}, nullptr, PRE);
    else { float rows[Sim<M>::ROW_SLOTS]; sim.substep(P, t, h, RowStore<1>{rows}, a, b, c, d, PlaneGround{}, -1.f, Strided{nullptr, 1}); }
    for (int i = 0; i < 13; ++i) root[i * N + e] = sim.root[i];
    for (int i = 0; i < ND; ++i) { dof[i * N + e] = sim.q[i]; dof[(ND + i) * N + e] = sim.qd[i]; }
}
int main() {
    const int N = 4096, W = N / LANES;
    SimParams P{0.0166f, 2, 4, {0, 0, -9.81f}, 0.02f, 0.f, 10.f, 0.5f, 1.f, 0.f, 1e-6f, 1.f};
    std::vector<float> root(13 * N, 0.f), dof(2 * ND * N, 0.f);
#ifdef HUM
    const float z0 = 1.0f;
#else
    const float z0 = 0.30f;
#endif
    for (int e = 0; e < N; ++e) { root[2 * N + e] = z0 + 0.001f * (e % 50); root[6 * N + e] = 1.f; }
    float *droot, *ddof, *dlamc, *dlaml, *dsens, *ddff; unsigned long long* dst;
    hipMalloc(&droot, root.size() * 4); hipMalloc(&ddof, dof.size() * 4); hipMalloc(&dlamc, 3 * NSPH * N * 4); hipMalloc(&dlaml, ND * N * 4);
    hipMalloc(&dsens, (6 * NSENS + 1) * N * 4); hipMalloc(&ddff, ND * N * 4); hipMalloc(&dst, W * 16 * 8);
    hipMemcpy(droot, root.data(), root.size() * 4, hipMemcpyHostToDevice); hipMemcpy(ddof, dof.data(), dof.size() * 4, hipMemcpyHostToDevice);
    hipMemset(dlamc, 0, 3 * NSPH * N * 4); hipMemset(dlaml, 0, ND * N * 4); hipMemset(dst, 0, W * 16 * 8);
    const size_t lds = LDS_ROWS ? (size_t)Sim<M>::ROW_SLOTS * LANES * 4 : 0;
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int rep = 0; rep < 5; ++rep) hipLaunchKernelGGL(k, dim3(W), dim3(LANES), lds, 0, N, P, droot, ddof, dlamc, dlaml, dsens, ddff, dst);
    float *proot, *pdof;   // pristine device copies of the initial state
    hipMalloc(&proot, root.size() * 4); hipMalloc(&pdof, dof.size() * 4);
    hipMemcpy(proot, root.data(), root.size() * 4, hipMemcpyHostToDevice); hipMemcpy(pdof, dof.data(), dof.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t ev0, ev1;
    hipEventCreate(&ev0); hipEventCreate(&ev1);
    hipDeviceSynchronize();
    hipEventRecord(ev0, 0);
    for (int rep = 0; rep < 20; ++rep) {
        hipMemcpyAsync(droot, proot, root.size() * 4, hipMemcpyDeviceToDevice, 0);
        hipMemcpyAsync(ddof, pdof, dof.size() * 4, hipMemcpyDeviceToDevice, 0);
    }
    hipEventRecord(ev1, 0);
    hipDeviceSynchronize();
    float ms0 = 0.f;
    hipEventElapsedTime(&ms0, ev0, ev1);
    hipDeviceSynchronize();
    hipEventRecord(ev0, 0);
    for (int rep = 0; rep < 20; ++rep) {
        // same initial state every launch: the timing must not depend on where the robots have fallen to
        hipMemcpyAsync(droot, proot, root.size() * 4, hipMemcpyDeviceToDevice, 0);
        hipMemcpyAsync(ddof, pdof, dof.size() * 4, hipMemcpyDeviceToDevice, 0);
        hipLaunchKernelGGL(k, dim3(W), dim3(LANES), lds, 0, N, P, droot, ddof, dlamc, dlaml, dsens, ddff, dst);
    }
    hipEventRecord(ev1, 0);
    hipError_t err = hipDeviceSynchronize();
    float ms = 0.f;
    hipEventElapsedTime(&ms, ev0, ev1);
    printf("kernel: %.1f us per launch (copies alone %.1f us)\n", 1e3f * (ms - ms0) / 20, 1e3f * ms0 / 20);
    if (err != hipSuccess) { printf("hip error %s\n", hipGetErrorString(err)); return 1; }
    std::vector<unsigned long long> st(W * 16);
    hipMemcpy(st.data(), dst, st.size() * 8, hipMemcpyDeviceToHost);
    const char* names[8] = {"stage warm", "tree pass (FK+dyn+H)", "factor + w", "limit rows", "contact rows", "warm apply", "PGS", "finish"};
    double tot = 0;
    for (int p = 0; p < 8; ++p) {
        double s = 0;
        for (int w = 0; w < W; ++w) s += (double)(st[w * 16 + p + 1] - st[w * 16 + p]);
        s /= W; tot += s;
        printf("%-24s %10.0f ticks\n", names[p], s);
    }
    printf("%-24s %10.0f ticks (s_memtime, 100 MHz on gfx9 => x10 ns)\n", "sum", tot);
    return 0;
}
