// kernels_articulation.hip -- the Articulation task: a robot compiled at RUN TIME from whatever file gym.load_asset names (assets/runtime.py emits
// gen/model_articulation.h from the parsed description and rebuilds this one translation unit; the stock library carries the AMP humanoid,
// reference amp/humanoid_amp_base.py:177).  gym.simulate with efforts and per-dof position drives, the rigid-body / Jacobian / mass-matrix tensors.
// The one-wave sub-step in whichever row-store form the robot's size selects (core/engine.hpp: static rows in LDS, the compact store, or rows in
// scratch) -- the limb-per-wave forms need role tables that are dealt per robot at build time.
#include <cstdlib>
#include "step_kernels.hpp"
#include "arena_layout.hpp"
#include "gen/model_articulation.h"
#include "tasks/articulation.hpp"

ArticulationMeta mi_articulation_meta() { return ArticulationMeta{ModelArticulation::ND, ModelArticulation::NB, ModelArticulation::NSENS, ModelArticulation::NSPH, ModelArticulation::FIXED}; }

namespace mi {

using AM = ModelArticulation;
static_assert(sizeof(MiArticulationParams) == sizeof(ArticulationParams), "MiArticulationParams layout");
static_assert(AM::ND <= kMaxDof, "MI_MAX_DOF dofs");

__global__ __launch_bounds__(64) void articulation_substep_kernel(View v, SimParams P, ArticulationParams p) {
    extern __shared__ float lds_rows[];
    constexpr int LANES = Sim<AM>::LANES;
    const int e = xcd_env_base<LANES>(blockIdx.x) + threadIdx.x;
    if (e >= v.N) return;
    constexpr bool PRESTAGE = rows_fit_lds<AM>() && Sim<AM>::STAGES_LAM;
    if constexpr (PRESTAGE) prestage_warm_start<AM>(v, e, lds_rows);
    if constexpr (rows_fit_lds<AM>()) {
        if constexpr (LANES == 64) articulation_substep_env<AM>(v, P, p, e, RowStore<64>(lds_rows + threadIdx.x), PRESTAGE);
        else articulation_substep_env<AM>(v, P, p, e, RowStore<LANES>{lds_rows + threadIdx.x}, PRESTAGE);
    } else {
        float rows[Sim<AM>::ROW_SLOTS];
        articulation_substep_env<AM>(v, P, p, e, RowStore<1>{rows}, false);
    }
}
// the same for an env that holds free / static boxes beside a fixed-base actor (core/scene_engine.hpp).  A kernel of its own: the scene's code does
// not touch the register allocation of the kernel above.  SCENE_LANES envs per workgroup, one per lane: the sub-step is a long serial program per
// env (narrow phase, row build and sweeps walk data-dependent contact lists), so a batch is spread over many small workgroups -- 4096 envs are 512
// workgroups of 8, two resident per CU -- and the row store (8 KB per env: 48 contact slots) plus last sub-step's warm-start table live in LDS
// as [slot][lane]: in a per-lane array (scratch) every one of the sub-step's ~10^4 row accesses was a trip to memory (2.1 ms per sub-step at any
// batch size, profiles/r5t_scene_time.txt).
constexpr int SCENE_WARM = 4 * 48;
constexpr int SCENE_ROWS = SceneRows<AM>::value;
// (a robot with long kinematic chains has wider contact slots: the Kuka + Allegro's 23 dofs need 99 KB at 8 envs -- one workgroup per CU then; 4 envs per
//  workgroup if even that does not fit)
constexpr int SCENE_LANES_MAX = ((size_t)(SCENE_ROWS + SCENE_WARM) * 8 * sizeof(float) <= 160 * 1024) ? 8 : 4;
// Round 6: LANES envs per workgroup is a launch-time choice (scene_lanes below: 4 or 8).  Measured on FrankaCubeStack (profiles/r6s_scene_lanes_ab.txt,
// ms per gym.simulate() at 1024 / 4096 / 16384 envs): 8 lanes 0.80 / 0.88 / 2.55, 4 lanes 0.74 / 1.03 / 2.70, 2 lanes 0.77 / 1.52 / 4.37, 1 lane
// 0.97 / 2.31 / 7.86 -- a wavefront with ONE env is slower than one with eight once every SIMD has one: the wavefronts do not wait for each
// other's divergent loops, they compete for instruction fetch (the kernel holds the whole register file, one wavefront per SIMD, and its code is far
// larger than the instruction cache).
template <int LANES>
__global__ __launch_bounds__(64) void articulation_scene_substep_kernel(View v, SimParams P, ArticulationParams p) {
    extern __shared__ float lds_scene[];
    const int e = blockIdx.x * LANES + threadIdx.x;
    if (e >= v.N) return;
    if constexpr (AM::FIXED == 1) {
        float* rows = lds_scene + threadIdx.x;
        float* warm = lds_scene + SCENE_ROWS * LANES + threadIdx.x;
        const int N = v.N;
        for (int k = 0; k < SCENE_WARM; ++k) warm[k * LANES] = v.scene_warm[(size_t)k * N + e];
        articulation_scene_substep_env<AM>(v, P, p, e, RowStore<LANES>{rows}, Strided{warm, LANES});
        for (int k = 0; k < SCENE_WARM; ++k) v.scene_warm[(size_t)k * N + e] = warm[k * LANES];
    }
}
__global__ void articulation_reset_kernel(View v, ArticulationParams p, const long long* __restrict__ ids, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (ids ? n : v.N)) return;
    const int e = ids ? (int)ids[i] : i;
    if (e < 0 || e >= v.N) return;
    articulation_reset_env<AM>(v, p, e);
}
__global__ __launch_bounds__(64) void articulation_body_states_kernel(View v) {
    const int e = blockIdx.x * 64 + threadIdx.x;
    const int N = v.N;
    if (e >= N) return;
    const int body = blockIdx.y;
    sfor<AM::NB>([&](auto B_) MI_LAMBDA {
        constexpr int b = B_;
        if (body == b) {         // wave-uniform
            Sim<AM> sim;
            sfor<13>([&](auto K) MI_LAMBDA { sim.root[K] = v.root[K * N + e]; });
            sfor<AM::ND>([&](auto D) MI_LAMBDA {
                if constexpr (Sim<AM>::on_chain(AM::dof_body[D], b)) { sim.q[D] = v.dof[D * N + e]; sim.qd[D] = v.dof[(AM::ND + D) * N + e]; }
            });
            float o[13];
            sim.template body_state<b>(o);
            sfor<13>([&](auto K) MI_LAMBDA { v.body_state[(b * 13 + K) * N + e] = o[K]; });
        }
    });
}
template <int b>
__global__ __launch_bounds__(64) void articulation_jacobian_kernel(View v, float* __restrict__ out) {
    const int e = blockIdx.x * 64 + threadIdx.x;
    const int N = v.N;
    if (e >= N) return;
    constexpr int NV = AM::NV;
    Sim<AM> sim;
    sfor<13>([&](auto K) MI_LAMBDA { sim.root[K] = v.root[K * N + e]; });
    sfor<AM::ND>([&](auto D) MI_LAMBDA {
        sim.qd[D] = 0.f;
        if constexpr (Sim<AM>::on_chain(AM::dof_body[D], b)) sim.q[D] = v.dof[D * N + e]; else sim.q[D] = 0.f;
    });
    float J[6 * NV];
    sim.template body_jacobian<b>(J);
    float* o = out + ((size_t)e * AM::NB + b) * 6 * NV;
    sfor<6 * NV>([&](auto K) MI_LAMBDA { o[K] = J[K]; });
}
__global__ __launch_bounds__(64) void articulation_mass_matrix_kernel(View v, SimParams P, float* __restrict__ out) {
    const int e = blockIdx.x * 64 + threadIdx.x;
    const int N = v.N;
    if (e >= N) return;
    constexpr int NV = AM::NV;
    Sim<AM> sim;
    sfor<13>([&](auto K) MI_LAMBDA { sim.root[K] = v.root[K * N + e]; });
    sfor<AM::ND>([&](auto D) MI_LAMBDA { sim.q[D] = v.dof[D * N + e]; sim.qd[D] = 0.f; });
    float H[NV * NV];
    sim.mass_matrix(P, H);
    float* o = out + (size_t)e * NV * NV;
    for (int k = 0; k < NV * NV; ++k) o[k] = H[k];
}

// envs per workgroup of the scene sub-step: MI_SCENE_LANES (4, 8: A/B runs) or by batch size
static int scene_lanes(const int N) {
    static const int forced = [] { const char* e = getenv("MI_SCENE_LANES"); return e ? atoi(e) : 0; }();
    int lanes = forced > 0 ? forced : (N <= 1024 ? 4 : 8);
    if (lanes != 4 && lanes != 8) lanes = 8;
    return lanes > SCENE_LANES_MAX ? SCENE_LANES_MAX : lanes;
}
hipError_t launch_simulate_articulation(const View& v, const SimParams& P, const ArticulationParams& p, hipStream_t s) {
    if (articulation_has_scene(p)) {
        if (AM::FIXED != 1) return hipErrorInvalidValue;
        static_assert((size_t)(SCENE_ROWS + SCENE_WARM) * SCENE_LANES_MAX * sizeof(float) <= 160 * 1024, "the scene's row store of one workgroup fits the LDS of a CU");
        const int lanes = scene_lanes(v.N);
        hipError_t err = hipSuccess;
        sfor<2>([&](auto L_) {
            constexpr int LANES = 4 << decltype(L_)::value;
            if constexpr (LANES <= SCENE_LANES_MAX) {
                if (lanes == LANES && err == hipSuccess) {
                    constexpr size_t lds = (size_t)(SCENE_ROWS + SCENE_WARM) * LANES * sizeof(float);
                    static unsigned long long scene_configured = 0ull;
                    err = ensure_dynamic_lds((const void*)articulation_scene_substep_kernel<LANES>, lds, &scene_configured);
                    if (err != hipSuccess) return;
                    for (int i = 0; i < P.substeps; ++i)
                        hipLaunchKernelGGL(articulation_scene_substep_kernel<LANES>, dim3((v.N + LANES - 1) / LANES), dim3(LANES), lds, s, v, P, p);
                }
            }
        });
        return err != hipSuccess ? err : hipGetLastError();
    }
    constexpr size_t lds = rows_fit_lds<AM>() ? lds_bytes<AM>() : 0;
    constexpr int LANES = Sim<AM>::LANES;
    static unsigned long long configured = 0ull;
    if (hipError_t e = ensure_dynamic_lds((const void*)articulation_substep_kernel, lds, &configured); e != hipSuccess) return e;
    for (int i = 0; i < P.substeps; ++i) hipLaunchKernelGGL(articulation_substep_kernel, dim3(xcd_grid<LANES>(v.N)), dim3(LANES), lds, s, v, P, p);
    return hipGetLastError();
}
hipError_t launch_reset_articulation(const View& v, const ArticulationParams& p, const long long* ids, int n, hipStream_t s) {
    const int cnt = ids ? n : v.N;
    hipLaunchKernelGGL(articulation_reset_kernel, dim3((cnt + 127) / 128), dim3(128), 0, s, v, p, ids, n);
    return hipGetLastError();
}
hipError_t launch_body_states_articulation(const View& v, hipStream_t s) {
    hipLaunchKernelGGL(articulation_body_states_kernel, dim3((v.N + 63) / 64, AM::NB), dim3(64), 0, s, v);
    return hipGetLastError();
}
hipError_t launch_kinematics_articulation(const View& v, const SimParams& P, float* out_j, float* out_h, hipStream_t s) {
    if (out_j) sfor<AM::NB>([&](auto B_) { hipLaunchKernelGGL((articulation_jacobian_kernel<decltype(B_)::value>), dim3((v.N + 63) / 64), dim3(64), 0, s, v, out_j); });
    if (out_h) hipLaunchKernelGGL(articulation_mass_matrix_kernel, dim3((v.N + 63) / 64), dim3(64), 0, s, v, P, out_h);
    return hipGetLastError();
}

}  // namespace mi
