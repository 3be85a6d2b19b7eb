// delassus_mfma_bench.hip -- north_star: "MFMA used only for the small dense mass-matrix / Jacobian contractions".  Measured here, for the
// Humanoid's constraint solve at 8192 envs, which of two formulations of the SAME solve is faster on gfx950:
//
//   (A) Delassus form on the matrix cores: the whitened rows G [R x NV] (R = 66 rows: 21 joint limits + 15 contacts x 3; NV = 27) are
//       dense; A = G G^T (66 x 66, padded to 80 x 80 = 5 x 5 tiles of 16 x 16, K = 27 padded to 28) is built with
//       v_mfma_f32_16x16x4_f32, kept in LDS, and 4 projected Gauss-Seidel sweeps run on A (lambda-space: r_i = b_i + sum_j A_ij lam_j),
//       4 envs per wave (16 lanes per env share a row's dot product, DPP reduction).
//   (B) the engine's chain-sparse form: one env per lane, rows over their kinematic chain (<= 15 of the 27 coordinates, the Humanoid's
//       own chain tables), rows in LDS as [slot][lane], the whitened velocity w [27] in registers, 4 sweeps of
//       v_n = g . w;  lam update;  w += g dlam  -- what core/engine.hpp / engine_mwc.hpp run (here as ONE sequence on one wave of 32 envs).
//
// Both kernels do the row-space work only (no tree pass, no row build): (A) additionally pays for dense rows in the row build, which is
// not charged here.  Output: microseconds per launch at 8192 envs and the ratio.  Build + run: tools/mfma/run.sh (needs a GPU).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../isaacgymenvs_amd/csrc/gen/model_humanoid.h"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

using M = ModelHumanoid;
constexpr int NV = 27, R = 66, RP = 80, KP = 28, NLIM = 21, NCON = 15, SWEEPS = 4;
typedef float float4_ __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------ (A) Delassus form, MFMA build + PGS on A from LDS
// G: [N][RP][KP] (rows >= R and columns >= NV are zero), b: [N][RP], lam out: [N][RP]
constexpr int ENVS_PER_WAVE = 4, ALD = RP + 1;      // LDS row stride of A (+1: conflict-free column walks)
__global__ __launch_bounds__(64) void delassus_mfma_kernel(const float* __restrict__ G, const float* __restrict__ b, float* __restrict__ lam_out, int N) {
    extern __shared__ float lds[];                  // [ENVS_PER_WAVE][RP][ALD] A  |  [ENVS_PER_WAVE][RP][KP] G staging
    float* A = lds;
    float* Gs = lds + ENVS_PER_WAVE * RP * ALD;
    const int lane = threadIdx.x, e0 = blockIdx.x * ENVS_PER_WAVE;
    // stage G of the wave's 4 envs (coalesced)
    for (int i = lane; i < ENVS_PER_WAVE * RP * KP; i += 64) {
        const int e = e0 + i / (RP * KP);
        Gs[i] = e < N ? G[(size_t)e * RP * KP + i % (RP * KP)] : 0.f;
    }
    __syncthreads();
    // A = G G^T tile by tile (lower triangle + diagonal, mirrored on store): A-operand lane l -> G[16 I + l % 16][4 kk + l / 16]
    for (int el = 0; el < ENVS_PER_WAVE; ++el) {
        const float* g = Gs + el * RP * KP;
        float* a = A + el * RP * ALD;
        for (int I = 0; I < RP / 16; ++I)
            for (int J = 0; J <= I; ++J) {
                float4_ c = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kk = 0; kk < KP / 4; ++kk) {
                    const float av = g[(16 * I + lane % 16) * KP + 4 * kk + lane / 16];
                    const float bv = g[(16 * J + lane % 16) * KP + 4 * kk + lane / 16];
                    c = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, c, 0, 0, 0);
                }
                // lane l holds C[i = 4 (l / 16) + r][j = l % 16]
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = 16 * I + 4 * (lane / 16) + r, j = 16 * J + lane % 16;
                    a[i * ALD + j] = c[r];
                    a[j * ALD + i] = c[r];
                }
            }
    }
    __syncthreads();
    // PGS on A: 16 lanes per env; lane q of the env's group owns columns q, q + 16, ... (its lam entries live in registers)
    const int el = lane / 16, q = lane % 16, e = e0 + el;
    const float* a = A + el * RP * ALD;
    float lam[RP / 16], bb[RP / 16];
#pragma unroll
    for (int t = 0; t < RP / 16; ++t) { lam[t] = 0.f; bb[t] = e < N ? b[(size_t)e * RP + q + 16 * t] : 0.f; }
    for (int it = 0; it < SWEEPS; ++it) {
        for (int i = 0; i < R; ++i) {
            float s = 0.f;
#pragma unroll
            for (int t = 0; t < RP / 16; ++t) s += a[i * ALD + q + 16 * t] * lam[t];
            // reduce over the 16 lanes of the env
            s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64); s += __shfl_xor(s, 8, 64);
            const int ti = i / 16, qi = i % 16;
            // the owner of column i holds lam_i and b_i: everybody computes the update with broadcast values
            float li = 0.f, bi = 0.f;
#pragma unroll
            for (int t = 0; t < RP / 16; ++t) if (t == ti) { li = lam[t]; bi = bb[t]; }
            li = __shfl(li, el * 16 + qi, 64); bi = __shfl(bi, el * 16 + qi, 64);
            const float aii = a[i * ALD + i];
            float nl = li - (s + bi) / (aii + 1e-6f);
            nl = (i < NLIM || (i - NLIM) % 3 == 0) ? fmaxf(nl, 0.f) : nl;    // limit rows / contact normals are one-sided
#pragma unroll
            for (int t = 0; t < RP / 16; ++t) if (t == ti && q == qi) lam[t] = nl;
        }
    }
    if (e < N)
#pragma unroll
        for (int t = 0; t < RP / 16; ++t) lam_out[(size_t)e * RP + q + 16 * t] = lam[t];
}

// ------------------------------------------------------------------ (B) chain-sparse form, one env per lane
// rows: limit row of dof d over [gi, anc(gi)...]; contact c (3 rows) over the chain of body (c % (NB - 1)) + 1.  Store [slot][lane].
template <int I, int N_, class F> struct SFor { static __device__ __forceinline__ void run(F& f) { f(std::integral_constant<int, I>{}); SFor<I + 1, N_, F>::run(f); } };
template <int N_, class F> struct SFor<N_, N_, F> { static __device__ __forceinline__ void run(F&) {} };
template <int N_, class F> __device__ __forceinline__ void sfor(F&& f) { SFor<0, N_, F>::run(f); }
constexpr int CH = M::MAXCHAIN;                      // 15
constexpr int con_body(int c) { return c % (M::NB - 1) + 1; }
constexpr int SLOTS = NLIM * (CH + 2) + NCON * (3 * CH + 4);
constexpr int LANES_B = 32;
__global__ __launch_bounds__(64) void chain_sparse_kernel(const float* __restrict__ rows_in, const float* __restrict__ w_in, float* __restrict__ w_out, int N) {
    extern __shared__ float lds[];                  // [SLOTS][LANES_B]
    const int lane = threadIdx.x, e = blockIdx.x * LANES_B + lane;
    if (lane >= LANES_B || e >= N) return;
    float* rs = lds + lane;
    for (int s = 0; s < SLOTS; ++s) rs[s * LANES_B] = rows_in[(size_t)s * N + e];       // (the row build of the real kernel writes these)
    float w[NV];
    sfor<NV>([&](auto I) { w[I] = w_in[(size_t)decltype(I)::value * N + e]; });
    for (int it = 0; it < SWEEPS; ++it) {
        int zero;
        asm volatile("s_mov_b32 %0, 0" : "=s"(zero));
        float* r = rs + zero;
        sfor<NLIM>([&](auto D_) {
            constexpr int d = decltype(D_)::value, gi = M::OFF + d, base = d * (CH + 2);
            float g[CH];
            sfor<M::nanc[gi] + 1>([&](auto K) { g[K] = r[(base + K) * LANES_B]; });
            float vn = g[0] * w[gi];
            sfor<M::nanc[gi]>([&](auto A_) { vn += g[1 + A_] * w[M::anc[gi][A_]]; });
            const float lo = r[(base + CH) * LANES_B], ainv = r[(base + CH + 1) * LANES_B];
            const float nl = fmaxf(lo - vn * ainv, 0.f), dl = nl - lo;
            r[(base + CH) * LANES_B] = nl;
            w[gi] += g[0] * dl;
            sfor<M::nanc[gi]>([&](auto A_) { w[M::anc[gi][A_]] += g[1 + A_] * dl; });
        });
        sfor<NCON>([&](auto C_) {
            constexpr int c = decltype(C_)::value, b = con_body(c), base = NLIM * (CH + 2) + c * (3 * CH + 4);
            float g[3][CH], lm[3];
            sfor<3>([&](auto K) { sfor<M::chain_len[b]>([&](auto Q) { g[K][Q] = r[(base + K * CH + Q) * LANES_B]; }); lm[K] = r[(base + 3 * CH + 1 + K) * LANES_B]; });
            const float ainv = r[(base + 3 * CH) * LANES_B];
            auto dot = [&](auto K) { float s = 0.f; sfor<M::chain_len[b]>([&](auto Q) { s += g[K][Q] * w[M::chain[b][Q]]; }); return s; };
            auto add = [&](auto K, float dl) { sfor<M::chain_len[b]>([&](auto Q) { w[M::chain[b][Q]] += g[K][Q] * dl; }); };
            const float ln = fmaxf(lm[0] - dot(std::integral_constant<int, 0>{}) * ainv, 0.f);
            add(std::integral_constant<int, 0>{}, ln - lm[0]);
            float lt[2];
            sfor<2>([&](auto K) { const float dl = -dot(std::integral_constant<int, 1 + decltype(K)::value>{}) * ainv; lt[K] = lm[1 + K] + dl; add(std::integral_constant<int, 1 + decltype(K)::value>{}, dl); });
            const float lim = ln, n2 = lt[0] * lt[0] + lt[1] * lt[1];
            const float sc = (n2 > lim * lim) ? lim * __builtin_amdgcn_rsqf(fmaxf(n2, 1e-30f)) : 1.f;
            r[(base + 3 * CH + 1) * LANES_B] = ln;
            sfor<2>([&](auto K) { const float nl = lt[K] * sc; r[(base + 3 * CH + 2 + K) * LANES_B] = nl; add(std::integral_constant<int, 1 + decltype(K)::value>{}, nl - lt[K]); });
        });
    }
    sfor<NV>([&](auto I) { w_out[(size_t)decltype(I)::value * N + e] = w[I]; });
}

int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 8192, reps = 200;
    std::vector<float> G((size_t)N * RP * KP, 0.f), bvec((size_t)N * RP, 0.f), rows((size_t)SLOTS * N), w((size_t)NV * N);
    srand(1);
    auto rnd = [] { return (float)rand() / RAND_MAX - 0.5f; };
    for (int e = 0; e < N; ++e) for (int r = 0; r < R; ++r) { for (int k = 0; k < NV; ++k) G[((size_t)e * RP + r) * KP + k] = rnd(); bvec[(size_t)e * RP + r] = rnd(); }
    for (auto& x : rows) x = 0.2f * rnd();
    for (auto& x : w) x = rnd();
    float *dG, *db, *dl, *drows, *dw, *dwo;
    CHECK(hipMalloc(&dG, G.size() * 4)); CHECK(hipMalloc(&db, bvec.size() * 4)); CHECK(hipMalloc(&dl, bvec.size() * 4));
    CHECK(hipMalloc(&drows, rows.size() * 4)); CHECK(hipMalloc(&dw, w.size() * 4)); CHECK(hipMalloc(&dwo, w.size() * 4));
    CHECK(hipMemcpy(dG, G.data(), G.size() * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(db, bvec.data(), bvec.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(drows, rows.data(), rows.size() * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dw, w.data(), w.size() * 4, hipMemcpyHostToDevice));
    const size_t ldsA = (size_t)(ENVS_PER_WAVE * RP * ALD + ENVS_PER_WAVE * RP * KP) * 4, ldsB = (size_t)SLOTS * LANES_B * 4;
    CHECK(hipFuncSetAttribute((const void*)delassus_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsA));
    CHECK(hipFuncSetAttribute((const void*)chain_sparse_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsB));
    hipEvent_t t0, t1;
    CHECK(hipEventCreate(&t0)); CHECK(hipEventCreate(&t1));
    float msA = 0.f, msB = 0.f;
    for (int pass = 0; pass < 2; ++pass) {       // pass 0 warms up
        CHECK(hipEventRecord(t0));
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(delassus_mfma_kernel, dim3((N + ENVS_PER_WAVE - 1) / ENVS_PER_WAVE), dim3(64), ldsA, 0, dG, db, dl, N);
        CHECK(hipEventRecord(t1)); CHECK(hipEventSynchronize(t1)); CHECK(hipEventElapsedTime(&msA, t0, t1));
        CHECK(hipEventRecord(t0));
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(chain_sparse_kernel, dim3((N + LANES_B - 1) / LANES_B), dim3(64), ldsB, 0, drows, dw, dwo, N);
        CHECK(hipEventRecord(t1)); CHECK(hipEventSynchronize(t1)); CHECK(hipEventElapsedTime(&msB, t0, t1));
    }
    CHECK(hipGetLastError());
    const double flopA = (double)N * 15 * (KP / 4) * 2048.0;      // MFMA flops of the A build (15 tiles x 7 k-steps per env)
    printf("Humanoid constraint solve, %d envs, R = %d rows, NV = %d, %d sweeps (row-space work only)\n", N, R, NV, SWEEPS);
    printf("  (A) Delassus form: A = G G^T by v_mfma_f32_16x16x4_f32 (%d MFMAs / env) + PGS on A from LDS, 4 envs / wave : %8.1f us / launch  (MFMA build alone = %.2f TFLOP/s if it were the whole launch)\n",
           15 * (KP / 4), 1e3 * msA / reps, flopA / (1e-3 * msA / reps) / 1e12);
    printf("  (B) chain-sparse rows, one env per lane, w in registers, rows [slot][lane] in LDS (the engine's form, one wave)   : %8.1f us / launch\n", 1e3 * msB / reps);
    printf("  ratio (A) / (B) = %.2f   (LDS per workgroup: A %zu B, B %zu B)\n", msA / msB, ldsA, ldsB);
    return 0;
}
