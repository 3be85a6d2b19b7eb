// pgs_lanes_bench.hip -- north_star: "wavefront shuffles for per-body reductions".  At the BASELINE sizes the limb-per-wave kernels run with 16 (Ant, ANYmal) of
// a wave's 64 lanes live.  Measured here, for the part of an Ant leg role that is a chain of dependent reductions -- the projected Gauss-Seidel sweeps over
// the role's own rows (engine_mw.hpp P4: 14 rows of <= 8 chain entries, 4 sweeps) -- whether FOUR LANES PER ENV shorten that chain:
//
//   (A) one env per lane (what the engine runs): rows in LDS as [slot][lane], the 8 whitened-velocity coordinates w in registers; per row
//       v = g . w (8 dependent FMAs), the impulse update, w += g dlam (8 independent FMAs).  16 live lanes per wave.
//   (B) four lanes per env: lane j of an env's quad holds entries 2 j, 2 j + 1 of every row and of w; per row two FMAs, a quad reduction with two DPP
//       steps (quad_perm: no LDS round trip), the impulse update replicated in the quad, two FMAs.  64 live lanes per wave: the same 16 envs.
//
// Same data, same grid (4096 envs: 256 workgroups x 4 waves, each wave one leg block), results compared.  Output: microseconds per launch and the ratio.
// Build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 -o pgs_lanes_bench pgs_lanes_bench.hip && ./pgs_lanes_bench
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

constexpr int R = 14, C = 8, SWEEPS = 4, E = 16, ROLES = 4;
constexpr int ROWF = C + 3;                     // g[8], 1 / a, v*, lambda
// data: [N][ROLES][R][ROWF] rows, [N][ROLES][C] w in; out: [N][ROLES][C] w, [N][ROLES][R] lambda

// ------------------------------------------------------------------------------------------------ (A) one env per lane
__global__ __launch_bounds__(256) void pgs_one_lane(const float* __restrict__ rows_in, const float* __restrict__ w_in, float* __restrict__ w_out,
                                                    float* __restrict__ lam_out, int N, int nsweeps) {
    __shared__ float lds[ROLES][R * ROWF][E];   // [slot][lane] per role wave
    const int lane = threadIdx.x, role = threadIdx.y;
    if (lane >= E) return;
    const int e = blockIdx.x * E + lane;
    if (e >= N) return;
    const float* src = rows_in + ((size_t)e * ROLES + role) * R * ROWF;
    for (int k = 0; k < R * ROWF; ++k) lds[role][k][lane] = src[k];
    float w[C];
#pragma unroll
    for (int c = 0; c < C; ++c) w[c] = w_in[((size_t)e * ROLES + role) * C + c];
    for (int it = 0; it < nsweeps; ++it) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            float g[C];
#pragma unroll
            for (int c = 0; c < C; ++c) g[c] = lds[role][r * ROWF + c][lane];
            const float ainv = lds[role][r * ROWF + C][lane], vt = lds[role][r * ROWF + C + 1][lane], lam = lds[role][r * ROWF + C + 2][lane];
            float v = 0.f;
#pragma unroll
            for (int c = 0; c < C; ++c) v = fmaf(g[c], w[c], v);
            const float nl = fmaxf(lam - (v - vt) * ainv, 0.f), dl = nl - lam;
            lds[role][r * ROWF + C + 2][lane] = nl;
#pragma unroll
            for (int c = 0; c < C; ++c) w[c] = fmaf(g[c], dl, w[c]);
        }
    }
#pragma unroll
    for (int c = 0; c < C; ++c) w_out[((size_t)e * ROLES + role) * C + c] = w[c];
    for (int r = 0; r < R; ++r) lam_out[((size_t)e * ROLES + role) * R + r] = lds[role][r * ROWF + C + 2][lane];
}

// ------------------------------------------------------------------------------------------------ (B) four lanes per env, DPP quad reduction
__device__ __forceinline__ float quad_sum(float x) {
    // quad_perm [1,0,3,2] then [2,3,0,1]: two DPP moves, no LDS
    x += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, true));
    x += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0x4E, 0xF, 0xF, true));
    return x;
}
__global__ __launch_bounds__(256) void pgs_four_lanes(const float* __restrict__ rows_in, const float* __restrict__ w_in, float* __restrict__ w_out,
                                                      float* __restrict__ lam_out, int N, int nsweeps) {
    // per role wave: entries [r][half][lane64] (lane = env * 4 + j holds entries 2 j + half), 1 / a and v* [r][2][env].  The impulses live in
    // registers, replicated in the quad: every lane of a quad computes the same update from the same reduced v (the two DPP adds commute, so the
    // four results are bit-identical) -- keeping them in LDS with one writer per quad would be an unsynchronised exchange between lanes, which
    // the compiler may (and did) resolve by keeping a stale copy in the reading lanes.
    __shared__ float ent[ROLES][R * 2][64];
    __shared__ float sca[ROLES][R * 2][E];
    const int lane = threadIdx.x, role = threadIdx.y, j = lane & 3, el = lane >> 2;
    const int e = blockIdx.x * E + el;
    if (e >= N) return;
    const float* src = rows_in + ((size_t)e * ROLES + role) * R * ROWF;
    float lam[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        ent[role][2 * r][lane] = src[r * ROWF + 2 * j];
        ent[role][2 * r + 1][lane] = src[r * ROWF + 2 * j + 1];
        if (j < 2) sca[role][2 * r + j][el] = src[r * ROWF + C + j];
        lam[r] = src[r * ROWF + C + 2];
    }
    __syncthreads();
    float w0 = w_in[((size_t)e * ROLES + role) * C + 2 * j], w1 = w_in[((size_t)e * ROLES + role) * C + 2 * j + 1];
    for (int it = 0; it < nsweeps; ++it) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const float g0 = ent[role][2 * r][lane], g1 = ent[role][2 * r + 1][lane];
            const float ainv = sca[role][2 * r][el], vt = sca[role][2 * r + 1][el];
            const float v = quad_sum(fmaf(g1, w1, g0 * w0));
            const float nl = fmaxf(lam[r] - (v - vt) * ainv, 0.f), dl = nl - lam[r];
            lam[r] = nl;
            w0 = fmaf(g0, dl, w0);
            w1 = fmaf(g1, dl, w1);
        }
    }
    w_out[((size_t)e * ROLES + role) * C + 2 * j] = w0;
    w_out[((size_t)e * ROLES + role) * C + 2 * j + 1] = w1;
#pragma unroll
    for (int r = 0; r < R; ++r) if ((r & 3) == j) lam_out[((size_t)e * ROLES + role) * R + r] = lam[r];
}

// the same sweeps on the host, in double: what both kernels are checked against
static void pgs_host(const float* rows, const float* w_in, double* w, double* lam, int nsweeps) {
    for (int c = 0; c < C; ++c) w[c] = w_in[c];
    for (int r = 0; r < R; ++r) lam[r] = rows[r * ROWF + C + 2];
    for (int it = 0; it < nsweeps; ++it)
        for (int r = 0; r < R; ++r) {
            double v = 0;
            for (int c = 0; c < C; ++c) v += (double)rows[r * ROWF + c] * w[c];
            const double nl = fmax(lam[r] - (v - rows[r * ROWF + C + 1]) * rows[r * ROWF + C], 0.0), dl = nl - lam[r];
            lam[r] = nl;
            for (int c = 0; c < C; ++c) w[c] += (double)rows[r * ROWF + c] * dl;
        }
}

int main() {
    const int N = 4096;
    std::vector<float> rows((size_t)N * ROLES * R * ROWF), w((size_t)N * ROLES * C);
    srand(3);
    auto rnd = []() { return (float)rand() / RAND_MAX * 2.f - 1.f; };
    for (size_t i = 0; i < (size_t)N * ROLES * R; ++i) {
        float a = 1e-6f;
        for (int c = 0; c < C; ++c) { const float g = rnd(); rows[i * ROWF + c] = g; a += g * g; }
        rows[i * ROWF + C] = 1.f / a; rows[i * ROWF + C + 1] = 0.3f * rnd(); rows[i * ROWF + C + 2] = 0.f;
    }
    for (auto& x : w) x = rnd();
    float *d_rows, *d_w, *d_wo[2], *d_lo[2];
    CHECK(hipMalloc(&d_rows, rows.size() * 4)); CHECK(hipMalloc(&d_w, w.size() * 4));
    for (int k = 0; k < 2; ++k) { CHECK(hipMalloc(&d_wo[k], w.size() * 4)); CHECK(hipMalloc(&d_lo[k], (size_t)N * ROLES * R * 4)); }
    CHECK(hipMemcpy(d_rows, rows.data(), rows.size() * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(d_w, w.data(), w.size() * 4, hipMemcpyHostToDevice));
    const dim3 grid(N / E), block(64, ROLES);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    // two sweep counts: the product's 4 (with the fixed cost of bringing the rows into LDS) and 36; their difference / 32 is one sweep by itself
    float us[2][2];
    const int counts[2] = {SWEEPS, SWEEPS + 32};
    for (int c = 1; c >= 0; --c)
    for (int k = 0; k < 2; ++k) {
        for (int rep = 0; rep < 20; ++rep) {
            if (k == 0) hipLaunchKernelGGL(pgs_one_lane, grid, block, 0, 0, d_rows, d_w, d_wo[0], d_lo[0], N, counts[c]);
            else hipLaunchKernelGGL(pgs_four_lanes, grid, block, 0, 0, d_rows, d_w, d_wo[1], d_lo[1], N, counts[c]);
        }
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        const int reps = 500;
        for (int rep = 0; rep < reps; ++rep) {
            if (k == 0) hipLaunchKernelGGL(pgs_one_lane, grid, block, 0, 0, d_rows, d_w, d_wo[0], d_lo[0], N, counts[c]);
            else hipLaunchKernelGGL(pgs_four_lanes, grid, block, 0, 0, d_rows, d_w, d_wo[1], d_lo[1], N, counts[c]);
        }
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        us[c][k] = 1e3f * ms / reps;
    }
    std::vector<float> a(w.size()), b(w.size()), la((size_t)N * ROLES * R), lb((size_t)N * ROLES * R);
    CHECK(hipMemcpy(a.data(), d_wo[0], a.size() * 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(b.data(), d_wo[1], b.size() * 4, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(la.data(), d_lo[0], la.size() * 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(lb.data(), d_lo[1], lb.size() * 4, hipMemcpyDeviceToHost));
    double dw = 0, dl = 0, act = 0, dha = 0, dhb = 0;
    for (int i = 0; i < N * ROLES; ++i) {            // both against the host's double-precision sweeps (the last launches ran SWEEPS sweeps)
        double hw[C], hl[R];
        pgs_host(&rows[(size_t)i * R * ROWF], &w[(size_t)i * C], hw, hl, SWEEPS);
        for (int c = 0; c < C; ++c) { dha = fmax(dha, fabs(hw[c] - a[(size_t)i * C + c])); dhb = fmax(dhb, fabs(hw[c] - b[(size_t)i * C + c])); }
        for (int r = 0; r < R; ++r) { dha = fmax(dha, fabs(hl[r] - la[(size_t)i * R + r])); dhb = fmax(dhb, fabs(hl[r] - lb[(size_t)i * R + r])); }
    }
    for (size_t i = 0; i < a.size(); ++i) dw = fmax(dw, fabs((double)a[i] - b[i]));
    for (size_t i = 0; i < la.size(); ++i) { dl = fmax(dl, fabs((double)la[i] - lb[i])); act += la[i] > 0.f; }
    printf("PGS sweeps of an Ant leg block (%d rows x %d entries, %d sweeps), %d envs, 256 workgroups x 4 waves, incl. the load of the rows into LDS:\n", R, C, SWEEPS, N);
    printf("  (A) one env per lane, 16 live lanes / wave : %.2f us per launch (4 sweeps), %.2f us (36 sweeps) -> %.3f us per sweep\n", us[0][0], us[1][0], (us[1][0] - us[0][0]) / 32.f);
    printf("  (B) four lanes per env, DPP quad reduction  : %.2f us per launch (4 sweeps), %.2f us (36 sweeps) -> %.3f us per sweep\n", us[0][1], us[1][1], (us[1][1] - us[0][1]) / 32.f);
    printf("  A / B: %.2fx per launch at 4 sweeps, %.2fx per sweep\n", us[0][0] / us[0][1], (us[1][0] - us[0][0]) / (us[1][1] - us[0][1]));
    printf("  max |w_A - w_B| = %.2e, max |lambda_A - lambda_B| = %.2e, active rows %.0f %%\n", dw, dl, 100.0 * act / la.size());
    printf("  against the host's sweeps in double: (A) %.2e, (B) %.2e\n", dha, dhb);
    return (dw < 1e-4 && dl < 1e-4 && dha < 1e-4 && dhb < 1e-4) ? 0 : 1;
}
