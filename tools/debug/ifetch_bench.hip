// GPU microbenchmark (tools/debug): issue rate of ONE wavefront per SIMD running straight-line VALU code once (our sub-step kernels) vs a
// small loop body that stays in the instruction cache, VOP2 (4-byte) vs VOP3 (8-byte) encodings, and several waves per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 ifetch_bench.hip -o ifetch_bench     (results: DESIGN.md 5, first paragraph; the table is in git history, round 1)
#include <hip/hip_runtime.h>
#include <cstdio>
// straight-line VALU stream executed once: is a lone wave limited by instruction fetch?
#define R4(x) x x x x
#define R16(x) R4(R4(x))
#define R256(x) R16(R16(x))
#define R4096(x) R16(R256(x))
template <int MODE>
__global__ __launch_bounds__(64) void k(float* out, int reps, unsigned long long* cyc) {
    float a = threadIdx.x, b = a + 1, c = a + 2, d = a + 3;
    const float e = 1.0001f;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; ++r) {
        if (MODE == 0) {   // 16384 instructions, 4 independent chains, VOP2 (4 bytes each)
            R4096(asm volatile("v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e));)
        } else if (MODE == 2) {   // small body (256 x 4 VOP3 = 8 KB): fits the instruction cache
            R256(asm volatile("v_fma_f32 %0, %0, %4, %4\n v_fma_f32 %1, %1, %4, %4\n v_fma_f32 %2, %2, %4, %4\n v_fma_f32 %3, %3, %4, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e));)
        } else if (MODE == 3) {   // dependent chain, small body
            R256(asm volatile("v_fma_f32 %0, %0, %4, %4\n v_fma_f32 %0, %0, %4, %4\n v_fma_f32 %0, %0, %4, %4\n v_fma_f32 %0, %0, %4, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e));)
        } else {           // same with VOP3 encodings (8 bytes each): v_fma
            R4096(asm volatile("v_fma_f32 %0, %0, %4, %4\n v_fma_f32 %1, %1, %4, %4\n v_fma_f32 %2, %2, %4, %4\n v_fma_f32 %3, %3, %4, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e));)
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 64 + threadIdx.x] = a + b + c + d;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE> void run(int waves, int reps, const char* name) {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, waves * 64 * 4); hipMalloc(&cyc, waves * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 3; ++it) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k<MODE>, dim3(waves), dim3(64), 0, 0, out, reps, cyc);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[4096]; hipMemcpy(h, cyc, waves * 8, hipMemcpyDeviceToHost);
    double s = 0; for (int i = 0; i < waves; ++i) s += h[i];
    const double n = ((MODE == 2 || MODE == 3) ? 1024.0 : 16384.0) * reps;
    printf("%s waves=%d reps=%d: %.1f us, %.2f ns/instr, memtime ticks/instr %.3f\n", name, waves, reps, ms * 1e3, ms * 1e6 / n, s / waves / n);
}
int main() {
    run<0>(64, 1, "VOP2 once  ");  run<0>(64, 8, "VOP2 loop8 ");
    run<1>(64, 1, "VOP3 once  ");  run<1>(64, 8, "VOP3 loop8 ");
    run<1>(256, 1, "VOP3 once  "); run<1>(1024, 1, "VOP3 once  ");
    run<2>(64, 128, "VOP3 small body x128 "); run<3>(64, 128, "VOP3 dependent small ");
    run<2>(2048, 128, "VOP3 small, 8 waves/CU"); run<2>(4096, 128, "VOP3 small, 16 waves/CU");
    return 0;
}
