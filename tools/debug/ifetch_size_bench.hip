// GPU microbenchmark (tools/debug): does a lone wave slow down when its straight-line code outgrows the 64 KB instruction cache?
// (8 KB ... 192 KB of v_fma, executed once per launch, 4 launches each).  Answer on MI355X: no -- see DESIGN.md 5 (round-1 measurement, table in git history).
#include <hip/hip_runtime.h>
#include <cstdio>
#define R4(x) x x x x
#define R16(x) R4(R4(x))
#define R64(x) R4(R16(x))
#define R256(x) R16(R16(x))
#define BLK R256(asm volatile("v_fma_f32 %0, %0, %4, %4\n v_fma_f32 %1, %1, %4, %4\n v_fma_f32 %2, %2, %4, %4\n v_fma_f32 %3, %3, %4, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e));)
// NB x 1024 VOP3 instructions = NB x 8 KB of straight-line code, executed once per launch
template <int NB>
__global__ __launch_bounds__(64) void k(float* out, unsigned long long* cyc) {
    float a = threadIdx.x, b = a + 1, c = a + 2, d = a + 3;
    const float e = 1.0001f;
    unsigned long long t0 = __builtin_readcyclecounter();
    if (NB >= 1) { BLK } if (NB >= 2) { BLK } if (NB >= 3) { BLK } if (NB >= 4) { BLK } if (NB >= 5) { BLK } if (NB >= 6) { BLK }
    if (NB >= 7) { BLK } if (NB >= 8) { BLK } if (NB >= 9) { BLK } if (NB >= 10) { BLK } if (NB >= 11) { BLK } if (NB >= 12) { BLK }
    if (NB >= 13) { BLK } if (NB >= 14) { BLK } if (NB >= 15) { BLK } if (NB >= 16) { BLK } if (NB >= 17) { BLK } if (NB >= 18) { BLK }
    if (NB >= 19) { BLK } if (NB >= 20) { BLK } if (NB >= 21) { BLK } if (NB >= 22) { BLK } if (NB >= 23) { BLK } if (NB >= 24) { BLK }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 64 + threadIdx.x] = a + b + c + d;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int NB> void run(int waves) {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, waves * 64 * 4); hipMalloc(&cyc, 8 * waves * 8);
    for (int it = 0; it < 4; ++it) hipLaunchKernelGGL(k<NB>, dim3(waves), dim3(64), 0, 0, out, cyc + it * waves);
    hipDeviceSynchronize();
    static unsigned long long h[8 * 1024]; hipMemcpy(h, cyc, 4 * waves * 8, hipMemcpyDeviceToHost);
    printf("%3d KB, %4d waves:", NB * 8, waves);
    for (int it = 0; it < 4; ++it) { double s = 0; for (int i = 0; i < waves; ++i) s += h[it * waves + i]; printf("  %.2f", s / waves / (1024.0 * NB)); }
    printf("  ticks/instr per launch\n");
    hipFree(out); hipFree(cyc);
}
int main() {
    run<4>(64); run<6>(64); run<7>(64); run<8>(64); run<9>(64); run<10>(64); run<12>(64); run<16>(64); run<24>(64);
    run<8>(256); run<12>(256); run<24>(256);
    return 0;
}
