// rng.hpp -- counter-based uniform RNG for in-kernel episode resets.
// The reference draws reset noise with torch.rand on the sim device (torch_jit_utils.py:215-218, ant.py:257-258),
// whose stream cannot be reproduced inside a fused kernel.  We use a stateless integer hash of
// (seed, global env id, episode number, draw index): bit-exact between the HIP kernels and the CPU oracle
// (oracle/tasks.py: mi_uniform), independent of which envs reset together and of the env->GPU sharding.
#pragma once
#include <cstdint>
#include "engine.hpp"

namespace mi {
MI_HD uint32_t fmix32(uint32_t h) {
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return h;
}
MI_HD float uniform01(uint32_t seed, uint32_t env, uint32_t episode, uint32_t k) {
    uint32_t h = fmix32(seed ^ (env * 0x9E3779B1u));
    h = fmix32(h ^ (episode * 0x85EBCA77u));
    h = fmix32(h ^ (k * 0xC2B2AE3Du));
    return (float)(h >> 8) * (1.0f / 16777216.0f);
}

// Compensated accumulation (Knuth two-sum) for the cumulative job statistics: `hi + lo` carries ~48 bits, so that a sum that has grown past 2^24
// keeps taking small increments (a float count stops at 16,777,216; window means are differences of two snapshots -- parallel.py).
MI_HD void two_sum_acc(float& hi, float& lo, const float x) {   // (no products: FP contraction cannot touch it; the builds do not reassociate)
    const float t = hi + x, bp = t - hi;
    lo += (hi - (t - bp)) + (x - bp);
    hi = t;
}

// ---- in-kernel observation / action noise of the domain randomisation (reference vec_task.py:650-718: `noise_lambda` closures of
// torch.randn_like / rand_like ops run on the returned buffers).  Here the noise of element k of env e is a pure function of
// (seed, e, step, k): nothing to store, identical on every launch shape, reproducible by the CPU twin (oracle/tasks.py: mi_noise).
//   value = op(x, corr + white),   corr  = z_c * b_corr + a_corr            (gaussian)   z_c ~ N(0,1) drawn ONCE per (env, k)
//                                          z_c * (b_corr - a_corr) + a_corr  (uniform; the reference scales a normal draw here too)
//                                  white = N(a, b) or U(a, b), fresh every step
struct NoiseParams {      // mirrors MiNoiseParams (include/mi_engine.h)
    int dist;             // 0 off, 1 gaussian, 2 uniform
    int op;               // 0 additive, 1 scaling
    float a, b;           // gaussian: mean, std; uniform: low, high -- already blended by the schedule on the host
    float a_corr, b_corr; // the same for the per-env correlated part
    unsigned epoch;       // stream of the correlated draws: bumped on every global refresh of the randomisation (the reference re-draws `corr` then, vec_task.py:690,716)
};
MI_HD float gauss01(uint32_t seed, uint32_t env, uint32_t ctr, uint32_t k) {
    const float u1 = fmaxf(uniform01(seed, env, ctr, 2u * k), 5.9604645e-8f), u2 = uniform01(seed, env, ctr, 2u * k + 1u);
    return sqrtf(-2.f * logf(u1)) * cosf(6.283185307179586f * u2);
}
// stream: 0 observations, 1 actions (keeps the two tensors' draws apart)
MI_HD float apply_noise(const NoiseParams& p, uint32_t seed, uint32_t genv, uint32_t step, uint32_t stream, uint32_t k, float x) {
    const float zc = gauss01(seed, genv, 0xC0000000u ^ (p.epoch * 0x9E3779B1u + stream), k);
    const uint32_t ctr = 0x80000000u | (step * 2u + stream);
    float corr, white;
    if (p.dist == 1) { corr = zc * p.b_corr + p.a_corr; white = gauss01(seed, genv, ctr, k) * p.b + p.a; }
    else { corr = zc * (p.b_corr - p.a_corr) + p.a_corr; white = uniform01(seed, genv, ctr, k) * (p.b - p.a) + p.a; }
    const float nz = corr + white;
    return p.op == 0 ? x + nz : x * nz;
}
}  // namespace mi
