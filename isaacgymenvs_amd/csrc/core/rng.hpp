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
}  // namespace mi
