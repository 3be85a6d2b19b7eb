// kernels_mw_ant.hip -- multi-wave sub-step of the Ant (one leg per wave), gfx950.
// (A/B only: MI_EXTRA_HIPCC_FLAGS=-DMI_MW_HAS8=1 also builds the 8-env / two-workgroups-per-CU form of the one-launch kernel -- it needs 414
//  registers per lane, at two waves per SIMD it spills 384 VGPRs and runs 1.65x slower: profiles/r6c_ant_mw8_two_workgroups_per_cu_ab.txt)
#include "mw_kernels.hpp"
#include "gen/model_ant.h"

namespace mi {
template hipError_t launch_substeps_mw<ModelAnt, PlaneGround>(const View&, const SimParams&, const ActParams&, const float*, int, int, int, hipStream_t,
                                                              const PlaneGround&, int, const MwCmdNormTail*);
template hipError_t launch_substeps_mw_post<ModelAnt, false>(const View&, const SimParams&, const ActParams&, const float*, int, int, int, hipStream_t,
                                                            const LocoParams&);
}  // namespace mi

#if defined(MI_TIMING)
extern "C" int mi_debug_set_tstamp_mw(void* device_buffer) {   // debug builds only (tools/debug/mw_phases.py)
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(mi::g_mi_tstamp_mw), &device_buffer, sizeof(void*));
}
#endif
