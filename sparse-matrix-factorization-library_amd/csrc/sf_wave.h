// Wave-level (64 lanes) helpers of the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace sf {

// max over the 64 lanes of a wave, result uniform.  DPP reduction: row_shr 1/2/4/8 inside the four rows of 16 lanes, then
// row_bcast:15 and row_bcast:31 carry the row maxima up to lane 63 (gfx9 DPP controls 0x111.., 0x142, 0x143).  Lanes with no
// source keep `old` = 0, the identity of an unsigned max.
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false));
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// lowest lane holding the largest |v| among the lanes with active != 0 (-1 if there is none); *vmax = that |v|.
// |v| of a finite double orders like its bit pattern read as an unsigned integer: two 32-bit reductions (high word, then the
// low word among the lanes that tie on the high word) and a ballot.  Inactive lanes take no part (key below every |v|).
__device__ __forceinline__ int wave_argmax_abs(double v, bool active, double* vmax) {
    const uint32_t hi = (uint32_t)__double2hiint(v) & 0x7fffffffu, lo = (uint32_t)__double2loint(v);
    // +1 so that an active zero (key 1) still beats an inactive lane (key 0); the high word of a finite |v| is < 0x7ff00000
    const uint32_t khi = active ? hi + 1u : 0u;
    const uint32_t mhi = wave_max_u32(khi);
    if (mhi == 0u) { *vmax = 0.0; return -1; }
    const bool tie = khi == mhi;
    const uint32_t mlo = wave_max_u32(tie ? lo : 0u);
    const unsigned long long who = __ballot(tie && lo == mlo);
    *vmax = __hiloint2double((int)(mhi - 1u), (int)mlo);
    return __ffsll((long long)who) - 1;
}

}  // namespace sf
