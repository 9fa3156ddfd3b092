// bufops.h -- raw buffer addressing helpers shared by the register-resident streaming kernels (gfx950).
#pragma once
#include <hip/hip_runtime.h>

// Everything goes through raw buffer instructions: one wave-uniform descriptor (SGPRs) + a 32-bit per-lane byte offset + a
// wave-uniform SGPR offset.  Lanes / channels outside the band get an offset beyond num_records: their loads return 0 and their
// stores are dropped by the hardware range check, so the streaming loops carry no predicates and one VGPR of address state.
typedef unsigned int bufops_u32;
typedef __amdgpu_buffer_rsrc_t rsrc_t;
constexpr int AUX_SC1 = 16, AUX_NT = 2;
constexpr bufops_u32 OOB = 0x80000000u;                                    // >= any num_records used here (per-image extents < 2 GB)

__device__ __forceinline__ rsrc_t make_rsrc(const void* base, bufops_u32 bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
