// Internal helpers shared by the libptb_hip translation units (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ptb_hip.h"

namespace ptb {

// thread-local text of the last failing HIP call (ptb_last_hip_error)
void set_hip_error(hipError_t e);
void set_error_text(const char* text);   // the text ptb_last_hip_error() returns for PTB_ELAUNCH (non-HIP failures: RCCL)

inline int check_launch() {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_hip_error(e);
        return PTB_ELAUNCH;
    }
    return PTB_OK;
}

// Results that are written once and not read again by the same launch (merged maps, de-augmented tiles, augmented batches,
// gradients): non-temporal 16-byte stores, so that hundreds of MB of output do not displace the lines the kernel still reads from
// L2 / Infinity Cache.  -DPTB_NT_OUT=0 builds the plain-store variant for A/B runs (tools/build_variant.sh).
#ifndef PTB_NT_OUT
#define PTB_NT_OUT 1
#endif
__device__ __forceinline__ void out_store4(float* p, const float4 v) {
#if PTB_NT_OUT
    typedef float ptb_v4f __attribute__((ext_vector_type(4)));
    __builtin_nontemporal_store(ptb_v4f{v.x, v.y, v.z, v.w}, reinterpret_cast<ptb_v4f*>(p));
#else
    *reinterpret_cast<float4*>(p) = v;
#endif
}

// Index of the wave inside its workgroup as a SCALAR: every lane of a wave computes the same threadIdx.x >> 6, but only through
// readfirstlane does the compiler know it, and everything derived from it (grid-stride group index, image / plane offsets, base
// addresses) then runs on the scalar unit and the loads take an SGPR base -- on the loss kernels that was 7 of 25 vector
// instructions per element.
__device__ __forceinline__ int wave_id() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// tunables (ptb_set_tunable)
extern int g_chunk_rows;    // 16 | 32 | 64
extern int g_force_scalar;  // 0 | 1
extern int g_loss_grid_cap;  // workgroups per loss-kernel launch
extern int g_ms_tiled;       // 0 | 1: LDS-staged multiscale kernel
extern int g_ms_tile_w;      // 64 | 128: output tile width of the fused multiscale kernel
extern int g_rank_finish_fused;  // 0 | 1: one-launch finish of a rank's image (ShardedTileMerger, deferred bands)
extern int g_rs_xcd_map;     // 0 | 1: XCD-contiguous tile order in the Lovasz radix scatter
extern int g_lovasz_rankdot;     // 0 | 1: last level of the key-only Lovasz forward without a scatter
extern int g_lovasz_fused_dot;   // 0 | 1: the binning scatter of the Lovasz training path also evaluates the loss
extern int g_nt_grad_stores; // 0 | 1: non-temporal gradient stores in the fused loss backward
extern int g_stats_pk;       // 0 | 1: statistics-only / focal-only instances of the packed streaming loss kernel
extern int g_focal_pk_grid;  // workgroups of the packed-fp32 fused loss forward
extern int g_focal_pk;       // 0 | 1: A/B of the packed-fp32 instance of the fused loss forward
extern int g_fused_pix2;     // 0 | 1: A/B of the fused loss forward with 2 pixels per lane
extern int g_loss_prefetch;   // 0 | 1: the fused loss forward fetches the next pixel group while it computes the current one
extern int g_smf_bwd_stash;  // 0 | 2 | 4: softmax focal backward with the per-class terms kept in registers, pixels per lane
extern int g_band_rows;     // 32 | 64: rows per work item of band plans created from now on (A/B)
extern int g_band_half_pf;  // 0 | 1 | 2: the band plan kernel prefetches the next covering tile (1: half / bf16 sources only, 2: fp32 too)
extern int g_band_chan_loop; // 0 | 1: identity-view band launches, one workgroup per item over all channels
extern int g_band_lds_db;   // 0 | 1: double-buffered LDS tiles in the prefetching band plan instances
extern int g_band_rot_views;  // 0 | 1: odd work items of the band plan kernel issue their view loads starting at view NV / 2 (A/B)
extern int g_band_xcd;      // 0 | 1: XCD-aware workgroup order in the band plan kernel (A/B)
extern int g_ms_strip;       // 0 | 1..64: XCD-aware tile order of the fused multiscale kernel, strip width in tile columns
extern int g_ms_tile_rows;   // 64 | 32: output tile height of the fused multiscale kernel
extern int g_nt_loads;      // 0 | 1: non-temporal streaming loads in the linear view kernels

}  // namespace ptb
