// v3d_common.h -- shared helpers for libv3d_hip (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/v3d_hip.h"

#define V3D_D 64              // numDisparities this build is specialised for (one wavefront of d)
#define V3D_MAX_COST 32767
#define V3D_INVALID16 (-16)   // (minDisparity - 1) * 16

void v3d_set_error(const char* fmt, ...);

// library-wide tuning switches (v3d_set_option); the library reads no environment variables
struct v3d_lib_options {
    int gf_band1, gf_band2;   // rows per workgroup of the guided sweeps (measured best on 30 x 4K frames)
    int gf_tiled;             // 1: force the LDS-tiled guided kernel for every radius
    int gf_fused;             // 1: single-launch guided filter (a/b rows handed from stage-1 to stage-2 waves through LDS), 0: two sweeps through HBM
    int gf_band;              // rows per workgroup of the fused kernel (default 432); 0 = per launch: fewest rounds x (band + 4r)
    int gf_cols;              // strip width of the fused kernel: 256 (8 waves, two workgroups per CU) or 512 (16 waves, one)
    int gf_int1;              // 1: int16 disparity + exact 2x -> stage 1 of the fused kernel in exact integers (same bits)
    int corr_gather;          // 1: register-only gather-GEMM correlation (bit-identical, blends every position twice, slower)
    int corr_fused;           // 1: gather-GEMM through LDS for the 1x9 pattern (warped features never touch HBM), 0: warp kernel + GEMM kernel
};
extern v3d_lib_options g_v3d_opt;

#define V3D_HIP_CHECK(expr)                                                              \
    do {                                                                                 \
        hipError_t _e = (expr);                                                          \
        if (_e != hipSuccess) {                                                          \
            v3d_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            (void)hipGetLastError();   /* clear the sticky error so the next call starts clean */ \
            return V3D_ERR_HIP;                                                          \
        }                                                                                \
    } while (0)

#define V3D_LAUNCH_CHECK()                                                               \
    do {                                                                                 \
        hipError_t _e = hipGetLastError();                                               \
        if (_e != hipSuccess) {                                                          \
            v3d_set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(_e), __FILE__, __LINE__); \
            return V3D_ERR_HIP;                                                          \
        }                                                                                \
    } while (0)

static inline int v3d_cdiv(int a, int b) { return (a + b - 1) / b; }

#ifdef __HIPCC__
// ---- packed 2 x int16 arithmetic on one VGPR (v_pk_*_i16 / _u16 on gfx950) ----
typedef short v3d_s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short v3d_u16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ v3d_s16x2 as_s(uint32_t a) { return __builtin_bit_cast(v3d_s16x2, a); }
__device__ __forceinline__ v3d_u16x2 as_us(uint32_t a) { return __builtin_bit_cast(v3d_u16x2, a); }
__device__ __forceinline__ uint32_t as_u(v3d_s16x2 a) { return __builtin_bit_cast(uint32_t, a); }
__device__ __forceinline__ uint32_t as_u(v3d_u16x2 a) { return __builtin_bit_cast(uint32_t, a); }

__device__ __forceinline__ uint32_t pk_min(uint32_t a, uint32_t b) { return as_u(__builtin_elementwise_min(as_s(a), as_s(b))); }
__device__ __forceinline__ uint32_t pk_max(uint32_t a, uint32_t b) { return as_u(__builtin_elementwise_max(as_s(a), as_s(b))); }
__device__ __forceinline__ uint32_t pk_add(uint32_t a, uint32_t b) { return as_u((v3d_s16x2)(as_s(a) + as_s(b))); }
__device__ __forceinline__ uint32_t pk_sub(uint32_t a, uint32_t b) { return as_u((v3d_s16x2)(as_s(a) - as_s(b))); }
__device__ __forceinline__ uint32_t pk_add_sat(uint32_t a, uint32_t b) { return as_u(__builtin_elementwise_add_sat(as_s(a), as_s(b))); }
__device__ __forceinline__ uint32_t pk_subu_sat(uint32_t a, uint32_t b) { return as_u(__builtin_elementwise_sub_sat(as_us(a), as_us(b))); }
__device__ __forceinline__ uint32_t pk_minu(uint32_t a, uint32_t b) { return as_u(__builtin_elementwise_min(as_us(a), as_us(b))); }
__device__ __forceinline__ uint32_t pk_shr_u(uint32_t a, int n) { return as_u((v3d_u16x2)(as_us(a) >> (unsigned short)n)); }
__device__ __forceinline__ uint32_t pk_bcast(int v) { return ((uint32_t)v & 0xFFFFu) * 0x00010001u; }

// ({hi,lo} >> sh) & 0xffffffff  (v_alignbit_b32)
__device__ __forceinline__ uint32_t alignbit(uint32_t hi, uint32_t lo, uint32_t sh) { return __builtin_amdgcn_alignbit(hi, lo, sh); }

// ---- DPP lane movement (CDNA4 is a gfx9-family ISA: row_* controls act inside 16-lane rows) ----
#define V3D_DPP_QUAD(a, b, c, d) ((a) | ((b) << 2) | ((c) << 4) | ((d) << 6))
#define V3D_DPP_ROW_SHL(n) (0x100 + (n))   // lane i reads lane i+n
#define V3D_DPP_ROW_SHR(n) (0x110 + (n))   // lane i reads lane i-n
#define V3D_DPP_ROW_MIRROR 0x140
#define V3D_DPP_ROW_HALF_MIRROR 0x141

template <int CTRL>
__device__ __forceinline__ uint32_t dpp_mov(uint32_t old, uint32_t src)
{
    // lanes whose source is outside the row keep `old` (bound_ctrl = 0)
    return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)src, CTRL, 0xF, 0xF, false);
}
// butterfly exchange (every lane has a valid source): bound_ctrl form, which the DPP-combine pass can fold into
// the consuming VOP2 instruction (v_min_u32_dpp ...)
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_xchg(uint32_t src)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)src, CTRL, 0xF, 0xF, true);
}
// streaming accesses: once-read / once-written volume data bypasses cache retention (V3D_NT=0 at build time to disable)
#ifndef V3D_NT
#define V3D_NT 1
#endif
typedef uint32_t v3d_u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t v3d_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint4 ld_stream(const uint4* p)
{
#if V3D_NT
    const v3d_u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const v3d_u32x4*>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
#else
    return *p;
#endif
}
__device__ __forceinline__ uint2 ld_stream(const uint2* p)
{
#if V3D_NT
    const v3d_u32x2 v = __builtin_nontemporal_load(reinterpret_cast<const v3d_u32x2*>(p));
    return make_uint2(v.x, v.y);
#else
    return *p;
#endif
}
__device__ __forceinline__ void st_stream(uint4* p, uint4 v)
{
#if V3D_NT
    v3d_u32x4 t = { v.x, v.y, v.z, v.w };
    __builtin_nontemporal_store(t, reinterpret_cast<v3d_u32x4*>(p));
#else
    *p = v;
#endif
}
__device__ __forceinline__ void st_stream(uint2* p, uint2 v)
{
#if V3D_NT
    v3d_u32x2 t = { v.x, v.y };
    __builtin_nontemporal_store(t, reinterpret_cast<v3d_u32x2*>(p));
#else
    *p = v;
#endif
}

// ---- raw buffer access (range-checked: an offset with bit 31 set is out of range for every buffer this library
// builds, so a lane is switched off by its OFFSET, not by a branch).  Why it matters on CDNA: vmcnt counts loads
// AND stores in issue order, and once a VMEM instruction sits behind a branch the compiler's s_waitcnt model
// must assume it was not issued -- every later wait then degenerates to vmcnt(0) and a prefetch pipeline ends up
// waiting for the store it has just issued.  Unconditional instructions keep the counts exact.
#define V3D_BUF_OOB 0x80000000u
__device__ __forceinline__ __amdgpu_buffer_rsrc_t buf_rsrc(const void* base, uint32_t bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ uint32_t buf_load_u32(__amdgpu_buffer_rsrc_t r, uint32_t off) { return __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0); }
__device__ __forceinline__ void buf_store_stream(__amdgpu_buffer_rsrc_t r, uint32_t off, uint4 v)
{
    v3d_u32x4 t = { v.x, v.y, v.z, v.w };
    __builtin_amdgcn_raw_buffer_store_b128(t, r, off, 0, V3D_NT ? 2 : 0);
}
__device__ __forceinline__ void buf_store_stream(__amdgpu_buffer_rsrc_t r, uint32_t off, uint2 v)
{
    v3d_u32x2 t = { v.x, v.y };
    __builtin_amdgcn_raw_buffer_store_b64(t, r, off, 0, V3D_NT ? 2 : 0);
}

// XCD-aware tile order: workgroups are dealt round-robin to the 8 XCDs (each with its own 4 MB L2).  Map the linear
// workgroup id so that every XCD walks a CONTIGUOUS range of logical tiles: neighbouring tiles, which re-read each
// other's halo, then share an L2.  Speed only -- nothing may depend on the placement.
__device__ __forceinline__ uint32_t xcd_linear(uint32_t lin, uint32_t nb)
{
    const uint32_t x = lin & 7u, q = nb >> 3, r = nb & 7u;       // XCD x owns q (+1 if x < r) consecutive logical tiles
    return x * q + min(x, r) + (lin >> 3);
}
__device__ __forceinline__ void xcd_tile(int& bx, int& by, int& bz)
{
    const uint32_t nb = gridDim.x * gridDim.y * gridDim.z;
    const uint32_t lg = xcd_linear(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), nb);
    bx = (int)(lg % gridDim.x); by = (int)((lg / gridDim.x) % gridDim.y); bz = (int)(lg / (gridDim.x * gridDim.y));
}

#endif
