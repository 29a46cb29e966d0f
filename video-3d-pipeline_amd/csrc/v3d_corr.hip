// v3d_corr.hip -- CREStereo-style local group correlation lookup on the matrix cores
// (BASELINE.json config 4; SURVEY.md 8a-12 / Appendix B.2 form A).  The reference only *names*
// CREStereo (depth.py:1, CREStereo_model.txt); there is no reference code for this step.
//
//   fr'            = bilinear sample of fr at (x + flow_x, y + flow_y), zeros outside   (k_corr_warp)
//   out[g*9+k,y,x] = (1/64) * sum_{c in group g} fl[y,x,c] * fr'[clamp(y+dy), clamp(x+dx), c]
//
// It is a dense feature-channel contraction, so it runs on MFMA: one wave owns a 16-pixel row tile;
// per group, A = fl tile (16 px x 64 ch), B = the 32 warped positions x0-4 .. x0+27 (two 16-wide
// N tiles), 2 k-steps of v_mfma_f32_16x16x32_bf16 each; the 9 wanted diagonals of the 16x32 product
// are picked out of the accumulators.  Arithmetic intensity is ~4 flop/B: the kernel is HBM/L2
// bound, MFMA only removes the VALU bottleneck.
#include "v3d_common.h"
#include <stdlib.h>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float bf2f(unsigned short b) { return __uint_as_float((unsigned)b << 16); }

// one thread = one pixel x one 8-channel chunk
__global__ __launch_bounds__(256) void k_corr_warp(const unsigned short* __restrict__ fr, const float* __restrict__ flow,
                                                   int C, int h, int w, unsigned short* __restrict__ out)
{
    const int chunks = C >> 3;
    const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (size_t)h * w * chunks) return;
    const int ch = (int)(gid % chunks);
    const size_t pix = gid / chunks;
    const int x = (int)(pix % w), y = (int)(pix / w);
    const float sx = x + flow[pix], sy = y + flow[(size_t)h * w + pix];
    const float x0f = floorf(sx), y0f = floorf(sy);
    const float wx = sx - x0f, wy = sy - y0f;
    const int x0 = (int)x0f, y0 = (int)y0f;
    float acc[8];
#pragma unroll
    for (int k = 0; k < 8; k++) acc[k] = 0.f;
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int xx = x0 + i, yy = y0 + j;
            if (xx < 0 || xx >= w || yy < 0 || yy >= h) continue;
            const float wgt = (i ? wx : 1.f - wx) * (j ? wy : 1.f - wy);
            const uint4 v = *reinterpret_cast<const uint4*>(fr + ((size_t)yy * w + xx) * C + ch * 8);
            const unsigned u[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
            for (int k = 0; k < 4; k++) {
                acc[2 * k] += bf2f((unsigned short)(u[k] & 0xFFFFu)) * wgt;
                acc[2 * k + 1] += bf2f((unsigned short)(u[k] >> 16)) * wgt;
            }
        }
    bf16x8 o;
#pragma unroll
    for (int k = 0; k < 8; k++) o[k] = (__bf16)acc[k];
    *reinterpret_cast<bf16x8*>(out + pix * C + ch * 8) = o;
}

// one wave = 16 consecutive pixels of one row, all groups; 4 waves per block
template <int PATTERN>
__global__ __launch_bounds__(256) void k_corr(const unsigned short* __restrict__ fl, const unsigned short* __restrict__ frw,
                                              int C, int h, int w, int G, float* __restrict__ out)
{
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    const int xtiles = (w + 15) >> 4;
    const int gw = blockIdx.x * 4 + wib;
    if (gw >= xtiles * h) return;
    const int y = gw / xtiles, x0 = (gw - y * xtiles) << 4;
    const int r16 = lane & 15, q = lane >> 4;                 // MFMA operand row/col and k-quarter
    const size_t hw = (size_t)h * w;
    const float scale = 1.0f / 64.0f;
    constexpr int NROW = PATTERN == 0 ? 1 : 3;

    const int xa = min(x0 + r16, w - 1);                       // A operand: pixel x0 + r16
    for (int g = 0; g < G; g++) {
        bf16x8 a[2];
#pragma unroll
        for (int ks = 0; ks < 2; ks++)
            a[ks] = *reinterpret_cast<const bf16x8*>(fl + ((size_t)y * w + xa) * C + g * 64 + ks * 32 + q * 8);
#pragma unroll
        for (int ry = 0; ry < NROW; ry++) {
            const int dy = PATTERN == 0 ? 0 : ry - 1;
            const int yy = min(max(y + dy, 0), h - 1);
            f32x4 acc[2];
#pragma unroll
            for (int n = 0; n < 2; n++) {
                acc[n] = (f32x4){ 0.f, 0.f, 0.f, 0.f };
                const int xb = min(max(x0 - 4 + n * 16 + r16, 0), w - 1);   // B operand: warped position
#pragma unroll
                for (int ks = 0; ks < 2; ks++) {
                    const bf16x8 b = *reinterpret_cast<const bf16x8*>(frw + ((size_t)yy * w + xb) * C + g * 64 + ks * 32 + q * 8);
                    acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[ks], b, acc[n], 0, 0, 0);
                }
            }
            // accumulator element (reg r) of this lane: pixel i = 4q + r, position column j = r16 of tile n
            //   -> window offset k = j - i + 16 n  (dx = k - 4), wanted when 0 <= k <= 8 (1x9) or 3 <= k <= 5 (3x3)
#pragma unroll
            for (int n = 0; n < 2; n++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int i = 4 * q + r;
                    const int k = r16 - i + 16 * n;
                    const int x = x0 + i;
                    bool want; int plane;
                    if (PATTERN == 0) { want = (k >= 0 && k <= 8); plane = g * 9 + k; }
                    else { want = (k >= 3 && k <= 5); plane = g * 9 + ry * 3 + (k - 3); }
                    if (want && x < w) out[(size_t)plane * hw + (size_t)y * w + x] = acc[n][r] * scale;
                }
        }
    }
}

// gather-GEMM: one wave = 16 consecutive pixels of one row, all groups; 4 waves per block.  The B operand
// (warped right features) is never materialised: every lane owns one warped position of the N tile, turns
// its flow vector into 4 bilinear taps once per tile row, and for every (group, k-step) gathers its 8
// channels from the 4 taps (4 x 16-B loads, L2-resident), blends them in f32 and rounds to bf16 -- exactly
// the value k_corr_warp would have stored.
template <int PATTERN>
__global__ __launch_bounds__(256) void k_corr_gather(const unsigned short* __restrict__ fl, const unsigned short* __restrict__ fr,
                                                     const float* __restrict__ flow, int C, int h, int w, int G,
                                                     float* __restrict__ out)
{
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    const int xtiles = (w + 15) >> 4;
    const int gw = blockIdx.x * 4 + wib;
    if (gw >= xtiles * h) return;
    const int y = gw / xtiles, x0 = (gw - y * xtiles) << 4;
    const int r16 = lane & 15, q = lane >> 4;                 // MFMA operand row/col and k-quarter
    const size_t hw = (size_t)h * w;
    const float scale = 1.0f / 64.0f;
    constexpr int NROW = PATTERN == 0 ? 1 : 3;
    const int xa = min(x0 + r16, w - 1);                       // A operand: pixel x0 + r16

#pragma unroll
    for (int ry = 0; ry < NROW; ry++) {
        const int dy = PATTERN == 0 ? 0 : ry - 1;
        const int yy = min(max(y + dy, 0), h - 1);
        // bilinear taps of my two warped positions (one per N tile)
        const unsigned short* tap[2][4]; float wt[2][4];
#pragma unroll
        for (int n = 0; n < 2; n++) {
            const int xb = min(max(x0 - 4 + n * 16 + r16, 0), w - 1);
            const size_t pix = (size_t)yy * w + xb;
            const float sx = xb + flow[pix], sy = yy + flow[hw + pix];
            const float x0f = floorf(sx), y0f = floorf(sy);
            const float wx = sx - x0f, wy = sy - y0f;
            const int ix = (int)x0f, iy = (int)y0f;
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const int xx = ix + (t & 1), yv = iy + (t >> 1);
                const bool in = xx >= 0 && xx < w && yv >= 0 && yv < h;
                wt[n][t] = in ? ((t & 1) ? wx : 1.f - wx) * ((t >> 1) ? wy : 1.f - wy) : 0.f;     // zeros outside
                tap[n][t] = fr + ((size_t)min(max(yv, 0), h - 1) * w + min(max(xx, 0), w - 1)) * C;
            }
        }
        for (int g = 0; g < G; g++) {
            bf16x8 a[2];
#pragma unroll
            for (int ks = 0; ks < 2; ks++)
                a[ks] = *reinterpret_cast<const bf16x8*>(fl + ((size_t)y * w + xa) * C + g * 64 + ks * 32 + q * 8);
            f32x4 acc[2];
#pragma unroll
            for (int n = 0; n < 2; n++) {
                acc[n] = (f32x4){ 0.f, 0.f, 0.f, 0.f };
#pragma unroll
                for (int ks = 0; ks < 2; ks++) {
                    const int co = g * 64 + ks * 32 + q * 8;
                    float bl[8];
#pragma unroll
                    for (int k = 0; k < 8; k++) bl[k] = 0.f;
#pragma unroll
                    for (int t = 0; t < 4; t++) {              // same tap order as the oracle: (y0,x0) (y0,x1) (y1,x0) (y1,x1)
                        const uint4 v = *reinterpret_cast<const uint4*>(tap[n][t] + co);
                        const unsigned u[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            bl[2 * k] += bf2f((unsigned short)(u[k] & 0xFFFFu)) * wt[n][t];
                            bl[2 * k + 1] += bf2f((unsigned short)(u[k] >> 16)) * wt[n][t];
                        }
                    }
                    bf16x8 bq;
#pragma unroll
                    for (int k = 0; k < 8; k++) bq[k] = (__bf16)bl[k];
                    acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[ks], bq, acc[n], 0, 0, 0);
                }
            }
            // accumulator element (reg r) of this lane: pixel i = 4q + r, position column j = r16 of tile n
            //   -> window offset k = j - i + 16 n  (dx = k - 4), wanted when 0 <= k <= 8 (1x9) or 3 <= k <= 5 (3x3)
#pragma unroll
            for (int n = 0; n < 2; n++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int i = 4 * q + r;
                    const int k = r16 - i + 16 * n;
                    const int x = x0 + i;
                    bool want; int plane;
                    if (PATTERN == 0) { want = (k >= 0 && k <= 8); plane = g * 9 + k; }
                    else { want = (k >= 3 && k <= 5); plane = g * 9 + ry * 3 + (k - 3); }
                    if (want && x < w) out[(size_t)plane * hw + (size_t)y * w + x] = acc[n][r] * scale;
                }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Gather-GEMM through LDS (1x9 pattern; round 2).  A block owns 64 consecutive pixels of one row (4 waves x 16-pixel
// MFMA tiles).  Phase 1: the block warps the 72 right-feature positions x0-4 .. x0+67 its four tiles read -- every
// (position, 8-channel chunk) is blended ONCE by one thread, exactly as k_corr_warp blends it, and lands in LDS as
// bf16 (the warped features never touch HBM: the two-kernel form writes and re-reads 2 x 66 MB of them per lookup;
// the register-only gather-GEMM k_corr_gather blends every position twice, once per neighbouring tile).  Phase 2: each
// wave runs k_corr's MFMA loop with its B operands read from LDS, and hands the 9 wanted diagonals of a group to an
// LDS slab so that a plane's 16 pixels leave as one 64-byte segment instead of 16 scattered dwords.
// LDS rows are padded to 528 B: the 16 positions of a B tile start 4 banks apart.
// ------------------------------------------------------------------------------------------------
#define CF_PX 64
#define CF_NPOS (CF_PX + 8)
#define CF_PITCH 264                                    // bf16 elements per LDS row for C = 256 (+ 8 pad); generic: C + 8
__global__ __launch_bounds__(256) void k_corr_fused0(const unsigned short* __restrict__ fl, const unsigned short* __restrict__ fr,
                                                     const float* __restrict__ flow, int C, int h, int w, int G,
                                                     float* __restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int pitch = C + 8;                                    // elements
    unsigned short* sB = reinterpret_cast<unsigned short*>(smem);                      // [CF_NPOS][pitch]
    float* sO = reinterpret_cast<float*>(smem + (size_t)CF_NPOS * pitch * 2);          // [4 waves][9][16]
    const int tid = threadIdx.x, lane = tid & 63, wib = tid >> 6;
    const int xblocks = (w + CF_PX - 1) / CF_PX;
    const int y = blockIdx.x / xblocks, x0 = (blockIdx.x - y * xblocks) * CF_PX;
    const size_t hw = (size_t)h * w;
    const int chunks = C >> 3;

    // ---- phase 1: warp 72 positions x all channels into LDS (k_corr_warp's arithmetic, tap order and rounding) ----
    // A thread owns one 8-channel chunk of every `pstep`-th position.  Its flow vectors are fetched together and the four
    // taps of three positions at a time (12 independent 16-byte loads in flight): issued one position after the other, the
    // dependent flow -> tap round trips alone took ~27 us per block.
    {
        const int pstep = 256 / chunks;                          // positions covered per pass (8 for C = 256)
        const int ch = tid % chunks, p0 = tid / chunks;
        constexpr int MAXIT = 9;                                 // ceil(72 / 8); larger for narrower features: handled by the outer loop
        for (int pbase = 0; pbase < CF_NPOS; pbase += MAXIT * pstep) {
            float fx[MAXIT], fy[MAXIT];
#pragma unroll
            for (int i = 0; i < MAXIT; i++) {
                const int p = pbase + p0 + i * pstep;
                const int xb = min(max(x0 - 4 + min(p, CF_NPOS - 1), 0), w - 1);   // window positions are clamped (replicate), like k_corr
                const size_t pix = (size_t)y * w + xb;
                fx[i] = xb + flow[pix]; fy[i] = y + flow[hw + pix];
            }
#pragma unroll
            for (int i3 = 0; i3 < MAXIT; i3 += 3) {
                uint4 tv[3][4]; float wt[3][4];
#pragma unroll
                for (int ii = 0; ii < 3; ii++) {
                    const int i = i3 + ii;
                    const float x0f = floorf(fx[i]), y0f = floorf(fy[i]);
                    const float wx = fx[i] - x0f, wy = fy[i] - y0f;
                    const int ix = (int)x0f, iy = (int)y0f;
#pragma unroll
                    for (int t = 0; t < 4; t++) {                // tap order of k_corr_warp: (y0,x0) (y0,x1) (y1,x0) (y1,x1)
                        const int xx = ix + (t & 1), yy = iy + (t >> 1);
                        const bool in = xx >= 0 && xx < w && yy >= 0 && yy < h;
                        wt[ii][t] = ((t & 1) ? wx : 1.f - wx) * ((t >> 1) ? wy : 1.f - wy);
                        tv[ii][t] = make_uint4(0u, 0u, 0u, 0u);
                        if (in) tv[ii][t] = *reinterpret_cast<const uint4*>(fr + ((size_t)yy * w + xx) * C + ch * 8);
                        else wt[ii][t] = 0.f;                    // (skipped taps add nothing in k_corr_warp; +0 * 0 here)
                    }
                }
#pragma unroll
                for (int ii = 0; ii < 3; ii++) {
                    const int p = pbase + p0 + (i3 + ii) * pstep;
                    float acc[8];
#pragma unroll
                    for (int k = 0; k < 8; k++) acc[k] = 0.f;
#pragma unroll
                    for (int t = 0; t < 4; t++) {
                        const unsigned u[4] = { tv[ii][t].x, tv[ii][t].y, tv[ii][t].z, tv[ii][t].w };
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            acc[2 * k] += bf2f((unsigned short)(u[k] & 0xFFFFu)) * wt[ii][t];
                            acc[2 * k + 1] += bf2f((unsigned short)(u[k] >> 16)) * wt[ii][t];
                        }
                    }
                    bf16x8 o;
#pragma unroll
                    for (int k = 0; k < 8; k++) o[k] = (__bf16)acc[k];
                    if (p < CF_NPOS) *reinterpret_cast<bf16x8*>(sB + (size_t)p * pitch + ch * 8) = o;
                }
            }
        }
    }
    __syncthreads();

    // ---- phase 2: one 16-pixel MFMA tile per wave ----
    const int xw = x0 + 16 * wib;                               // first pixel of this wave's tile
    const int r16 = lane & 15, q = lane >> 4;
    const float scale = 1.0f / 64.0f;
    const int xa = min(xw + r16, w - 1);
    float* myO = sO + wib * 9 * 16;
    for (int g = 0; g < G; g++) {
        bf16x8 a[2];
#pragma unroll
        for (int ks = 0; ks < 2; ks++)
            a[ks] = *reinterpret_cast<const bf16x8*>(fl + ((size_t)y * w + xa) * C + g * 64 + ks * 32 + q * 8);
        f32x4 acc[2];
#pragma unroll
        for (int n = 0; n < 2; n++) {
            acc[n] = (f32x4){ 0.f, 0.f, 0.f, 0.f };
            const int p = min(16 * wib + n * 16 + r16, CF_NPOS - 1);      // position xw - 4 + n*16 + r16; columns past 71 are never wanted
#pragma unroll
            for (int ks = 0; ks < 2; ks++) {
                const bf16x8 b = *reinterpret_cast<const bf16x8*>(sB + (size_t)p * pitch + g * 64 + ks * 32 + q * 8);
                acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[ks], b, acc[n], 0, 0, 0);
            }
        }
        // accumulator element (reg r) of this lane: pixel i = 4q + r, position column j = r16 of tile n -> offset k = j - i + 16 n
#pragma unroll
        for (int n = 0; n < 2; n++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int i = 4 * q + r;
                const int k = r16 - i + 16 * n;
                if (k >= 0 && k <= 8) myO[k * 16 + i] = acc[n][r] * scale;
            }
        // the slab is private to the wave: its LDS accesses complete in order, no barrier needed
#pragma unroll
        for (int t = 0; t < 3; t++) {
            const int e = lane + 64 * t;                        // 144 = 9 planes x 16 pixels
            if (e < 144) {
                const int k = e >> 4, i = e & 15;
                if (xw + i < w) out[(size_t)(g * 9 + k) * hw + (size_t)y * w + xw + i] = myO[e];
            }
        }
    }
}

extern "C" size_t v3d_corr_ws_bytes(int C, int h, int w)
{
    if (C < 1 || h < 1 || w < 1) return 0;
    return (size_t)C * h * w * sizeof(unsigned short);
}

extern "C" int v3d_corr_lookup(const uint16_t* fl, const uint16_t* fr, const float* flow, int C, int h, int w,
                               int G, int pattern, float* out, void* ws, void* stream)
{
    if (!fl || !fr || !flow || !out || !ws) { v3d_set_error("null pointer"); return V3D_ERR_ARG; }
    if (G < 1 || C != 64 * G) { v3d_set_error("need C == 64*G channels (got C=%d, G=%d)", C, G); return V3D_ERR_UNSUPPORTED; }
    if (h < 1 || w < 1) { v3d_set_error("bad geometry"); return V3D_ERR_ARG; }
    if (pattern != 0 && pattern != 1) { v3d_set_error("pattern must be 0 (1x9) or 1 (3x3)"); return V3D_ERR_ARG; }
    hipStream_t st = (hipStream_t)stream;
    const int nwaves = ((w + 15) / 16) * h;
    // measured (270x480x256, MI355X): gather-GEMM through LDS 37 us (1x9; round 2, the default); warp + GEMM 67 us (1x9) / 91 us
    // (3x3); register-only gather-GEMM 76 / 198 us (it blends every position twice and waits for each tap round trip)
    if (g_v3d_opt.corr_fused && pattern == 0 && !g_v3d_opt.corr_gather) {
        const size_t smem = (size_t)CF_NPOS * (C + 8) * 2 + 4 * 9 * 16 * sizeof(float);
        if (smem <= 160 * 1024) {
            V3D_HIP_CHECK(hipFuncSetAttribute((const void*)k_corr_fused0, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
            hipLaunchKernelGGL(k_corr_fused0, dim3(v3d_cdiv(w, CF_PX) * h), dim3(256), smem, st, fl, fr, flow, C, h, w, G, out);
            V3D_LAUNCH_CHECK();
            return V3D_OK;
        }
    }
    if (g_v3d_opt.corr_gather) {
        if (pattern == 0) hipLaunchKernelGGL(k_corr_gather<0>, dim3(v3d_cdiv(nwaves, 4)), dim3(256), 0, st, fl, fr, flow, C, h, w, G, out);
        else hipLaunchKernelGGL(k_corr_gather<1>, dim3(v3d_cdiv(nwaves, 4)), dim3(256), 0, st, fl, fr, flow, C, h, w, G, out);
        V3D_LAUNCH_CHECK();
        return V3D_OK;
    }
    unsigned short* frw = reinterpret_cast<unsigned short*>(ws);
    const size_t nthreads = (size_t)h * w * (C / 8);
    hipLaunchKernelGGL(k_corr_warp, dim3((unsigned)((nthreads + 255) / 256)), dim3(256), 0, st, fr, flow, C, h, w, frw);
    const int waves = ((w + 15) / 16) * h;
    if (pattern == 0) hipLaunchKernelGGL(k_corr<0>, dim3(v3d_cdiv(waves, 4)), dim3(256), 0, st, fl, frw, C, h, w, G, out);
    else hipLaunchKernelGGL(k_corr<1>, dim3(v3d_cdiv(waves, 4)), dim3(256), 0, st, fl, frw, C, h, w, G, out);
    V3D_LAUNCH_CHECK();
    return V3D_OK;
}
