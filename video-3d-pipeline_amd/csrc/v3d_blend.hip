// v3d_blend.hip -- the "hybrid" step of reference depth.py:344-374: blend the stereo disparity with a monocular
// depth map (the reference gets it from DPT; here it is any float32 map the host supplies):
//   :353-354  mono = cv2.resize(mono, (W, H))            INTER_LINEAR on a float32 image
//   :359-360  mono_n = (mono - min) / (max - min) * 64   float32; the blend is skipped when max == min
//   :363      combined = 0.7 * disparity + 0.3 * mono_n  disparity = compute() / 16 (invalid = -1.0)
//   :374      combined[combined <= 0] = 0
// Two streaming launches per batch: (1) resize into the output buffer + per-frame min/max (wave shuffle reduction,
// one atomic pair per wave, order-preserving uint encoding); (2) normalise + blend + clamp in place.
// Every float operation is a single correctly rounded f32 op in the reference's order (no contraction), so the result is
// bit-identical to the NumPy expression on the same resized map.  HBM-bound: 2 B + 4 B in, 4 B written twice per pixel.
#include "v3d_common.h"

__device__ __forceinline__ unsigned bl_f2ord(float f) { unsigned u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ __forceinline__ float bl_ord2f(unsigned u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u); }

__global__ void k_blend_mm_init(unsigned* mm, int n)
{
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i < n) { mm[2 * i] = 0xFFFFFFFFu; mm[2 * i + 1] = 0u; }
}

#define BL_ROWS 8
// cv2.resize INTER_LINEAR, CV_32F: horizontal taps first, then vertical; x taps outside the row snap to the border
// pixel with weight 0, y rows are clamped with the weights kept
__global__ __launch_bounds__(256) void k_mono_resize_minmax(const float* __restrict__ mono, int mw, int mh, size_t mono_stride,
                                                            int W, int H, double scx, double scy, int identity,
                                                            float* __restrict__ out, unsigned* __restrict__ mm)
{
    const int f = blockIdx.z;
    const float* src = mono + (size_t)f * mono_stride;
    float* dst = out + (size_t)f * W * H;
    const int x = blockIdx.x * 256 + threadIdx.x;
    const int y0 = blockIdx.y * BL_ROWS;
    unsigned lo = 0xFFFFFFFFu, hi = 0u;
    if (x < W) {
        float fx = (float)((x + 0.5) * scx - 0.5);
        int sx = (int)floorf(fx);
        fx = __fsub_rn(fx, (float)sx);
        if (sx < 0) { fx = 0.f; sx = 0; }
        if (sx >= mw - 1) { fx = 0.f; sx = mw - 1; }
        const int sx1 = sx + 1 < mw ? sx + 1 : sx;
        const float a0 = __fsub_rn(1.f, fx), a1 = fx;
        for (int y = y0; y < min(y0 + BL_ROWS, H); y++) {
            float v;
            if (identity) v = src[(size_t)y * mw + x];
            else {
                float fy = (float)((y + 0.5) * scy - 0.5);
                const int sy = (int)floorf(fy);
                fy = __fsub_rn(fy, (float)sy);
                const int ya = min(max(sy, 0), mh - 1), yb = min(max(sy + 1, 0), mh - 1);
                const float b0 = __fsub_rn(1.f, fy), b1 = fy;
                const float* ra = src + (size_t)ya * mw;
                const float* rb = src + (size_t)yb * mw;
                const float r0 = __fadd_rn(__fmul_rn(ra[sx], a0), __fmul_rn(ra[sx1], a1));
                const float r1 = __fadd_rn(__fmul_rn(rb[sx], a0), __fmul_rn(rb[sx1], a1));
                v = __fadd_rn(__fmul_rn(r0, b0), __fmul_rn(r1, b1));
            }
            dst[(size_t)y * W + x] = v;
            const unsigned o = bl_f2ord(v);
            lo = min(lo, o); hi = max(hi, o);
        }
    }
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) { lo = min(lo, (unsigned)__shfl_xor((int)lo, s)); hi = max(hi, (unsigned)__shfl_xor((int)hi, s)); }
    if ((threadIdx.x & 63) == 0 && lo <= hi) { atomicMin(mm + 2 * f, lo); atomicMax(mm + 2 * f + 1, hi); }
}

__global__ __launch_bounds__(256) void k_mono_blend(const int16_t* __restrict__ disp16, size_t npx, const unsigned* __restrict__ mm,
                                                    float ws, float wm, float* __restrict__ out)
{
    const int f = blockIdx.y;
    const float mn = bl_ord2f(mm[2 * f]), mx = bl_ord2f(mm[2 * f + 1]);
    const bool flat = !(mx > mn);
    const float range = __fsub_rn(mx, mn);
    const int16_t* d16 = disp16 + (size_t)f * npx;
    float* o = out + (size_t)f * npx;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < npx; i += (size_t)gridDim.x * 256) {
        const float d = __fdiv_rn((float)d16[i], 16.0f);
        float c = d;
        if (!flat) {
            const float e = __fmul_rn(__fdiv_rn(__fsub_rn(o[i], mn), range), 64.0f);
            c = __fadd_rn(__fmul_rn(ws, d), __fmul_rn(wm, e));
        }
        o[i] = c <= 0.f ? 0.f : c;
    }
}

extern "C" size_t v3d_mono_blend_ws_bytes(int n) { return n > 0 ? (size_t)n * 2 * sizeof(unsigned) : 0; }

extern "C" int v3d_mono_blend_batch(const int16_t* disp16, int n, int W, int H, const float* mono, int mw, int mh,
                                    size_t mono_stride, float w_stereo, float w_mono, float* depth_out, void* ws, void* stream)
{
    if (!disp16 || !mono || !depth_out || !ws) { v3d_set_error("null pointer"); return V3D_ERR_ARG; }
    if (n < 1 || W < 1 || H < 1 || mw < 1 || mh < 1) { v3d_set_error("bad geometry"); return V3D_ERR_ARG; }
    if (n > 65535) { v3d_set_error("batch too large"); return V3D_ERR_ARG; }
    hipStream_t st = (hipStream_t)stream;
    unsigned* mm = reinterpret_cast<unsigned*>(ws);
    const double scx = 1.0 / ((double)W / mw), scy = 1.0 / ((double)H / mh);       // cv::resize: scale = 1 / inv_scale
    const int identity = (mw == W && mh == H) ? 1 : 0;                             // depth.py:352: same size is not resized
    hipLaunchKernelGGL(k_blend_mm_init, dim3(v3d_cdiv(n, 64)), dim3(64), 0, st, mm, n);
    hipLaunchKernelGGL(k_mono_resize_minmax, dim3(v3d_cdiv(W, 256), v3d_cdiv(H, BL_ROWS), n), dim3(256), 0, st,
                       mono, mw, mh, mono_stride, W, H, scx, scy, identity, depth_out, mm);
    const size_t npx = (size_t)W * H;
    const int blocks = (int)((npx + 255) / 256 < 1024 ? (npx + 255) / 256 : 1024);
    hipLaunchKernelGGL(k_mono_blend, dim3(blocks, n), dim3(256), 0, st, disp16, npx, mm, w_stereo, w_mono, depth_out);
    V3D_LAUNCH_CHECK();
    return V3D_OK;
}

extern "C" int v3d_mono_blend(const int16_t* disp16, int W, int H, const float* mono, int mw, int mh,
                              float w_stereo, float w_mono, float* depth_out, void* ws, void* stream)
{
    return v3d_mono_blend_batch(disp16, 1, W, H, mono, mw, mh, 0, w_stereo, w_mono, depth_out, ws, stream);
}
