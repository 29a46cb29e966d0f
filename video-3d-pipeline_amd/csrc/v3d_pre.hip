// v3d_pre.hip -- the per-frame steps either side of the matcher:
//   depth.py:250-268 split_sbs_frame (cv2.resize INTER_LANCZOS4 horizontal x2 "unsqueeze"),
//   depth.py:274-275 + 337-338 cvtColor (BGR->RGB->GRAY), depth.py:341/374 (/16, clamp),
//   depth.py:397-406 save_depth_map (min-max -> uint16).
// All pure streaming kernels: one read of the input, one write of the output.
#include "v3d_common.h"
#include <math.h>

struct LanczosTaps { short t[2][8]; };   // [0]: even output columns (fx = 0.75), [1]: odd (fx = 0.25)

// host: the 8 int16 taps cv::resize builds for fractional offset fx (coefficients scaled by 2^11,
// rounded half-to-even like cvRound)
static void lanczos4_taps_host(float x, short* taps)
{
    static const double s45 = 0.70710678118654752440084436210485;
    static const double cs[8][2] = { {1, 0}, {-s45, -s45}, {0, 1}, {s45, -s45}, {-1, 0}, {s45, s45}, {0, -1}, {-s45, s45} };
    const double PI = 3.1415926535897932384626433832795;
    float c[8], sum = 0.f;
    const double y0 = -(x + 3) * PI * 0.25, s0 = sin(y0), c0 = cos(y0);
    for (int i = 0; i < 8; i++) {
        const float yi = (x + 3 - i);
        if (fabsf(yi) >= 1e-6f) { const double y = -yi * PI * 0.25; c[i] = (float)((cs[i][0] * s0 + cs[i][1] * c0) / (y * y)); }
        else c[i] = 1e30f;
        sum += c[i];
    }
    sum = 1.f / sum;
    for (int i = 0; i < 8; i++) {
        long r = lrintf(c[i] * sum * 2048.f);
        taps[i] = (short)(r < -32768 ? -32768 : (r > 32767 ? 32767 : r));
    }
}

__device__ __forceinline__ int gray_of(int b, int g, int r) { return (r * 9798 + g * 19235 + b * 3735 + (1 << 14)) >> 15; }

// one thread = one output pixel of one eye; a block covers 256 output pixels of one row.  The source span
// (128 + 8 BGR pixels when unsqueezing, 256 otherwise) is staged through LDS with coalesced dword loads:
// per-tap byte loads at a 3-byte stride straight from HBM ran at 0.3 TB/s.
// GRAY: write luma only; else write the BGR triple.
#ifndef SBS_ROWS
#define SBS_ROWS 8
#endif
//    // rows per workgroup: the per-row work is tiny, one workgroup per row was bound by dispatch
// Unsqueezing, a thread computes the output PAIR (2m, 2m+1): the two 8-tap windows (source pixels m-4..m+3 with the
// frac-0.75 taps, m-3..m+4 with the frac-0.25 taps) overlap in 7 of 8 pixels, so 27 LDS bytes serve both instead of
// 48; a block covers 512 output pixels of one eye row.  Without unsqueeze a thread copies one pixel (256 per block).
template <bool GRAY>
__global__ __launch_bounds__(256) void k_split_sbs(const uint8_t* __restrict__ sbs, int W, int H, int pitch, int unsqueeze,
                                                   LanczosTaps taps, uint8_t* __restrict__ outL, uint8_t* __restrict__ outR,
                                                   size_t in_stride)
{
    __shared__ __attribute__((aligned(4))) uint8_t sRow[2][(256 + 8) * 3 + 16];
    const int hw = W >> 1, ow = unsqueeze ? W : hw;
    const int t = threadIdx.x;
    const int xb = blockIdx.x * (unsqueeze ? 512 : 256), eye = blockIdx.z & 1, f = blockIdx.z >> 1;
    const int ya = blockIdx.y * SBS_ROWS, yb = min(ya + SBS_ROWS, H);
    sbs += (size_t)f * in_stride;
    const size_t ostride = (size_t)ow * H * (GRAY ? 1 : 3);
    outL += f * ostride; outR += f * ostride;
    uint8_t* out = eye ? outR : outL;

    // source pixels [s0, s0 + ns) of the half row are needed by this block
    const int s0 = unsqueeze ? (xb >> 1) - 4 : xb;
    const int ns = unsqueeze ? 256 + 8 : 256;
    const bool interior = s0 >= 0 && (s0 + ns < hw || (eye == 0 && s0 + ns <= hw));
    int tp0[8], tp1[8];
#pragma unroll
    for (int k = 0; k < 8; k++) { tp0[k] = taps.t[0][k]; tp1[k] = taps.t[1][k]; }

    // Staging plan of this thread, the same for every row of the band: interior blocks copy ONE aligned dword of the
    // span (the <= 3 bytes of over-read stay inside the image row); edge blocks copy up to four bytes from clamped
    // pixels (replicated border).  The plan's loads for row y+1 are issued before row y is computed: a row is ~140
    // instructions, a load ~1.5 us -- fetched at the top of its own row, every row waited for its bytes.
    // (Dword copies need a row-invariant alignment: pitch a multiple of 4.  A half row that starts at the row's first
    //  byte on an unaligned address would be read from up to 3 bytes BEFORE the row -- before the caller's buffer for
    //  row 0 of frame 0: such blocks take the byte plan.)
    const uint8_t* col0 = sbs + (size_t)eye * hw * 3;               // this eye's half row starts here in every row
    const uintptr_t a = reinterpret_cast<uintptr_t>(col0 + (size_t)max(s0, 0) * 3);
    const bool head_unaligned = (eye | s0) == 0 && (reinterpret_cast<uintptr_t>(col0) & 3) != 0;
    const bool fast = interior && !head_unaligned && (pitch & 3) == 0;
    const int soff = fast ? (int)(a & 3) : 0;                       // byte offset of pixel s0 inside sR
    const int nd = (soff + ns * 3 + 3) >> 2;
    int boff[4];                                                    // byte plan: source offsets inside the half row
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int i = min(t + 256 * k, ns * 3 - 1), px = i / 3, c = i - px * 3;
        boff[k] = min(max(s0 + px, 0), hw - 1) * 3 + c;
    }
    uint32_t pre[4] = { 0u, 0u, 0u, 0u };
    auto load_row = [&](int y) {
        const uint8_t* row = col0 + (size_t)y * pitch;
        if (fast) {
            if (t < nd) pre[0] = reinterpret_cast<const uint32_t*>((reinterpret_cast<uintptr_t>(row + (size_t)s0 * 3)) & ~(uintptr_t)3)[t];
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++) pre[k] = row[boff[k]];
        }
    };
    load_row(ya);
    for (int y = ya; y < yb; y++) {
        uint8_t* sR = sRow[y & 1];                                  // double-buffered: one barrier per row
        if (fast) { if (t < nd) reinterpret_cast<uint32_t*>(sR)[t] = pre[0]; }
        else {
#pragma unroll
            for (int k = 0; k < 4; k++) if (t + 256 * k < ns * 3) sR[t + 256 * k] = (uint8_t)pre[k];
        }
        __syncthreads();
        if (y + 1 < yb) load_row(y + 1);                            // flies while this row is computed
        if (unsqueeze) {
            // source phase: fx = (x + 0.5) * 0.5 - 0.5 -> even x = 2m: sx = m - 1, frac 0.75; odd x = 2m + 1: sx = m, frac 0.25
            const int x = xb + 2 * t;                               // even member of the pair; m = x / 2 = s0 + 4 + t
            if (x >= ow) continue;
            const uint8_t* p0 = sR + soff + t * 3;                  // pixel m - 4
            int acc[2][3] = { { 0, 0, 0 }, { 0, 0, 0 } };
#pragma unroll
            for (int k = 0; k < 9; k++) {
                const int b = p0[3 * k], g = p0[3 * k + 1], r = p0[3 * k + 2];
                if (k < 8) { acc[0][0] += b * tp0[k]; acc[0][1] += g * tp0[k]; acc[0][2] += r * tp0[k]; }
                if (k > 0) { acc[1][0] += b * tp1[k - 1]; acc[1][1] += g * tp1[k - 1]; acc[1][2] += r * tp1[k - 1]; }
            }
#pragma unroll
            for (int n = 0; n < 2; n++) {
                if (x + n >= ow) break;
                // vertical pass is the identity tap (2048): (a * 2048 + 2^21) >> 22 == (a + 2^10) >> 11 exactly (floor
                // of the same rational), so the descale stays in 32 bits; saturate to u8
                const int b = min(max((acc[n][0] + 1024) >> 11, 0), 255);
                const int g = min(max((acc[n][1] + 1024) >> 11, 0), 255);
                const int r = min(max((acc[n][2] + 1024) >> 11, 0), 255);
                if (GRAY) out[(size_t)y * ow + x + n] = (uint8_t)gray_of(b, g, r);
                else { uint8_t* o = out + ((size_t)y * ow + x + n) * 3; o[0] = (uint8_t)b; o[1] = (uint8_t)g; o[2] = (uint8_t)r; }
            }
        } else {
            const int x = xb + t;
            if (x >= ow) continue;
            const uint8_t* p = sR + soff + t * 3;
            const int b = p[0], g = p[1], r = p[2];
            if (GRAY) out[(size_t)y * ow + x] = (uint8_t)gray_of(b, g, r);
            else { uint8_t* o = out + ((size_t)y * ow + x) * 3; o[0] = (uint8_t)b; o[1] = (uint8_t)g; o[2] = (uint8_t)r; }
        }
    }
}

static int split_common(const uint8_t* sbs, int W, int H, int pitch, int unsqueeze, uint8_t* L, uint8_t* R, bool gray, int n, size_t in_stride, hipStream_t st)
{
    if (n < 1) { v3d_set_error("bad batch"); return V3D_ERR_ARG; }
    if (!sbs || !L || !R) { v3d_set_error("null pointer"); return V3D_ERR_ARG; }
    if (W % 2 != 0) { v3d_set_error("SBS frame width must be even"); return V3D_ERR_ARG; }
    if (W < 2 || H < 1 || pitch < W * 3) { v3d_set_error("bad SBS geometry %dx%d pitch %d", W, H, pitch); return V3D_ERR_ARG; }
    LanczosTaps taps;
    lanczos4_taps_host(0.75f, taps.t[0]);
    lanczos4_taps_host(0.25f, taps.t[1]);
    const int ow = unsqueeze ? W : W / 2;
    if (gray) hipLaunchKernelGGL(k_split_sbs<true>, dim3(v3d_cdiv(ow, unsqueeze ? 512 : 256), v3d_cdiv(H, SBS_ROWS), 2 * n), dim3(256), 0, st, sbs, W, H, pitch, unsqueeze, taps, L, R, in_stride);
    else hipLaunchKernelGGL(k_split_sbs<false>, dim3(v3d_cdiv(ow, unsqueeze ? 512 : 256), v3d_cdiv(H, SBS_ROWS), 2 * n), dim3(256), 0, st, sbs, W, H, pitch, unsqueeze, taps, L, R, in_stride);
    V3D_LAUNCH_CHECK();
    return V3D_OK;
}

extern "C" int v3d_sbs_to_gray(const uint8_t* sbs, int W, int H, int pitch, int unsqueeze, uint8_t* L, uint8_t* R, void* stream)
{
    return split_common(sbs, W, H, pitch, unsqueeze, L, R, true, 1, 0, (hipStream_t)stream);
}
extern "C" int v3d_split_sbs(const uint8_t* sbs, int W, int H, int pitch, int unsqueeze, uint8_t* L, uint8_t* R, void* stream)
{
    return split_common(sbs, W, H, pitch, unsqueeze, L, R, false, 1, 0, (hipStream_t)stream);
}
// n frames: frame f at sbs + f*frame_stride (bytes); outputs packed [n][H][outW]
extern "C" int v3d_sbs_to_gray_batch(const uint8_t* sbs, int n, int W, int H, int pitch, size_t frame_stride, int unsqueeze, uint8_t* L, uint8_t* R, void* stream)
{
    return split_common(sbs, W, H, pitch, unsqueeze, L, R, true, n, frame_stride, (hipStream_t)stream);
}

__global__ __launch_bounds__(256) void k_bgr_to_gray(const uint8_t* __restrict__ bgr, size_t n, uint8_t* __restrict__ gray)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        gray[i] = (uint8_t)gray_of(bgr[3 * i], bgr[3 * i + 1], bgr[3 * i + 2]);
}
extern "C" int v3d_bgr_to_gray(const uint8_t* bgr, size_t n, uint8_t* gray, void* stream)
{
    if (!bgr || !gray) { v3d_set_error("null pointer"); return V3D_ERR_ARG; }
    if (n == 0) return V3D_OK;
    const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(k_bgr_to_gray, dim3(blocks), dim3(256), 0, (hipStream_t)stream, bgr, n, gray);
    V3D_LAUNCH_CHECK();
    return V3D_OK;
}

__global__ __launch_bounds__(256) void k_disp_to_depth(const int16_t* __restrict__ d, size_t n, float* __restrict__ out)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float f = (float)d[i] / 16.0f;
        out[i] = f <= 0.f ? 0.f : f;
    }
}
extern "C" int v3d_disp_to_depth(const int16_t* disp16, size_t n, float* out, void* stream)
{
    if (!disp16 || !out) { v3d_set_error("null pointer"); return V3D_ERR_ARG; }
    if (n == 0) return V3D_OK;
    const int blocks = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    hipLaunchKernelGGL(k_disp_to_depth, dim3(blocks), dim3(256), 0, (hipStream_t)stream, disp16, n, out);
    V3D_LAUNCH_CHECK();
    return V3D_OK;
}

// ---- save_depth_map normalisation: ((d - min) / (max - min) * 65535).astype(uint16) in float32 ----
__device__ __forceinline__ unsigned f2ord(float f) { unsigned u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ __forceinline__ float ord2f(unsigned u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u); }

__global__ void k_minmax_init(unsigned* mm) { mm[0] = 0xFFFFFFFFu; mm[1] = 0u; }
__global__ __launch_bounds__(256) void k_minmax(const float* __restrict__ d, size_t n, unsigned* mm)
{
    unsigned lo = 0xFFFFFFFFu, hi = 0u;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const unsigned o = f2ord(d[i]);
        lo = min(lo, o); hi = max(hi, o);
    }
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) { lo = min(lo, (unsigned)__shfl_xor((int)lo, s)); hi = max(hi, (unsigned)__shfl_xor((int)hi, s)); }
    if ((threadIdx.x & 63) == 0) { atomicMin(mm, lo); atomicMax(mm + 1, hi); }
}
__global__ __launch_bounds__(256) void k_norm_u16(const float* __restrict__ d, size_t n, const unsigned* __restrict__ mm, uint16_t* __restrict__ out)
{
    const float mn = ord2f(mm[0]), mx = ord2f(mm[1]);
    const bool flat = !(mx > mn);
    const float range = mx - mn;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float v = 0.f;
        if (!flat) {
            v = __fsub_rn(d[i], mn);
            v = __fdiv_rn(v, range);
            v = __fmul_rn(v, 65535.0f);
        }
        out[i] = (uint16_t)v;
    }
}
extern "C" int v3d_depth_to_u16(const float* depth, size_t n, uint16_t* out, float* ws, void* stream)
{
    if (!depth || !out || !ws) { v3d_set_error("null pointer"); return V3D_ERR_ARG; }
    if (n == 0) return V3D_OK;
    hipStream_t st = (hipStream_t)stream;
    unsigned* mm = reinterpret_cast<unsigned*>(ws);
    const int blocks = (int)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
    hipLaunchKernelGGL(k_minmax_init, dim3(1), dim3(1), 0, st, mm);
    hipLaunchKernelGGL(k_minmax, dim3(blocks), dim3(256), 0, st, depth, n, mm);
    hipLaunchKernelGGL(k_norm_u16, dim3(blocks), dim3(256), 0, st, depth, n, mm, out);
    V3D_LAUNCH_CHECK();
    return V3D_OK;
}

// ---- 4K depth -> the 16-bit sample of the PNG sink: round to nearest even (numpy.rint / torch.round), clamp to [0, 65535] ----
__global__ __launch_bounds__(256) void k_round_u16(const float* __restrict__ d, size_t n, uint16_t* __restrict__ out)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float v = rintf(d[i]);
        out[i] = (uint16_t)(v >= 65535.f ? 65535.f : v > 0.f ? v : 0.f);        // NaN -> 0
    }
}
extern "C" int v3d_round_to_u16(const float* depth, size_t n, uint16_t* out, void* stream)
{
    if (!depth || !out) { v3d_set_error("null pointer"); return V3D_ERR_ARG; }
    if (n == 0) return V3D_OK;
    const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(k_round_u16, dim3(blocks), dim3(256), 0, (hipStream_t)stream, depth, n, out);
    V3D_LAUNCH_CHECK();
    return V3D_OK;
}
