// v3d_guided.hip -- guided-filter joint upsampling of the 1080p depth to the 4K guide frame.
//
// Stands where reference upscale.py:21-73 (upscale_depth_maps_ffmpeg: ffmpeg `scale` filter)
// stands; re-specified by BASELINE.json / SURVEY.md 8a-11 + Appendix B.1 as He-Sun-Tang guided
// filtering:  p = bilinear(depth_lo), I = guide/255,
//   a = cov(I,p) / (var(I) + eps),  b = mean(p) - a*mean(I),  q = mean(a)*I + mean(b)
// with (2r+1)^2 box means clipped at the image border and divided by the true pixel count.
//
// Two sweeps, each one LDS-tiled kernel (64x16 output tile + r halo), box sums separable with
// register sliding windows.  HBM traffic: sweep 1 reads guide + depth_lo tiles, writes a,b;
// sweep 2 reads a,b tiles + guide, writes q.
// Numerics: all sums, a and b are float64 (gfx950 runs f64 vector math at half the f32 rate and this
// kernel is LDS-bound): next to zero-depth regions q is ~1e-4 while the window holds values ~40, and the
// 1e-3 *relative* parity bar cannot be met there with f32 cancellation in cov/var and in box(b).
#include "v3d_common.h"

// 1/x to full double precision: v_rcp_f64 seed (~26 bits) + two Newton steps (5 instructions instead of the
// ~30 of an IEEE division; the last-ulp difference is far inside the 1e-3 parity bar)
__device__ __forceinline__ double gf_rcp(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    r = fma(r, fma(-x, r, 1.0), r);
    r = fma(r, fma(-x, r, 1.0), r);
    return r;
}

#define GF_TX 64
#define GF_RUN 8      // outputs per thread along x in the horizontal pass
#define GF_RMAX 16
// tile height TY = 4 * RUNY: 16 rows for r <= 8, 8 rows above (keeps the f64 tile inside the 160 KiB LDS)

__device__ __forceinline__ double gf_bilinear(const float* __restrict__ src, int Ws, int Hs, double sx, double sy, int x, int y)
{
    const double fx = (x + 0.5) * sx - 0.5, fy = (y + 0.5) * sy - 0.5;
    const double x0f = floor(fx), y0f = floor(fy);
    const double wx = fx - x0f, wy = fy - y0f;
    const int x0 = (int)x0f, y0 = (int)y0f;
    const int xa = min(max(x0, 0), Ws - 1), xb = min(max(x0 + 1, 0), Ws - 1);
    const int ya = min(max(y0, 0), Hs - 1), yb = min(max(y0 + 1, 0), Hs - 1);
    const double top = (double)src[(size_t)ya * Ws + xa] * (1.0 - wx) + (double)src[(size_t)ya * Ws + xb] * wx;
    const double bot = (double)src[(size_t)yb * Ws + xa] * (1.0 - wx) + (double)src[(size_t)yb * Ws + xb] * wx;
    return top * (1.0 - wy) + bot * wy;
}

// NQ_IN planes staged (2), NQ_SUM planes summed (4 in sweep 1: I, p, II, Ip; 2 in sweep 2: a, b)
template <int SWEEP, int GF_RUNY>
__global__ __launch_bounds__(256) void k_gf(const float* __restrict__ depth_lo, int Wlo, int Hlo,
                                            const uint8_t* __restrict__ guide, int W, int H, int r, double eps,
                                            double* __restrict__ A, double* __restrict__ B, float* __restrict__ out)
{
    constexpr int NS = SWEEP == 1 ? 4 : 2;
    constexpr int GF_TY = 4 * GF_RUNY;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int rows = GF_TY + 2 * r;                 // staged rows
    const int pitch = (GF_TX + 2 * r) | 1;          // odd pitch: row-adjacent threads hit different banks
    const int hp = GF_TX + 1;                       // pitch of the horizontal-sum planes
    double* t0 = smem;                              // I  (sweep 1) / a (sweep 2)
    double* t1 = t0 + rows * pitch;                 // p  (sweep 1) / b (sweep 2)
    double* hs = t1 + rows * pitch;                 // [NS][rows][hp]

    const int tid = threadIdx.x;
    const int ox = blockIdx.x * GF_TX, oy = blockIdx.y * GF_TY;
    const double sx = (double)Wlo / (double)W, sy = (double)Hlo / (double)H;

    // ---- stage the halo tile; out-of-image entries contribute zero ----
    const int tw = GF_TX + 2 * r;
    for (int i = tid; i < rows * tw; i += 256) {
        const int ty = i / tw, tx = i - ty * tw;
        const int gx = ox - r + tx, gy = oy - r + ty;
        double v0 = 0.0, v1 = 0.0;
        if (gx >= 0 && gx < W && gy >= 0 && gy < H) {
            if (SWEEP == 1) {
                v0 = (double)guide[(size_t)gy * W + gx] / 255.0;
                v1 = gf_bilinear(depth_lo, Wlo, Hlo, sx, sy, gx, gy);
            } else {
                v0 = A[(size_t)gy * W + gx];
                v1 = B[(size_t)gy * W + gx];
            }
        }
        t0[ty * pitch + tx] = v0;
        t1[ty * pitch + tx] = v1;
    }
    __syncthreads();

    // ---- horizontal sliding sums: task = (row, run of GF_RUN outputs) ----
    const int nruns = GF_TX / GF_RUN;
    for (int task = tid; task < rows * nruns; task += 256) {
        const int row = task % rows, run = task / rows;
        const double* r0 = t0 + row * pitch + run * GF_RUN;
        const double* r1 = t1 + row * pitch + run * GF_RUN;
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        for (int k = 0; k <= 2 * r; k++) {
            const double u = r0[k], v = r1[k];
            s0 += u; s1 += v;
            if (SWEEP == 1) { s2 += u * u; s3 += u * v; }
        }
        double* h = hs + row * hp + run * GF_RUN;
        const int hplane = rows * hp;
        for (int j = 0; j < GF_RUN; j++) {
            h[j] = s0; h[hplane + j] = s1;
            if (SWEEP == 1) { h[2 * hplane + j] = s2; h[3 * hplane + j] = s3; }
            if (j + 1 < GF_RUN) {
                const double un = r0[j + 2 * r + 1], vn = r1[j + 2 * r + 1];  // column entering the window
                const double uo = r0[j], vo = r1[j];                          // column leaving it
                s0 += un - uo; s1 += vn - vo;
                if (SWEEP == 1) { s2 += un * un - uo * uo; s3 += un * vn - uo * vo; }
            }
        }
    }
    __syncthreads();

    // ---- vertical sliding sums + the per-pixel algebra: task = (x, run of GF_RUNY outputs) ----
    {
        const int x = tid % GF_TX, runy = tid / GF_TX;                    // 64 x 4 tasks
        const int hplane = rows * hp;
        const double* h = hs + (runy * GF_RUNY) * hp + x;
        double s[NS];
#pragma unroll
        for (int q = 0; q < NS; q++) s[q] = 0.0;
        for (int k = 0; k <= 2 * r; k++)
#pragma unroll
            for (int q = 0; q < NS; q++) s[q] += h[q * hplane + k * hp];
        const int gx = ox + x;
        const int cx = min(gx + r, W - 1) - max(gx - r, 0) + 1;
        for (int j = 0; j < GF_RUNY; j++) {
            const int gy = oy + runy * GF_RUNY + j;
            if (gx < W && gy < H) {
                const int cy = min(gy + r, H - 1) - max(gy - r, 0) + 1;
                const double cnt = (double)(cx * cy);
                if (SWEEP == 1) {
                    const double mI = s[0] / cnt, mp = s[1] / cnt, mII = s[2] / cnt, mIp = s[3] / cnt;
                    const double var = mII - mI * mI, cov = mIp - mI * mp;
                    const double a = cov / (var + eps);
                    A[(size_t)gy * W + gx] = a;
                    B[(size_t)gy * W + gx] = mp - a * mI;
                } else {
                    const double I = (double)guide[(size_t)gy * W + gx] / 255.0;
                    out[(size_t)gy * W + gx] = (float)((s[0] / cnt) * I + (s[1] / cnt));
                }
            }
            if (j + 1 < GF_RUNY) {
                const int kn = j + 2 * r + 1;
#pragma unroll
                for (int q = 0; q < NS; q++) s[q] += h[q * hplane + kn * hp] - h[q * hplane + j * hp];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Fast path for r in {4, 8}: column-marching sweeps.  A workgroup owns a strip of 256 - 2r output
// columns (+ r halo each side, one thread per column) and marches down a band of rows.  The vertical box
// sums slide in REGISTERS (a compile-time ring of the last 2r+1 rows' values per column: add the entering
// row, subtract the leaving one); only the per-row vertical sums cross lanes, through a 2 x NS x 256 f64
// LDS row buffer, for the 2r+1-tap horizontal sum.  16 KB of LDS per workgroup -> full occupancy, one
// barrier per row, every input element is read once per band (+ 2r warm-up rows).
// ------------------------------------------------------------------------------------------------
template <int SWEEP, int RR>
__global__ __launch_bounds__(256) void k_gfm(const float* __restrict__ depth_lo, int Wlo, int Hlo,
                                             const uint8_t* __restrict__ guide, int W, int H, double eps, int band_h,
                                             double* __restrict__ A, double* __restrict__ B, float* __restrict__ out,
                                             size_t depth_stride, size_t guide_stride)
{
    constexpr int R = 2 * RR + 1;
    // sweep 1 sums {g, g*g} as exact int32 (the guide is 8-bit: sum(I) = sum(g)/255, sum(I*I) = sum(g*g)/255^2) and
    // {p, g*p} in f64; sweep 2 sums {a, b} in f64
    __shared__ double sVd[2][2][256];
    __shared__ int sVi[2][2][256];
    {   // frame of the batch
        const size_t f = blockIdx.z, n4 = (size_t)W * H;
        depth_lo += f * depth_stride; guide += f * guide_stride; A += f * 2 * n4; B += f * 2 * n4; out += f * n4;
    }
    const int tid = threadIdx.x;
    const int gx = blockIdx.x * (256 - 2 * RR) - RR + tid;          // this thread's image column
    const int ya = blockIdx.y * band_h, yb = min(ya + band_h, H);
    const bool col_ok = gx >= 0 && gx < W;
    const bool out_col = tid >= RR && tid < 256 - RR && gx < W;
    const double sx = (double)Wlo / (double)W, sy = (double)Hlo / (double)H;
    const int cx = min(gx + RR, W - 1) - max(gx - RR, 0) + 1;
    const int nsteps = (yb - ya) + 2 * RR;

    // bilinear source coordinates are separable: the x part is a per-thread constant, the y part per row
    int bxa = 0, bxb = 0; double bwx = 0.0;
    if (SWEEP == 1) {
        const double fx = (gx + 0.5) * sx - 0.5, x0f = floor(fx);
        bwx = fx - x0f;
        bxa = min(max((int)x0f, 0), Wlo - 1); bxb = min(max((int)x0f + 1, 0), Wlo - 1);
    }
    double r0[R], r1[R], v0 = 0.0, v1 = 0.0;     // sweep 1: r1 = p ring, v0 = sum p, v1 = sum g*p ; sweep 2: a, b rings and sums
    int rg[R], vg = 0, vgg = 0;                   // sweep 1: g ring, sum g, sum g*g
#pragma unroll
    for (int j = 0; j < R; j++) { r0[j] = 0.0; r1[j] = 0.0; rg[j] = 0; }

    for (int t0 = 0; t0 < nsteps; t0 += R) {
#pragma unroll
        for (int j = 0; j < R; j++) {
            const int t = t0 + j;
            if (t < nsteps) {                                        // uniform
                const int e = ya - RR + t;                           // row entering the window
                const bool in = col_ok && e >= 0 && e < H;
                if (SWEEP == 1) {
                    int gn = 0; double pn = 0.0;
                    if (in) {
                        gn = guide[(size_t)e * W + gx];
                        const double fy = (e + 0.5) * sy - 0.5, y0f = floor(fy), wy = fy - y0f;
                        const float* ra = depth_lo + (size_t)min(max((int)y0f, 0), Hlo - 1) * Wlo;
                        const float* rb = depth_lo + (size_t)min(max((int)y0f + 1, 0), Hlo - 1) * Wlo;
                        const double top = (double)ra[bxa] * (1.0 - bwx) + (double)ra[bxb] * bwx;
                        const double bot = (double)rb[bxa] * (1.0 - bwx) + (double)rb[bxb] * bwx;
                        pn = top * (1.0 - wy) + bot * wy;
                    }
                    const int go = rg[j]; const double po = r1[j];   // row e - R leaves (zeros during warm-up)
                    rg[j] = gn; r1[j] = pn;
                    vg += gn - go; vgg += gn * gn - go * go;
                    v0 += pn - po; v1 += (double)gn * pn - (double)go * po;
                } else {
                    double n0 = 0.0, n1 = 0.0;
                    if (in) { n0 = __builtin_nontemporal_load(A + (size_t)e * W + gx); n1 = __builtin_nontemporal_load(B + (size_t)e * W + gx); }
                    const double o0 = r0[j], o1 = r1[j];
                    r0[j] = n0; r1[j] = n1;
                    v0 += n0 - o0; v1 += n1 - o1;
                }
                const int y = e - RR;                                // output row whose window is now complete
                if (y >= ya) {                                       // uniform
                    const int buf = t & 1;
                    sVd[buf][0][tid] = v0; sVd[buf][1][tid] = v1;
                    if (SWEEP == 1) { sVi[buf][0][tid] = vg; sVi[buf][1][tid] = vgg; }
                    __syncthreads();
                    if (out_col) {
                        double s0 = 0.0, s1 = 0.0; int sg = 0, sgg = 0;
#pragma unroll
                        for (int k = -RR; k <= RR; k++) {
                            s0 += sVd[buf][0][tid + k]; s1 += sVd[buf][1][tid + k];
                            if (SWEEP == 1) { sg += sVi[buf][0][tid + k]; sgg += sVi[buf][1][tid + k]; }
                        }
                        const int cy = min(y + RR, H - 1) - max(y - RR, 0) + 1;
                        const double inv = gf_rcp((double)(cx * cy));
                        if (SWEEP == 1) {
                            const double mI = (double)sg * (inv * (1.0 / 255.0)), mp = s0 * inv;
                            const double mII = (double)sgg * (inv * (1.0 / 65025.0)), mIp = s1 * (inv * (1.0 / 255.0));
                            const double var = mII - mI * mI, cov = mIp - mI * mp;
                            const double a = cov * gf_rcp(var + eps);
                            __builtin_nontemporal_store(a, A + (size_t)y * W + gx);
                            __builtin_nontemporal_store(mp - a * mI, B + (size_t)y * W + gx);
                        } else {
                            const double I = (double)guide[(size_t)y * W + gx] * (1.0 / 255.0);
                            __builtin_nontemporal_store((float)((s0 * inv) * I + (s1 * inv)), out + (size_t)y * W + gx);
                        }
                    }
                }
            }
        }
    }
}

template <int RR>
static void launch_gfm(const float* depth_lo, int Wlo, int Hlo, const uint8_t* guide, int W, int H, double eps,
                       double* A, double* B, float* out, int n, size_t depth_stride, size_t guide_stride, hipStream_t st)
{
    // band heights: sweep 1 is f64-compute bound (smaller bands = more workgroups), sweep 2 is bound by its halo'd
    // re-reads of the f64 a/b planes (taller bands = less halo)
    const char* e1 = getenv("V3D_GF_BAND1");
    const char* e2 = getenv("V3D_GF_BAND2");
    const int band1 = e1 ? atoi(e1) : 40, band2 = e2 ? atoi(e2) : 270;
    const dim3 grid1(v3d_cdiv(W, 256 - 2 * RR), v3d_cdiv(H, band1), n), grid2(v3d_cdiv(W, 256 - 2 * RR), v3d_cdiv(H, band2), n);
    hipLaunchKernelGGL((k_gfm<1, RR>), grid1, dim3(256), 0, st, depth_lo, Wlo, Hlo, guide, W, H, eps, band1, A, B, out, depth_stride, guide_stride);
    hipLaunchKernelGGL((k_gfm<2, RR>), grid2, dim3(256), 0, st, depth_lo, Wlo, Hlo, guide, W, H, eps, band2, A, B, out, depth_stride, guide_stride);
}

extern "C" size_t v3d_guided_upscale_ws_bytes(int W, int H)
{
    if (W < 1 || H < 1) return 0;
    return (size_t)W * H * sizeof(double) * 2;
}

static size_t gf_smem(int r, int ns, int ty)
{
    const int rows = ty + 2 * r;
    const int pitch = (GF_TX + 2 * r) | 1;
    return sizeof(double) * ((size_t)2 * rows * pitch + (size_t)ns * rows * (GF_TX + 1));
}

// n frames: frame f at depth_lo + f*depth_stride (floats), guide + f*guide_stride (bytes), out + f*W*H;
// ws must hold n * v3d_guided_upscale_ws_bytes(W, H)
extern "C" int v3d_guided_upscale_batch(const float* depth_lo, int Wlo, int Hlo, size_t depth_stride, const uint8_t* guide,
                                        int W, int H, size_t guide_stride, int n, int r, float eps, float* out, void* ws, void* stream)
{
    if (n < 1) { v3d_set_error("bad batch"); return V3D_ERR_ARG; }
    if (!depth_lo || !guide || !out || !ws) { v3d_set_error("null pointer"); return V3D_ERR_ARG; }
    if (Wlo < 1 || Hlo < 1 || W < 1 || H < 1) { v3d_set_error("bad geometry"); return V3D_ERR_ARG; }
    if (r < 1 || r > GF_RMAX) { v3d_set_error("radius %d outside [1, %d]", r, GF_RMAX); return V3D_ERR_UNSUPPORTED; }
    if (!(eps >= 0.f)) { v3d_set_error("eps must be >= 0"); return V3D_ERR_ARG; }
    hipStream_t st = (hipStream_t)stream;
    double* A = reinterpret_cast<double*>(ws);
    double* B = A + (size_t)W * H;
    if (!getenv("V3D_GF_TILED") && (r == 4 || r == 8)) {     // larger rings spill: r = 16 takes the tiled kernel
        if (r == 4) launch_gfm<4>(depth_lo, Wlo, Hlo, guide, W, H, (double)eps, A, B, out, n, depth_stride, guide_stride, st);
        else launch_gfm<8>(depth_lo, Wlo, Hlo, guide, W, H, (double)eps, A, B, out, n, depth_stride, guide_stride, st);
        V3D_LAUNCH_CHECK();
        return V3D_OK;
    }
    for (int f = 0; f < n; f++) {
    const float* depth_lo_f = depth_lo + (size_t)f * depth_stride; const uint8_t* guide_f = guide + (size_t)f * guide_stride;
    double* A = reinterpret_cast<double*>(ws) + (size_t)f * 2 * W * H; double* B = A + (size_t)W * H; float* out_f = out + (size_t)f * W * H;
    const int ty = r <= 8 ? 16 : 8;
    const dim3 grid(v3d_cdiv(W, GF_TX), v3d_cdiv(H, ty));
    const size_t sm1 = gf_smem(r, 4, ty), sm2 = gf_smem(r, 2, ty);
    if (ty == 16) {
        // above the default dynamic-LDS limit: opt in (160 KiB per CU on gfx950)
        V3D_HIP_CHECK(hipFuncSetAttribute((const void*)k_gf<1, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm1));
        V3D_HIP_CHECK(hipFuncSetAttribute((const void*)k_gf<2, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm2));
        hipLaunchKernelGGL((k_gf<1, 4>), grid, dim3(256), sm1, st, depth_lo_f, Wlo, Hlo, guide_f, W, H, r, (double)eps, A, B, out_f);
        hipLaunchKernelGGL((k_gf<2, 4>), grid, dim3(256), sm2, st, depth_lo_f, Wlo, Hlo, guide_f, W, H, r, (double)eps, A, B, out_f);
    } else {
        V3D_HIP_CHECK(hipFuncSetAttribute((const void*)k_gf<1, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm1));
        V3D_HIP_CHECK(hipFuncSetAttribute((const void*)k_gf<2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm2));
        hipLaunchKernelGGL((k_gf<1, 2>), grid, dim3(256), sm1, st, depth_lo_f, Wlo, Hlo, guide_f, W, H, r, (double)eps, A, B, out_f);
        hipLaunchKernelGGL((k_gf<2, 2>), grid, dim3(256), sm2, st, depth_lo_f, Wlo, Hlo, guide_f, W, H, r, (double)eps, A, B, out_f);
    }
    }
    V3D_LAUNCH_CHECK();
    return V3D_OK;
}

extern "C" int v3d_guided_upscale(const float* depth_lo, int Wlo, int Hlo, const uint8_t* guide, int W, int H,
                                  int r, float eps, float* out, void* ws, void* stream)
{
    return v3d_guided_upscale_batch(depth_lo, Wlo, Hlo, 0, guide, W, H, 0, 1, r, eps, out, ws, stream);
}
