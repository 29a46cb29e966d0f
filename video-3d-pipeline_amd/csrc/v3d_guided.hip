// v3d_guided.hip -- guided-filter joint upsampling of the 1080p depth to the 4K guide frame.
//
// Stands where reference upscale.py:21-73 (upscale_depth_maps_ffmpeg: ffmpeg `scale` filter)
// stands; re-specified by BASELINE.json / SURVEY.md 8a-11 + Appendix B.1 as He-Sun-Tang guided
// filtering:  p = bilinear(depth_lo), I = guide/255,
//   a = cov(I,p) / (var(I) + eps),  b = mean(p) - a*mean(I),  q = mean(a)*I + mean(b)
// with (2r+1)^2 box means clipped at the image border and divided by the true pixel count.
//
// Two sweeps (k_gfm for r in {4, 8}: column-marching strips; k_gf for other radii: LDS tiles), box sums
// separable with sliding windows.  HBM traffic: sweep 1 reads guide + depth_lo, writes a,b; sweep 2 reads a,b +
// guide, writes q.
// Numerics: all sums, a and b are float64 (full rate per instruction on gfx950, but no packed form): next to
// zero-depth regions q is ~1e-4 while the window holds values ~40, and the 1e-3 *relative* parity bar cannot be
// met there with f32 cancellation in cov/var and in box(b).
#include "v3d_common.h"
#include <type_traits>

// 1/x to full double precision: v_rcp_f64 seed (~26 bits) + two Newton steps (5 instructions instead of the
// ~30 of an IEEE division; the last-ulp difference is far inside the 1e-3 parity bar)
__device__ __forceinline__ double gf_rcp(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    r = fma(r, fma(-x, r, 1.0), r);
    r = fma(r, fma(-x, r, 1.0), r);
    return r;
}

// the low-resolution depth comes either as float32 (depth.py's frame-out surface, any provider) or straight as the matcher's
// int16 disparity x16: then depth.py:341 `/16` and depth.py:374 `<= 0 -> 0` happen in the load (the float map never exists)
template <typename TD> __device__ __forceinline__ float gf_ld(const TD* p, size_t i);
template <> __device__ __forceinline__ float gf_ld<float>(const float* p, size_t i) { return p[i]; }
template <> __device__ __forceinline__ float gf_ld<int16_t>(const int16_t* p, size_t i) { const int d = p[i]; return d > 0 ? (float)d * 0.0625f : 0.f; }

#define GF_TX 64
#define GF_RUN 8      // outputs per thread along x in the horizontal pass
#define GF_RMAX 16
// tile height TY = 4 * RUNY: 16 rows for r <= 8, 8 rows above (keeps the f64 tile inside the 160 KiB LDS)

template <typename TD>
__device__ __forceinline__ double gf_bilinear(const TD* __restrict__ src, int Ws, int Hs, double sx, double sy, int x, int y)
{
    const double fx = (x + 0.5) * sx - 0.5, fy = (y + 0.5) * sy - 0.5;
    const double x0f = floor(fx), y0f = floor(fy);
    const double wx = fx - x0f, wy = fy - y0f;
    const int x0 = (int)x0f, y0 = (int)y0f;
    const int xa = min(max(x0, 0), Ws - 1), xb = min(max(x0 + 1, 0), Ws - 1);
    const int ya = min(max(y0, 0), Hs - 1), yb = min(max(y0 + 1, 0), Hs - 1);
    const double top = (double)gf_ld(src, (size_t)ya * Ws + xa) * (1.0 - wx) + (double)gf_ld(src, (size_t)ya * Ws + xb) * wx;
    const double bot = (double)gf_ld(src, (size_t)yb * Ws + xa) * (1.0 - wx) + (double)gf_ld(src, (size_t)yb * Ws + xb) * wx;
    return top * (1.0 - wy) + bot * wy;
}

// NQ_IN planes staged (2), NQ_SUM planes summed (4 in sweep 1: I, p, II, Ip; 2 in sweep 2: a, b)
template <int SWEEP, int GF_RUNY, typename TD>
__global__ __launch_bounds__(256) void k_gf(const TD* __restrict__ depth_lo, int Wlo, int Hlo,
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
// columns (+ r halo each side, one thread per column) and marches down a band of rows, TWO rows per step.
// The vertical box sums slide in REGISTERS (a compile-time ring of the last 2r+1 rows' values per column:
// add the entering row, subtract the leaving one); only the per-row vertical sums cross lanes, through a
// double-buffered LDS row pair, for the 2r+1-tap horizontal sum.  In that phase a thread owns a PAIR of adjacent
// output pixels of one of the two rows: their windows share 2r of 2r+1 columns, so 2r+2 values -- r+1 aligned
// 16-byte LDS reads per quantity -- and 2r+1 adds serve both (half the adds, and ds_read_b128 moves 256 B/clk
// where the ds_read2_b64 pairs the compiler forms from single f64 reads move 128: measured, the one-pixel-per-
// thread version spent 72 % of its time in the LDS pipe).  One barrier per two rows, every input element is read
// once per band (+ 2r warm-up rows), a/b/out leave as 16/16/8-byte stores.
// ------------------------------------------------------------------------------------------------
#ifndef GF_CH
#define GF_CH 3      // 16-byte LDS reads in flight per plane in the horizontal phase
#endif
typedef double v3d_f64x2 __attribute__((ext_vector_type(2)));
typedef float v3d_f32x2 __attribute__((ext_vector_type(2)));

template <int SWEEP, int RR, typename TD>
__global__ __launch_bounds__(256) void k_gfm(const TD* __restrict__ depth_lo, int Wlo, int Hlo,
                                             const uint8_t* __restrict__ guide, int W, int H, double eps, int band_h,
                                             double* __restrict__ A, double* __restrict__ B, float* __restrict__ out,
                                             size_t depth_stride, size_t guide_stride)
{
    static_assert(RR % 2 == 0, "pairs must not straddle the strip's halo boundary");
    constexpr int R = 2 * RR + 1, NOUT = 256 - 2 * RR;
    // sweep 1 sums {g, g*g} as exact int32 (the guide is 8-bit: sum(I) = sum(g)/255, sum(I*I) = sum(g*g)/255^2) and
    // {p, g*p} in f64; sweep 2 sums {a, b} in f64
    __shared__ __attribute__((aligned(16))) double sVd[2][2][2][256];       // [buffer][row of the pair][quantity][column]
    __shared__ __attribute__((aligned(16))) int2 sVi[2][2][256];            // [buffer][row][column] {sum g, sum g*g}
    {   // frame of the batch
        const size_t f = blockIdx.z, n4 = (size_t)W * H;
        depth_lo += f * depth_stride; guide += f * guide_stride; A += f * 2 * n4; B += f * 2 * n4; out += f * n4;
    }
    const int tid = threadIdx.x;
    const int gx = blockIdx.x * NOUT - RR + tid;                    // vertical phase: this thread's image column
    const int ya = blockIdx.y * band_h, yb = min(ya + band_h, H);
    const bool col_ok = gx >= 0 && gx < W;
    const double sx = (double)Wlo / (double)W, sy = (double)Hlo / (double)H;
    const int nsteps = (yb - ya) + 2 * RR;

    // horizontal phase: row hrow of the step's pair, output pixels (hgx, hgx + 1)
    const int hrow = tid >> 7, hq = tid & 127;
    const int hgx = blockIdx.x * NOUT - RR + 2 * hq;
    const bool pair_in = 2 * hq >= RR && 2 * hq < 256 - RR;
    const bool px0 = pair_in && hgx < W, px1 = pair_in && hgx + 1 < W;
    const bool vec_ok = px1 && (W & 1) == 0;                        // both pixels in the image and rows pair-aligned

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

    // inputs of the two rows a step consumes, fetched one step ahead (the loads of step s+1 fly during step s)
    // (loads are UNCONDITIONAL, from clamped addresses, and the out-of-image case is a select afterwards: behind a
    //  branch the compiler's s_waitcnt model can no longer count them and waits for the prefetch it just issued)
    struct RowIn { int g; float a0, a1, b0, b1; double n0, n1; bool in; };
    const int gxc = min(max(gx, 0), W - 1);
    auto fetch_row = [&](int t) -> RowIn {
        RowIn q; q.g = 0; q.a0 = q.a1 = q.b0 = q.b1 = 0.f; q.n0 = q.n1 = 0.0;
        const int e = ya - RR + t;                                       // row entering the window
        q.in = col_ok && e >= 0 && e < H && t < nsteps;
        const size_t o = (size_t)min(max(e, 0), H - 1) * W + gxc;
        if (SWEEP == 1) {
            q.g = guide[o];
            const double fy = (e + 0.5) * sy - 0.5, y0f = floor(fy);
            const TD* ra = depth_lo + (size_t)min(max((int)y0f, 0), Hlo - 1) * Wlo;
            const TD* rb = depth_lo + (size_t)min(max((int)y0f + 1, 0), Hlo - 1) * Wlo;
            q.a0 = gf_ld(ra, bxa); q.a1 = gf_ld(ra, bxb); q.b0 = gf_ld(rb, bxa); q.b1 = gf_ld(rb, bxb);
        } else {
            q.n0 = __builtin_nontemporal_load(A + o); q.n1 = __builtin_nontemporal_load(B + o);
        }
        return q;
    };
    RowIn nx[2] = { fetch_row(0), fetch_row(1) };

    int buf = 0;
    for (int t0 = 0; t0 < nsteps; t0 += 2 * R) {
#pragma unroll
        for (int s = 0; s < R; s++) {
            const int tA = t0 + 2 * s;
            if (tA < nsteps) {                                           // uniform
                const int y0 = ya - 2 * RR + tA;                         // output row completed by the first row of the pair
                const bool emit = y0 >= ya;                              // uniform (2*RR is even: pairs never straddle ya)
                const RowIn cur[2] = { nx[0], nx[1] };
                nx[0] = fetch_row(tA + 2); nx[1] = fetch_row(tA + 3);
#pragma unroll
                for (int rr = 0; rr < 2; rr++) {
                    const int j = (2 * s + rr) % R;                      // compile-time ring slot
                    if (SWEEP == 1) {
                        int gn = 0; double pn = 0.0;
                        if (cur[rr].in) {
                            gn = cur[rr].g;
                            const int e = ya - RR + tA + rr;
                            const double fy = (e + 0.5) * sy - 0.5, wy = fy - floor(fy);
                            // lerps as a + w * (b - a): one subtract and one fma each (the sweep is VALU-bound; the
                            // last-ulp difference to the (1-w)*a + w*b form is 1e-13 of the 1e-3 bar)
                            const double a0 = (double)cur[rr].a0, b0 = (double)cur[rr].b0;
                            const double top = fma(bwx, (double)cur[rr].a1 - a0, a0);
                            const double bot = fma(bwx, (double)cur[rr].b1 - b0, b0);
                            pn = fma(wy, bot - top, top);
                        }
                        const int go = rg[j]; const double po = r1[j];   // row e - R leaves (zeros during warm-up)
                        rg[j] = gn; r1[j] = pn;
                        vg += gn - go; vgg += gn * gn - go * go;
                        v0 += pn - po; v1 = fma((double)gn, pn, fma(-(double)go, po, v1));
                    } else {
                        const double n0 = cur[rr].in ? cur[rr].n0 : 0.0, n1 = cur[rr].in ? cur[rr].n1 : 0.0;
                        const double o0 = r0[j], o1 = r1[j];
                        r0[j] = n0; r1[j] = n1;
                        v0 += n0 - o0; v1 += n1 - o1;
                    }
                    if (emit) {
                        sVd[buf][rr][0][tid] = v0; sVd[buf][rr][1][tid] = v1;
                        if (SWEEP == 1) sVi[buf][rr][tid] = make_int2(vg, vgg);
                    }
                }
                if (emit) {
                    __syncthreads();
                    const int y = y0 + hrow;
                    if (px0 && y < yb) {
                        // 2r+2 columns hgx-r .. hgx+1+r: the inner 2r are common to both pixels' windows.  Reads go out plane
                        // by plane in groups of three, fenced for the scheduler: all of them at once would hold 100+ registers
                        double fq[2], lq[2], cq[2];                       // per plane: first / last column, common part
#pragma unroll
                        for (int q = 0; q < 2; q++) {
                            const v3d_f64x2* d = reinterpret_cast<const v3d_f64x2*>(&sVd[buf][hrow][q][0]) + (hq - RR / 2);
                            double f = 0.0, l = 0.0, c = 0.0;
#pragma unroll
                            for (int ch = 0; ch <= RR; ch += GF_CH) {
                                v3d_f64x2 w[GF_CH];
#pragma unroll
                                for (int i = 0; i < GF_CH; i++) if (ch + i <= RR) w[i] = d[ch + i];
#pragma unroll
                                for (int i = 0; i < GF_CH; i++) if (ch + i <= RR) {
                                    if (ch + i == 0) { f = w[i].x; c = w[i].y; }
                                    else if (ch + i == RR) { c += w[i].x; l = w[i].y; }
                                    else { c += w[i].x; c += w[i].y; }
                                }
                                asm volatile("" : "+v"(c) :: "memory");   // operand: the chunk's adds cannot sink below the next chunk's reads
                            }
                            fq[q] = f; lq[q] = l; cq[q] = c;
                        }
                        int g0 = 0, gg0 = 0, gl = 0, ggl = 0, cg = 0, cgg = 0;
                        if (SWEEP == 1) {
                            const int4* gi = reinterpret_cast<const int4*>(&sVi[buf][hrow][0]) + (hq - RR / 2);
#pragma unroll
                            for (int ch = 0; ch <= RR; ch += 3) {
                                int4 u[3];
#pragma unroll
                                for (int i = 0; i < 3; i++) if (ch + i <= RR) u[i] = gi[ch + i];
#pragma unroll
                                for (int i = 0; i < 3; i++) if (ch + i <= RR) {
                                    if (ch + i == 0) { g0 = u[i].x; gg0 = u[i].y; cg = u[i].z; cgg = u[i].w; }
                                    else if (ch + i == RR) { cg += u[i].x; cgg += u[i].y; gl = u[i].z; ggl = u[i].w; }
                                    else { cg += u[i].x + u[i].z; cgg += u[i].y + u[i].w; }
                                }
                                asm volatile("" : "+v"(cg), "+v"(cgg) :: "memory");
                            }
                        }
                        const double s0a = cq[0] + fq[0], s0b = cq[0] + lq[0], s1a = cq[1] + fq[1], s1b = cq[1] + lq[1];
                        const int cy = min(y + RR, H - 1) - max(y - RR, 0) + 1;
                        // (window widths recomputed here: two thread constants fewer keep sweep 1 inside 128 VGPRs)
                        const int cx0 = min(hgx + RR, W - 1) - max(hgx - RR, 0) + 1, cx1 = min(hgx + 1 + RR, W - 1) - max(hgx + 1 - RR, 0) + 1;
                        const double inva = gf_rcp((double)(cx0 * cy)), invb = cx1 == cx0 ? inva : gf_rcp((double)(cx1 * cy));
                        const size_t o = (size_t)y * W + hgx;
                        if (SWEEP == 1) {
                            const int sga = cg + g0, sgb = cg + gl, sgga = cgg + gg0, sggb = cgg + ggl;
                            double ab[2][2];
#pragma unroll
                            for (int n = 0; n < 2; n++) {
                                const double inv = n ? invb : inva;
                                const double mI = (double)(n ? sgb : sga) * (inv * (1.0 / 255.0)), mp = (n ? s0b : s0a) * inv;
                                const double mII = (double)(n ? sggb : sgga) * (inv * (1.0 / 65025.0)), mIp = (n ? s1b : s1a) * (inv * (1.0 / 255.0));
                                const double var = fma(-mI, mI, mII), cov = fma(-mI, mp, mIp);
                                const double a = cov * gf_rcp(var + eps);
                                ab[n][0] = a; ab[n][1] = fma(-a, mI, mp);
                            }
                            if (vec_ok) {
                                v3d_f64x2 va = { ab[0][0], ab[1][0] }, vb = { ab[0][1], ab[1][1] };
                                __builtin_nontemporal_store(va, reinterpret_cast<v3d_f64x2*>(A + o));
                                __builtin_nontemporal_store(vb, reinterpret_cast<v3d_f64x2*>(B + o));
                            } else {
                                A[o] = ab[0][0]; B[o] = ab[0][1];
                                if (px1) { A[o + 1] = ab[1][0]; B[o + 1] = ab[1][1]; }
                            }
                        } else {
                            const double Ia = (double)guide[o] * (1.0 / 255.0);
                            const double Ib = px1 ? (double)guide[o + 1] * (1.0 / 255.0) : 0.0;
                            const float qa = (float)((s0a * inva) * Ia + (s1a * inva)), qb = (float)((s0b * invb) * Ib + (s1b * invb));
                            if (vec_ok) {
                                v3d_f32x2 vq = { qa, qb };
                                __builtin_nontemporal_store(vq, reinterpret_cast<v3d_f32x2*>(out + o));
                            } else {
                                out[o] = qa;
                                if (px1) out[o + 1] = qb;
                            }
                        }
                    }
                    buf ^= 1;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Fused form for r in {4, 8}: BOTH box stages in one launch, a/b never touch HBM (the two-sweep form moves
// 2 x 16 B per 4K pixel through HBM for the f64 a/b planes: 8 of its 11 GB per 30 frames).
//
// A 512-thread workgroup owns a strip of 256 columns (224 outputs at r = 8) and marches down a band of rows, two
// rows per step, with its eight waves SPECIALISED:
//   waves 0-3 (stage 1)  exactly k_gfm<1>: vertical sliding sums of {g, g*g, p, g*p} in a register ring, horizontal
//                        window sums through LDS (pixel pairs), the per-pixel algebra -> a, b -- written to an LDS
//                        row pair instead of HBM;
//   waves 4-7 (stage 2)  exactly k_gfm<2>, two steps behind: pick the a/b rows up from LDS (one column per thread),
//                        vertical sliding sums in a second register ring, horizontal window sums through LDS, q -> HBM.
// Each role keeps ONE ring (<= 128 VGPRs, four waves per SIMD, two workgroups per CU); putting both rings into one
// thread needs ~180 VGPRs and halves the occupancy of a kernel that lives on its waves covering each other's LDS and
// barrier stalls.  All hand-offs are double-buffered LDS rows, so a step costs ONE workgroup barrier: in phase p
// stage 1 runs H1(p-1) then V1(p), stage 2 runs H2(p-3) then V2(p-2).  The arithmetic is that of the two-sweep kernels; since
// round 3 the window sums are re-associated (aligned pair sums, see PAIR SUMS below): equal to rounding, not by construction bit for bit.
// HBM traffic: guide + depth_lo (x 256/224 strip overlap, + 4r warm-up rows per band) in, q out.
// Measured on 30 4K frames: 2.15 ms against 2.72 ms for the two sweeps (VALU pipe 61 % busy, LDS pipe 62 %: the kernel is
// bound by its ~170 mostly-f64 instructions per pixel, no longer by HBM).  Tried, no gain: a wave-uniform fast path that
// skips the count reciprocals away from the border (more spills, 2.26 ms), s_setprio for the stage-1 waves (2.21-2.25 ms).
// Round 3 (34 frames, same box, tools/gf_ab.py): a build whose horizontal phases skip a third of H1's and 60 % of H2's reads and
// adds (a proxy build, results garbage) runs 65.7 us per frame against 75.1 -- so the window sums are worth ~15-20 %.  The form that saves them,
// SLIDING RUNS (a lane owns 4 or 8 consecutive outputs of a row, first window summed once, then one add and one subtract per
// output and plane; 22 or 30 adds for 4 or 8 outputs instead of 34 or 68), needs a quarter or an eighth of the lanes, so the
// horizontal phase of a step was given to a rotating worker group of two waves (runs of 4) or one wave (runs of 8) while the
// role's other waves went to the barrier: identical output, total instructions -15 %, and SLOWER -- 97.4 us per frame (runs of
// 4) and 151.4 (runs of 8) against 75.7.  A step is one barrier-to-barrier interval; its length is the LATENCY of the longest
// wave, and a worker that sums a window and then walks 8 dependent a/b solves (two reciprocal refinements each, ~25 dependent
// f64 operations per output) is three times as long as a wave that does one pixel pair.  The pair form spreads exactly that
// chain over all lanes; what remains is its instruction count.  (Kernel kept out of the tree; numbers in DESIGN.md.)
// ------------------------------------------------------------------------------------------------
// value of the lane's pair partner (lane ^ 1): two DPP moves for the halves of a double
__device__ __forceinline__ double gf_partner(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), V3D_DPP_QUAD(1, 0, 3, 2), 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), V3D_DPP_QUAD(1, 0, 3, 2), 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ int gf_partner(int v) { return __builtin_amdgcn_update_dpp(0, v, V3D_DPP_QUAD(1, 0, 3, 2), 0xF, 0xF, true); }

// PAIR SUMS (round 3).  The horizontal phases are bound by the LDS pipe: a 2r+2-column window costs r+1 16-byte reads per plane
// and thread, ~1400 LDS cycles per workgroup and step.  The vertical phase therefore also leaves, next to every column sum, the
// sum of each ALIGNED column pair (own value + the lane partner's, one DPP exchange): a window that starts on an even column is
// r+1 pair sums, and the two outputs of a thread are that total minus the one column each of them does not cover --
// r+1 8-byte reads + 2 singles instead of r+1 16-byte reads, r+2 adds instead of 2r+1.
// I1 (int16 disparity in, exact 2x upscale): stage 1's four window sums are EXACT INTEGERS.  The disparity is d/16 with d <= 1023
// and the x2 bilinear weights are {1,3}/4 per axis, so P = 256 p = sum w d (w in {1,3,9}) is an integer <= 16368; over a 17 x 17
// window sum P < 2^23 and sum g P < 2^31.  The ring holds (P << 8 | g) in ONE register per row (17 instead of 34 + 5), the
// vertical sums are four int32, the row pair's sums cross the lanes as one int4 per column (16 bytes instead of 24), and the
// horizontal phase adds integers three at a time (v_add3_u32).  The a/b algebra converts the four exact sums to f64 -- the very
// values the f64 sums of the general path hold (they are exact there too) -- so the output is bit-identical to it.
template <int RR, int COLS, typename TD, bool I1>
__global__ __launch_bounds__(2 * COLS, 4) void k_gff(const TD* __restrict__ depth_lo, int Wlo, int Hlo,
                                                const uint8_t* __restrict__ guide, int W, int H, double eps, int band_h,
                                                float* __restrict__ out, size_t depth_stride, size_t guide_stride)
{
    static_assert(RR % 2 == 0, "pairs must not straddle the strip's halo boundaries");
    static_assert(!I1 || std::is_same<TD, int16_t>::value, "the integer stage 1 takes the int16 disparity");
    constexpr int R = 2 * RR + 1, NOUT = COLS - 4 * RR, HP = COLS / 2;      // HP pixel pairs per row
    static_assert(COLS == 256 || COLS == 512, "strip width");
    // stage 1's row-pair sums: general path {sum p, sum g*p} f64 planes + {sum g, sum g*g} int2; I1: one int4 {sum g, sum g*g, sum P, sum g*P}
    constexpr int S1D = 2 * 2 * 2 * COLS * 8, S1I = 2 * 2 * COLS * 8;
    __shared__ __attribute__((aligned(16))) unsigned char sS1[I1 ? 2 * 2 * COLS * 16 : S1D + S1I];
    double (*sV1)[2][2][COLS] = reinterpret_cast<double (*)[2][2][COLS]>(sS1);               // [buffer][row of the pair][sum p | sum g*p][column]
    int2 (*sVi)[2][COLS] = reinterpret_cast<int2 (*)[2][COLS]>(sS1 + (I1 ? 0 : S1D));         // [buffer][row][column] {sum g, sum g*g}
    int4 (*sVq)[2][COLS] = reinterpret_cast<int4 (*)[2][COLS]>(sS1);                          // I1: [buffer][row][column]
    (void)sV1; (void)sVi; (void)sVq;
    __shared__ __attribute__((aligned(16))) double sAB[2][2][2][COLS];        // [buffer][row][a | b][column]   stage 1 -> stage 2
    __shared__ __attribute__((aligned(16))) double sV2[2][2][2][COLS];        // [buffer][row][sum a | sum b][column]
    __shared__ __attribute__((aligned(16))) double sP2[2][2][2][COLS / 2];    // [buffer][row][sum a | sum b][aligned column pair]
    __shared__ __attribute__((aligned(16))) int4 sPq[I1 ? 2 : 1][2][I1 ? COLS / 2 : 1];   // I1: [buffer][row][aligned column pair]
    (void)sPq;
    {   // frame of the batch
        const size_t f = blockIdx.z, n4 = (size_t)W * H;
        depth_lo += f * depth_stride; guide += f * guide_stride; out += f * n4;
    }
    const int role = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / COLS));    // 0: stage 1 (first half of the waves), 1: stage 2
    const int t = threadIdx.x % COLS;
    const int gx0 = blockIdx.x * NOUT - 2 * RR;                      // image column of strip column 0
    const int gx = gx0 + t;                                          // vertical phases: this thread's column
    const int ya = blockIdx.y * band_h, yb = min(ya + band_h, H);
    const int nrows = (yb - ya) + 4 * RR;                            // input rows ya - 2r .. yb - 1 + 2r
    const int NS = (nrows + 1) >> 1;                                 // steps (row pairs)
    const int NP = NS + 3;                                           // phases: the last H2 runs three phases behind the last V1
    const int hq = t % HP, hgx = gx0 + 2 * hq;                       // horizontal phases: pixel pair (2hq, 2hq+1) of row t / HP of the step

    for (int i = threadIdx.x; i < 2 * 2 * 2 * COLS; i += 2 * COLS) (&sAB[0][0][0][0])[i] = 0.0;
    __syncthreads();

    if (role == 0) {
      if constexpr (I1) {
        // ================= stage 1, exact integer sums (int16 disparity, exact 2x) =================
        const bool col_ok = gx >= 0 && gx < W;
        const bool pair_in = 2 * hq >= RR && 2 * hq < COLS - RR;      // a/b columns of the strip
        const bool px0 = pair_in && hgx >= 0 && hgx < W, px1 = pair_in && hgx + 1 >= 0 && hgx + 1 < W;
        // x2 bilinear source columns and weights (quarters): even x = 2k: (k-1, k) x (1, 3); odd: (k, k+1) x (3, 1)
        const int kx = gx >> 1;
        const int bxa = min(max((gx & 1) ? kx : kx - 1, 0), Wlo - 1), bxb = min(max((gx & 1) ? kx + 1 : kx, 0), Wlo - 1);
        const int wxa = (gx & 1) ? 3 : 1, wxb = 4 - wxa;
        uint32_t ring[R];                        // (P << 8) | g of the last 2r+1 rows
        int vg = 0, vgg = 0, vP = 0, vgP = 0;
#pragma unroll
        for (int j = 0; j < R; j++) ring[j] = 0u;
        struct RowIn { int g, a0, a1, b0, b1; bool in; };
        const int gxc = min(max(gx, 0), W - 1);
        auto fetch_row = [&](int tt) -> RowIn {        // unconditional loads from clamped addresses
            RowIn q;
            const int e = ya - 2 * RR + tt;                                      // input row entering the window
            q.in = col_ok && e >= 0 && e < H && tt < nrows;
            const int ec = min(max(e, 0), H - 1), ky = ec >> 1;
            q.g = guide[(size_t)ec * W + gxc];
            const TD* ra = depth_lo + (size_t)min(max((ec & 1) ? ky : ky - 1, 0), Hlo - 1) * Wlo;
            const TD* rb = depth_lo + (size_t)min(max((ec & 1) ? ky + 1 : ky, 0), Hlo - 1) * Wlo;
            q.a0 = max((int)ra[bxa], 0); q.a1 = max((int)ra[bxb], 0); q.b0 = max((int)rb[bxa], 0); q.b1 = max((int)rb[bxb], 0);   // depth.py:374: <= 0 -> 0
            return q;
        };
        RowIn nx[2] = { fetch_row(0), fetch_row(1) };
        for (int p0 = 0; p0 < NP; p0 += R) {
#pragma unroll
            for (int sp = 0; sp < R; sp++) {
                const int p = p0 + sp;
                if (p < NP) {                                                    // uniform
                    int tl = t;
                    asm volatile("" : "+v"(tl));                                 // (nothing of the horizontal phase hoisted out of the loop)
                    const int hrow = tl / HP, hq = tl % HP, hgx = gx0 + 2 * hq;
                    // ---- H1(p-1): window sums of the row pair V1(p-1) left in LDS -> a, b of two pixels -> sAB ----
                    const int j = p - 1;
                    if (j >= RR && j < NS && pair_in) {
                        const int hb = j & 1;
                        const int y = ya - 3 * RR + 2 * j + hrow;                // centre row of this window
                        double ab[2][2] = { { 0.0, 0.0 }, { 0.0, 0.0 } };
                        if (y >= 0 && y < H && px0) {
                            // columns 2hq - r .. 2hq + 1 + r = the r + 1 aligned pairs hq - r/2 .. hq + r/2: their total, then each
                            // output drops the one column it does not cover (output 0 the last, output 1 the first)
                            const int4* gp = &sPq[hb][hrow][hq - RR / 2];
                            const int4 f = sVq[hb][hrow][2 * hq - RR], l = sVq[hb][hrow][2 * hq + RR + 1];
                            int4 c = make_int4(0, 0, 0, 0);
#pragma unroll
                            for (int ch = 0; ch <= RR; ch += 3) {
                                int4 u[3];
#pragma unroll
                                for (int i = 0; i < 3; i++) if (ch + i <= RR) u[i] = gp[ch + i];
#pragma unroll
                                for (int i = 0; i < 3; i++) if (ch + i <= RR) { c.x += u[i].x; c.y += u[i].y; c.z += u[i].z; c.w += u[i].w; }
                                asm volatile("" : "+v"(c.x), "+v"(c.y), "+v"(c.z), "+v"(c.w) :: "memory");
                            }
                            const int cy = min(y + RR, H - 1) - max(y - RR, 0) + 1;
                            const int cx0 = min(hgx + RR, W - 1) - max(hgx - RR, 0) + 1, cx1 = min(hgx + 1 + RR, W - 1) - max(hgx + 1 - RR, 0) + 1;
                            const double inva = gf_rcp((double)(cx0 * cy)), invb = cx1 == cx0 ? inva : gf_rcp((double)(cx1 * cy));
#pragma unroll
                            for (int n = 0; n < 2; n++) {
                                const int4 e = n ? f : l;                    // the column this output does NOT cover
                                const double inv = n ? invb : inva;
                                const double s0 = (double)(c.z - e.z) * (1.0 / 256.0), s1 = (double)(c.w - e.w) * (1.0 / 256.0);   // sum p, sum g*p: exact
                                const double mI = (double)(c.x - e.x) * (inv * (1.0 / 255.0)), mp = s0 * inv;
                                const double mII = (double)(c.y - e.y) * (inv * (1.0 / 65025.0)), mIp = s1 * (inv * (1.0 / 255.0));
                                const double var = fma(-mI, mI, mII), cov = fma(-mI, mp, mIp);
                                const double a = cov * gf_rcp(var + eps);
                                ab[n][0] = a; ab[n][1] = fma(-a, mI, mp);
                            }
                            if (!px1) { ab[1][0] = 0.0; ab[1][1] = 0.0; }
                        }
                        // rows and columns outside the image hand ZERO a/b to stage 2 (its windows count in-image pixels only)
                        const v3d_f64x2 va2 = { ab[0][0], ab[1][0] }, vb2 = { ab[0][1], ab[1][1] };
                        *reinterpret_cast<v3d_f64x2*>(&sAB[hb][hrow][0][2 * hq]) = va2;
                        *reinterpret_cast<v3d_f64x2*>(&sAB[hb][hrow][1][2 * hq]) = vb2;
                    }
                    // ---- V1(p): two input rows enter the column's window ----
                    if (p < NS) {
                        const int tA = 2 * p;
                        const RowIn cur[2] = { nx[0], nx[1] };
                        nx[0] = fetch_row(tA + 2); nx[1] = fetch_row(tA + 3);
                        const bool emit = p >= RR;                               // uniform
#pragma unroll
                        for (int rr = 0; rr < 2; rr++) {
                            const int slot = (2 * sp + rr) % R;                  // compile-time ring slot
                            int gn = 0, Pn = 0;
                            if (cur[rr].in) {
                                gn = cur[rr].g;
                                const int e = ya - 2 * RR + tA + rr;             // (in the image here)
                                const int wya = (e & 1) ? 3 : 1;
                                Pn = wya * (wxa * cur[rr].a0 + wxb * cur[rr].a1) + (4 - wya) * (wxa * cur[rr].b0 + wxb * cur[rr].b1);   // 256 p
                            }
                            const int go = (int)(ring[slot] & 0xFFu), Po = (int)(ring[slot] >> 8);
                            ring[slot] = ((uint32_t)Pn << 8) | (uint32_t)gn;
                            vg += gn - go; vgg += gn * gn - go * go;
                            vP += Pn - Po; vgP += gn * Pn - go * Po;
                            if (emit) {
                                sVq[p & 1][rr][t] = make_int4(vg, vgg, vP, vgP);
                                const int4 ps = make_int4(vg + gf_partner(vg), vgg + gf_partner(vgg), vP + gf_partner(vP), vgP + gf_partner(vgP));
                                if (!(t & 1)) sPq[p & 1][rr][t >> 1] = ps;
                            }
                        }
                    }
                    __syncthreads();
                }
            }
        }
      } else {
        // ================= stage 1: guide + depth -> a, b (k_gfm<1>'s arithmetic) =================
        const bool col_ok = gx >= 0 && gx < W;
        const double sx = (double)Wlo / (double)W, sy = (double)Hlo / (double)H;
        const bool pair_in = 2 * hq >= RR && 2 * hq < COLS - RR;      // a/b columns of the strip
        const bool px0 = pair_in && hgx >= 0 && hgx < W, px1 = pair_in && hgx + 1 >= 0 && hgx + 1 < W;
        int bxa, bxb; double bwx;
        {
            const double fx = (gx + 0.5) * sx - 0.5, x0f = floor(fx);
            bwx = fx - x0f;
            bxa = min(max((int)x0f, 0), Wlo - 1); bxb = min(max((int)x0f + 1, 0), Wlo - 1);
        }
        double r1[R], v0 = 0.0, v1 = 0.0;        // p ring, sum p, sum g*p
        uint32_t rgw[(R + 3) / 4];               // g ring, one BYTE per row (the guide is 8-bit): 5 registers instead of 17 keep
        int vg = 0, vgg = 0;                     //   the kernel inside the 128 VGPRs of four waves per SIMD; sum g, sum g*g
#pragma unroll
        for (int j = 0; j < R; j++) r1[j] = 0.0;
#pragma unroll
        for (int j = 0; j < (R + 3) / 4; j++) rgw[j] = 0u;
        struct RowIn { int g; float a0, a1, b0, b1; bool in; };
        const int gxc = min(max(gx, 0), W - 1);
        auto fetch_row = [&](int tt) -> RowIn {        // unconditional loads from clamped addresses (see k_gfm)
            RowIn q;
            const int e = ya - 2 * RR + tt;                                      // input row entering the window
            q.in = col_ok && e >= 0 && e < H && tt < nrows;
            const size_t o = (size_t)min(max(e, 0), H - 1) * W + gxc;
            q.g = guide[o];
            const double fy = (e + 0.5) * sy - 0.5, y0f = floor(fy);
            const TD* ra = depth_lo + (size_t)min(max((int)y0f, 0), Hlo - 1) * Wlo;
            const TD* rb = depth_lo + (size_t)min(max((int)y0f + 1, 0), Hlo - 1) * Wlo;
            q.a0 = gf_ld(ra, bxa); q.a1 = gf_ld(ra, bxb); q.b0 = gf_ld(rb, bxa); q.b1 = gf_ld(rb, bxb);
            return q;
        };
        RowIn nx[2] = { fetch_row(0), fetch_row(1) };
        for (int p0 = 0; p0 < NP; p0 += R) {
#pragma unroll
            for (int sp = 0; sp < R; sp++) {
                const int p = p0 + sp;
                if (p < NP) {                                                    // uniform
                    // (thread constants of the horizontal phase are re-derived per phase, see stage 2: nothing hoisted, nothing spilled)
                    int tl = t;
                    asm volatile("" : "+v"(tl));
                    const int hrow = tl / HP, hq = tl % HP, hgx = gx0 + 2 * hq;
                    // ---- H1(p-1): window sums of the row pair V1(p-1) left in LDS -> a, b of two pixels -> sAB ----
                    const int j = p - 1;
                    if (j >= RR && j < NS && pair_in) {
                        const int hb = j & 1;
                        const int y = ya - 3 * RR + 2 * j + hrow;                // centre row of this window
                        double ab[2][2] = { { 0.0, 0.0 }, { 0.0, 0.0 } };
                        if (y >= 0 && y < H && px0) {
                            double fq[2], lq[2], cq[2];
#pragma unroll
                            for (int q = 0; q < 2; q++) {
                                const v3d_f64x2* d = reinterpret_cast<const v3d_f64x2*>(&sV1[hb][hrow][q][0]) + (hq - RR / 2);
                                double f = 0.0, l = 0.0, c = 0.0;
#pragma unroll
                                for (int ch = 0; ch <= RR; ch += GF_CH) {
                                    v3d_f64x2 w[GF_CH];
#pragma unroll
                                    for (int i = 0; i < GF_CH; i++) if (ch + i <= RR) w[i] = d[ch + i];
#pragma unroll
                                    for (int i = 0; i < GF_CH; i++) if (ch + i <= RR) {
                                        if (ch + i == 0) { f = w[i].x; c = w[i].y; }
                                        else if (ch + i == RR) { c += w[i].x; l = w[i].y; }
                                        else { c += w[i].x; c += w[i].y; }
                                    }
                                    asm volatile("" : "+v"(c) :: "memory");
                                }
                                fq[q] = f; lq[q] = l; cq[q] = c;
                            }
                            int g0 = 0, gg0 = 0, gl = 0, ggl = 0, cg = 0, cgg = 0;
                            {
                                const int4* gi = reinterpret_cast<const int4*>(&sVi[hb][hrow][0]) + (hq - RR / 2);
#pragma unroll
                                for (int ch = 0; ch <= RR; ch += 3) {
                                    int4 u[3];
#pragma unroll
                                    for (int i = 0; i < 3; i++) if (ch + i <= RR) u[i] = gi[ch + i];
#pragma unroll
                                    for (int i = 0; i < 3; i++) if (ch + i <= RR) {
                                        if (ch + i == 0) { g0 = u[i].x; gg0 = u[i].y; cg = u[i].z; cgg = u[i].w; }
                                        else if (ch + i == RR) { cg += u[i].x; cgg += u[i].y; gl = u[i].z; ggl = u[i].w; }
                                        else { cg += u[i].x + u[i].z; cgg += u[i].y + u[i].w; }
                                    }
                                    asm volatile("" : "+v"(cg), "+v"(cgg) :: "memory");
                                }
                            }
                            const double s0a = cq[0] + fq[0], s0b = cq[0] + lq[0], s1a = cq[1] + fq[1], s1b = cq[1] + lq[1];
                            const int cy = min(y + RR, H - 1) - max(y - RR, 0) + 1;
                            const int cx0 = min(hgx + RR, W - 1) - max(hgx - RR, 0) + 1, cx1 = min(hgx + 1 + RR, W - 1) - max(hgx + 1 - RR, 0) + 1;
                            // away from the image border every window holds (2r+1)^2 pixels: a wave whose pixels are all interior
                            // (the common case) skips both reciprocal refinements (~10 f64 instructions per pixel pair)
                            const double inva = gf_rcp((double)(cx0 * cy)), invb = cx1 == cx0 ? inva : gf_rcp((double)(cx1 * cy));
                            const int sga = cg + g0, sgb = cg + gl, sgga = cgg + gg0, sggb = cgg + ggl;
#pragma unroll
                            for (int n = 0; n < 2; n++) {
                                const double inv = n ? invb : inva;
                                const double mI = (double)(n ? sgb : sga) * (inv * (1.0 / 255.0)), mp = (n ? s0b : s0a) * inv;
                                const double mII = (double)(n ? sggb : sgga) * (inv * (1.0 / 65025.0)), mIp = (n ? s1b : s1a) * (inv * (1.0 / 255.0));
                                const double var = fma(-mI, mI, mII), cov = fma(-mI, mp, mIp);
                                const double a = cov * gf_rcp(var + eps);
                                ab[n][0] = a; ab[n][1] = fma(-a, mI, mp);
                            }
                            if (!px1) { ab[1][0] = 0.0; ab[1][1] = 0.0; }
                        }
                        // rows and columns outside the image hand ZERO a/b to stage 2 (its windows count in-image pixels only)
                        const v3d_f64x2 va2 = { ab[0][0], ab[1][0] }, vb2 = { ab[0][1], ab[1][1] };
                        *reinterpret_cast<v3d_f64x2*>(&sAB[hb][hrow][0][2 * hq]) = va2;
                        *reinterpret_cast<v3d_f64x2*>(&sAB[hb][hrow][1][2 * hq]) = vb2;
                    }
                    // ---- V1(p): two input rows enter the column's window ----
                    if (p < NS) {
                        const int tA = 2 * p;
                        const RowIn cur[2] = { nx[0], nx[1] };
                        nx[0] = fetch_row(tA + 2); nx[1] = fetch_row(tA + 3);
                        const bool emit = p >= RR;                               // uniform
#pragma unroll
                        for (int rr = 0; rr < 2; rr++) {
                            const int slot = (2 * sp + rr) % R;                  // compile-time ring slot
                            int gn = 0; double pn = 0.0;
                            if (cur[rr].in) {
                                gn = cur[rr].g;
                                const int e = ya - 2 * RR + tA + rr;
                                const double fy = (e + 0.5) * sy - 0.5, wy = fy - floor(fy);
                                const double a0 = (double)cur[rr].a0, b0 = (double)cur[rr].b0;
                                const double top = fma(bwx, (double)cur[rr].a1 - a0, a0);
                                const double bot = fma(bwx, (double)cur[rr].b1 - b0, b0);
                                pn = fma(wy, bot - top, top);
                            }
                            const int go = (int)((rgw[slot >> 2] >> (8 * (slot & 3))) & 0xFFu); const double po = r1[slot];
                            rgw[slot >> 2] = (rgw[slot >> 2] & ~(0xFFu << (8 * (slot & 3)))) | ((uint32_t)gn << (8 * (slot & 3)));
                            r1[slot] = pn;
                            vg += gn - go; vgg += gn * gn - go * go;
                            v0 += pn - po; v1 = fma((double)gn, pn, fma(-(double)go, po, v1));
                            if (emit) {
                                sV1[p & 1][rr][0][t] = v0; sV1[p & 1][rr][1][t] = v1;
                                sVi[p & 1][rr][t] = make_int2(vg, vgg);
                            }
                        }
                    }
                    __syncthreads();
                }
            }
        }
      }
    } else {
        // ================= stage 2: a, b -> q (k_gfm<2>'s arithmetic) =================
        const bool pair_out = 2 * hq >= 2 * RR && 2 * hq < COLS - 2 * RR;           // output columns of the strip
        const bool px0 = pair_out && hgx < W, px1 = pair_out && hgx + 1 < W;
        const bool vec_ok = px1 && (W & 1) == 0;
        double r0[R], r1[R], va = 0.0, vb = 0.0;
#pragma unroll
        for (int j = 0; j < R; j++) { r0[j] = 0.0; r1[j] = 0.0; }
        for (int p0 = 0; p0 < NP; p0 += R) {
#pragma unroll
            for (int sp = 0; sp < R; sp++) {
                const int p = p0 + sp;
                if (p < NP) {                                                    // uniform
                    // Thread constants are RE-DERIVED from the thread index in every phase (the asm makes the value opaque, so
                    // nothing derived from it can be hoisted out of the loop): with two 34-register rings this role has no
                    // room to keep LDS addresses, window widths and column indices live across phases -- hoisted, they spill.
                    int tl = t;
                    asm volatile("" : "+v"(tl));
                    const int hrow = tl / HP, hq = tl % HP, hgx = gx0 + 2 * hq;
                    // ---- H2(p-3): window sums of the a/b row pair -> q of two pixels ----
                    {
                        const int j = p - 3;
                        const int y = ya - 4 * RR + 2 * j + hrow;
                        if (j >= 2 * RR && j < NS && px0 && y < yb) {
                            const int hb = j & 1;
                            // per plane: the total of the r + 1 aligned pair sums hq - r/2 .. hq + r/2 (columns 2hq - r .. 2hq + r + 1), then
                            // output 0 drops the last column, output 1 the first
                            double wa[2], wb[2];
#pragma unroll
                            for (int q = 0; q < 2; q++) {
                                const double* d = &sP2[hb][hrow][q][hq - RR / 2];
                                const double f = sV2[hb][hrow][q][2 * hq - RR], l = sV2[hb][hrow][q][2 * hq + RR + 1];
                                double c = 0.0;
#pragma unroll
                                for (int ch = 0; ch <= RR; ch += 3) {
                                    double w[3];
#pragma unroll
                                    for (int i = 0; i < 3; i++) if (ch + i <= RR) w[i] = d[ch + i];
#pragma unroll
                                    for (int i = 0; i < 3; i++) if (ch + i <= RR) c += w[i];
                                    asm volatile("" : "+v"(c) :: "memory");
                                }
                                wa[q] = c - l; wb[q] = c - f;
                                asm volatile("" : "+v"(wa[q]), "+v"(wb[q]) :: "memory");
                            }
                            const double s0a = wa[0], s0b = wb[0], s1a = wa[1], s1b = wb[1];
                            const int cy = min(y + RR, H - 1) - max(y - RR, 0) + 1;
                            const int cx0 = min(hgx + RR, W - 1) - max(hgx - RR, 0) + 1, cx1 = min(hgx + 1 + RR, W - 1) - max(hgx + 1 - RR, 0) + 1;
                            const double inva = gf_rcp((double)(cx0 * cy)), invb = cx1 == cx0 ? inva : gf_rcp((double)(cx1 * cy));
                            const size_t o = (size_t)y * W + hgx;
                            const double Ia = (double)guide[o] * (1.0 / 255.0);
                            const double Ib = px1 ? (double)guide[o + 1] * (1.0 / 255.0) : 0.0;
                            const float qa = (float)((s0a * inva) * Ia + (s1a * inva)), qb = (float)((s0b * invb) * Ib + (s1b * invb));
                            if (vec_ok) {
                                v3d_f32x2 vq = { qa, qb };
                                __builtin_nontemporal_store(vq, reinterpret_cast<v3d_f32x2*>(out + o));
                            } else {
                                out[o] = qa;
                                if (px1) out[o + 1] = qb;
                            }
                        }
                    }
                    // ---- V2(p-2): the a/b row pair H1(p-2) left in LDS enters the column's window ----
                    {
                        const int j = p - 2;
                        if (j >= RR && j < NS) {
                            const bool emit = j >= 2 * RR;                       // uniform
#pragma unroll
                            for (int rr = 0; rr < 2; rr++) {
                                const int slot = (2 * sp + rr) % R;              // compile-time ring slot
                                const double n0 = sAB[j & 1][rr][0][tl], n1 = sAB[j & 1][rr][1][tl];
                                const double o0 = r0[slot], o1 = r1[slot];
                                r0[slot] = n0; r1[slot] = n1;
                                va += n0 - o0; vb += n1 - o1;
                                if (emit) {
                                    sV2[j & 1][rr][0][tl] = va; sV2[j & 1][rr][1][tl] = vb;
                                    const double pa = va + gf_partner(va), pb = vb + gf_partner(vb);        // aligned pair sums (both lanes of a pair form the same)
                                    if (!(tl & 1)) { sP2[j & 1][rr][0][tl >> 1] = pa; sP2[j & 1][rr][1][tl >> 1] = pb; }
                                }
                            }
                        }
                    }
                    __syncthreads();
                }
            }
        }
    }
}

template <int RR, typename TD>
static void launch_gff(const TD* depth_lo, int Wlo, int Hlo, const uint8_t* guide, int W, int H, double eps,
                       float* out, int n, size_t depth_stride, size_t guide_stride, hipStream_t st)
{
    const int cols = g_v3d_opt.gf_cols == 512 ? 512 : 256;
    int band = g_v3d_opt.gf_band;
    if (band <= 0) {
        // auto: every band pays 4r warm-up rows and the launch runs in whole "rounds" of the resident workgroups (equal-length
        // workgroups: two per CU at 256 columns, one at 512), so pick the band count that minimises rounds x (band + 4r) --
        // e.g. 34 4K frames on 256 CUs: 8 bands of 270 rows are 9.56 rounds = 10 x 302 row steps, 5 bands of 432 are 5.98 = 6 x 464
        int dev = 0, ncu = 256;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
        const long slots = (long)ncu * (cols == 512 ? 1 : 2), strips = v3d_cdiv(W, cols - 4 * RR);
        long best = -1;
        for (int nb = 1; nb <= 64 && nb <= H; nb++) {
            const int b = (v3d_cdiv(H, nb) + 1) & ~1;
            const long wgs = strips * v3d_cdiv(H, b) * n, rounds = (wgs + slots - 1) / slots, cost = rounds * (b + 4 * RR);
            if (best < 0 || cost < best) { best = cost; band = b; }
        }
    }
    if (g_v3d_opt.gf_cols == 512) {        // 512-column strips, 16 waves, one workgroup per CU: half the strip-halo recompute
        const dim3 grid(v3d_cdiv(W, 512 - 4 * RR), v3d_cdiv(H, band), n);
        hipLaunchKernelGGL((k_gff<RR, 512, TD, false>), grid, dim3(1024), 0, st, depth_lo, Wlo, Hlo, guide, W, H, eps, band, out, depth_stride, guide_stride);
        return;
    }
    const dim3 grid(v3d_cdiv(W, 256 - 4 * RR), v3d_cdiv(H, band), n);
    if constexpr (std::is_same<TD, int16_t>::value) {
        if (g_v3d_opt.gf_int1 && W == 2 * Wlo && H == 2 * Hlo) {          // int16 disparity, exact 2x: stage 1 in exact integers
            hipLaunchKernelGGL((k_gff<RR, 256, TD, true>), grid, dim3(512), 0, st, depth_lo, Wlo, Hlo, guide, W, H, eps, band, out, depth_stride, guide_stride);
            return;
        }
    }
    hipLaunchKernelGGL((k_gff<RR, 256, TD, false>), grid, dim3(512), 0, st, depth_lo, Wlo, Hlo, guide, W, H, eps, band, out, depth_stride, guide_stride);
}

template <int RR, typename TD>
static void launch_gfm(const TD* depth_lo, int Wlo, int Hlo, const uint8_t* guide, int W, int H, double eps,
                       double* A, double* B, float* out, int n, size_t depth_stride, size_t guide_stride, hipStream_t st)
{
    // band heights (measured sweep, 30 x 4K frames): each band pays 2r warm-up rows; sweep 1 is VALU-bound, sweep 2
    // is bound by its re-reads of the f64 a/b planes
    const int band1 = g_v3d_opt.gf_band1, band2 = g_v3d_opt.gf_band2;
    const dim3 grid1(v3d_cdiv(W, 256 - 2 * RR), v3d_cdiv(H, band1), n), grid2(v3d_cdiv(W, 256 - 2 * RR), v3d_cdiv(H, band2), n);
    hipLaunchKernelGGL((k_gfm<1, RR, TD>), grid1, dim3(256), 0, st, depth_lo, Wlo, Hlo, guide, W, H, eps, band1, A, B, out, depth_stride, guide_stride);
    hipLaunchKernelGGL((k_gfm<2, RR, TD>), grid2, dim3(256), 0, st, depth_lo, Wlo, Hlo, guide, W, H, eps, band2, A, B, out, depth_stride, guide_stride);
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

// n frames: frame f at depth_lo + f*depth_stride (elements), guide + f*guide_stride (bytes), out + f*W*H;
// ws must hold n * v3d_guided_upscale_ws_bytes(W, H)
template <typename TD>
static int guided_upscale_batch(const TD* depth_lo, int Wlo, int Hlo, size_t depth_stride, const uint8_t* guide,
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
    if (!g_v3d_opt.gf_tiled && g_v3d_opt.gf_fused && (r == 4 || r == 8)) {
        if (r == 4) launch_gff<4>(depth_lo, Wlo, Hlo, guide, W, H, (double)eps, out, n, depth_stride, guide_stride, st);
        else launch_gff<8>(depth_lo, Wlo, Hlo, guide, W, H, (double)eps, out, n, depth_stride, guide_stride, st);
        V3D_LAUNCH_CHECK();
        return V3D_OK;
    }
    if (!g_v3d_opt.gf_tiled && (r == 4 || r == 8)) {     // larger rings spill: r = 16 takes the tiled kernel
        if (r == 4) launch_gfm<4>(depth_lo, Wlo, Hlo, guide, W, H, (double)eps, A, B, out, n, depth_stride, guide_stride, st);
        else launch_gfm<8>(depth_lo, Wlo, Hlo, guide, W, H, (double)eps, A, B, out, n, depth_stride, guide_stride, st);
        V3D_LAUNCH_CHECK();
        return V3D_OK;
    }
    for (int f = 0; f < n; f++) {
    const TD* depth_lo_f = depth_lo + (size_t)f * depth_stride; const uint8_t* guide_f = guide + (size_t)f * guide_stride;
    double* A = reinterpret_cast<double*>(ws) + (size_t)f * 2 * W * H; double* B = A + (size_t)W * H; float* out_f = out + (size_t)f * W * H;
    const int ty = r <= 8 ? 16 : 8;
    const dim3 grid(v3d_cdiv(W, GF_TX), v3d_cdiv(H, ty));
    const size_t sm1 = gf_smem(r, 4, ty), sm2 = gf_smem(r, 2, ty);
    if (ty == 16) {
        // above the default dynamic-LDS limit: opt in (160 KiB per CU on gfx950)
        V3D_HIP_CHECK(hipFuncSetAttribute((const void*)k_gf<1, 4, TD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm1));
        V3D_HIP_CHECK(hipFuncSetAttribute((const void*)k_gf<2, 4, TD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm2));
        hipLaunchKernelGGL((k_gf<1, 4, TD>), grid, dim3(256), sm1, st, depth_lo_f, Wlo, Hlo, guide_f, W, H, r, (double)eps, A, B, out_f);
        hipLaunchKernelGGL((k_gf<2, 4, TD>), grid, dim3(256), sm2, st, depth_lo_f, Wlo, Hlo, guide_f, W, H, r, (double)eps, A, B, out_f);
    } else {
        V3D_HIP_CHECK(hipFuncSetAttribute((const void*)k_gf<1, 2, TD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm1));
        V3D_HIP_CHECK(hipFuncSetAttribute((const void*)k_gf<2, 2, TD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm2));
        hipLaunchKernelGGL((k_gf<1, 2, TD>), grid, dim3(256), sm1, st, depth_lo_f, Wlo, Hlo, guide_f, W, H, r, (double)eps, A, B, out_f);
        hipLaunchKernelGGL((k_gf<2, 2, TD>), grid, dim3(256), sm2, st, depth_lo_f, Wlo, Hlo, guide_f, W, H, r, (double)eps, A, B, out_f);
    }
    }
    V3D_LAUNCH_CHECK();
    return V3D_OK;
}

extern "C" int v3d_guided_upscale_batch(const float* depth_lo, int Wlo, int Hlo, size_t depth_stride, const uint8_t* guide,
                                        int W, int H, size_t guide_stride, int n, int r, float eps, float* out, void* ws, void* stream)
{
    return guided_upscale_batch<float>(depth_lo, Wlo, Hlo, depth_stride, guide, W, H, guide_stride, n, r, eps, out, ws, stream);
}

extern "C" int v3d_guided_upscale(const float* depth_lo, int Wlo, int Hlo, const uint8_t* guide, int W, int H,
                                  int r, float eps, float* out, void* ws, void* stream)
{
    return guided_upscale_batch<float>(depth_lo, Wlo, Hlo, 0, guide, W, H, 0, 1, r, eps, out, ws, stream);
}

// the same filter fed with the matcher's int16 disparity (x16, <= 0 invalid): depth.py:341 `/16` and :374 `<= 0 -> 0` are
// applied as the values are loaded, so the stereo-only pipeline never writes or re-reads the float32 depth plane
extern "C" int v3d_guided_upscale_disp16_batch(const int16_t* disp16, int Wlo, int Hlo, size_t disp_stride, const uint8_t* guide,
                                               int W, int H, size_t guide_stride, int n, int r, float eps, float* out, void* ws, void* stream)
{
    return guided_upscale_batch<int16_t>(disp16, Wlo, Hlo, disp_stride, guide, W, H, guide_stride, n, r, eps, out, ws, stream);
}
