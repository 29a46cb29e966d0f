// v3d_sgbm.hip -- semi-global block matching for gfx950 (MI355X), D = 64, blockSize = 5.
//
// Replaces cv2.StereoSGBM_create(...).compute(left_gray, right_gray) as invoked at
// reference depth.py:315-325, 341 (OpenCV MODE_SGBM 5-path default, MODE_HH 8-path optional).
// Stage map: a-4 k_prefilter + k_cost; a-5 k_vdd (the three top-down paths in one lock-step
// pass) + k_hfused (both horizontal paths), a-6 the WTA tail of k_hfused; a-7/a-8 k_lrcheck_median + k_ccl_*.
// k_chain<...> (one launch per path direction, WTA tail on the last) is the fallback behind V3D_VDD=0 /
// V3D_HFUSED=0 / v3d_sgbm_set_lockstep(h, 0) and is parity-tested like the default path.
//
// HBM layout (per frame, W1 = W - 64):
//   rec        : uint4 [H][W]        pre-filter records, left then right image: {grad, grad_lo, grad_hi, 0 | raw, raw_lo, raw_hi, 0}
//   C, S       : int16 [H][W1][64]   d fastest: one pixel = one 128-B line
//   wta        : u32 [H][W]          WTA record per pixel (min S, sub-pixel disparity, winning d); the right-view map is
//                                    formed from it inside k_lrcheck_median (LDS min-scatter), never in HBM
#include "v3d_common.h"
#include <vector>
#include <string.h>
#include <type_traits>

// ---- experiment switches (tools/build_variant.sh; every default = the product path) ----
#ifndef V3D_CK_NT
#define V3D_CK_NT 1            // k_hfused's checkpoints written / read with streaming hints: they are re-read ~1 ms later, long after L2 has turned
                               // over, and as plain accesses they evict the C lines consecutive 96-byte pixels share (4.41 -> 4.22 ms per 34 frames)
#endif
#ifndef V3D_C12
#define V3D_C12 1               // 1: the cost volume C is stored as 12 bits per disparity (96 bytes per pixel, C - P2 <= 25 * 93 < 4096); 0: int16
#endif
#ifndef V3D_X_SPLIT
#define V3D_X_SPLIT 0           // 1: k_hfused's phase 1 (left->right scan, checkpoints) as its own launch k_hscan
#endif
// (Round 3's timing proxies -- a 12-bit C in each of the three big kernels, an SGM chain step hosted in k_cost, k_hfused's
//  phase 1 reading an L1- / L2-resident window -- lived here as V3D_X_C12 / V3D_X_COSTCHAIN / V3D_X_P1L2 builds; their numbers
//  are in DESIGN.md and profiles/r03_experiments/, the code in the history (commit "bench: default batch = one lock-step launch").)
// S (aggregated costs): int16 [H][W1][64] -- offsets in ELEMENTS
#define VOL_PX V3D_D                                               // elements between pixel x and x + 1 of a row
__host__ __device__ static inline size_t vol_row(int y, int W1) { return (size_t)y * W1 * V3D_D; }
__host__ __device__ static inline size_t vol_frame(int H, int W1) { return (size_t)H * W1 * V3D_D; }
// C (matching costs): [H][W1] pixels of C_PXB bytes -- offsets in BYTES.  With V3D_C12 a pixel is 64 x 12 bits, disparity d at
// bit 12 d, holding C - P2 (the 5x5 box sum alone: <= 2325); every reader adds P2 back as it unpacks, so the recurrences see the
// same int16 C as before.  A quarter fewer bytes on each of C's four touches (one write, three reads).
#define C_PXB (V3D_C12 ? 96 : 128)
__host__ __device__ static inline size_t c_row(int y, int W1) { return (size_t)y * W1 * C_PXB; }
__host__ __device__ static inline size_t c_frame(int H, int W1) { return (size_t)H * W1 * C_PXB; }

// ------------------------------------------------------------------------------------------------
// a-4 (i): x-Sobel pre-filter + raw plane + Birchfield-Tomasi half-sample intervals, both images.
// ------------------------------------------------------------------------------------------------
// A workgroup owns 252 output columns (+2 halo each side, one thread per column) and marches down a band of rows:
// per row a thread loads ONE byte per image (the row entering the 3-row Sobel window), the x+-1 neighbours come
// from LDS, and the horizontal differences of the two older rows ride along in registers -- 2 byte loads per pixel
// instead of the 14 a one-row-per-workgroup version issues (that one was bound by its VMEM instruction count).
// The records trail the gradients by one row so that both LDS exchanges of a step share a single barrier.
#ifndef PF_BAND
#define PF_BAND 64
#endif
__global__ __launch_bounds__(256) void k_prefilter(const uint8_t* __restrict__ img1, const uint8_t* __restrict__ img2,
                                                   int W, int H, int pitch, size_t frame_stride, int ft,
                                                   uint4* __restrict__ rec)
{
    __shared__ uint8_t sI[2][2][256];              // [step parity][image][column] bytes of the entering row
    __shared__ unsigned short sGR[2][2][256];      // [step parity][image][column] grad | raw << 8 of the row one step back
    const int t = threadIdx.x, f = blockIdx.z;
    const int x = blockIdx.x * 252 - 2 + t;        // bytes valid for all t, gradients for t in [1, 254], records for [2, 253]
    const int xc = min(max(x, 0), W - 1);
    const bool xin = x > 0 && x < W - 1;           // else tab[0] = ft for both planes (OpenCV leaves the border columns at zero gradient)
    const int ya = blockIdx.y * PF_BAND, yb = min(ya + PF_BAND, H);
    const uint8_t* I0 = img1 + f * frame_stride + xc;
    const uint8_t* I1 = img2 + f * frame_stride + xc;
    auto ld = [&](int y) -> uint32_t {             // both images' bytes of row clamp(y), unconditional loads
        const size_t o = (size_t)min(max(y, 0), H - 1) * pitch;
        return (uint32_t)I0[o] | ((uint32_t)I1[o] << 8);
    };
    // rows clamp(ya-1) and ya prime the window; their horizontal differences need an exchange each
    int dm[2], d0[2], r0[2];                       // dh(row y-1), dh(row y), raw(row y) per image
    {
        const uint32_t vm = ld(ya - 1), v0 = ld(ya);
        sI[0][0][t] = (uint8_t)vm; sI[0][1][t] = (uint8_t)(vm >> 8);
        sI[1][0][t] = (uint8_t)v0; sI[1][1][t] = (uint8_t)(v0 >> 8);
        __syncthreads();
        const int tl = max(t - 1, 0), tr = min(t + 1, 255);
#pragma unroll
        for (int im = 0; im < 2; im++) {
            dm[im] = (int)sI[0][im][tr] - (int)sI[0][im][tl];
            d0[im] = (int)sI[1][im][tr] - (int)sI[1][im][tl];
            r0[im] = (v0 >> (8 * im)) & 0xFF;
        }
        __syncthreads();
    }
    uint32_t nxt = ld(ya + 1);
    int gp[2] = { 0, 0 }, rp[2] = { 0, 0 };         // grad / raw of the previous row (whose record is still owed)
    const int tl = max(t - 1, 0), tr = min(t + 1, 255);
    for (int y = ya; y <= yb; y++) {                // one extra step flushes the last row's record
        const int par = y & 1;
        const uint32_t ve = nxt;                    // row clamp(y+1)
        nxt = ld(y + 2);
        sI[par][0][t] = (uint8_t)ve; sI[par][1][t] = (uint8_t)(ve >> 8);
        sGR[par][0][t] = (unsigned short)(gp[0] | (rp[0] << 8));
        sGR[par][1][t] = (unsigned short)(gp[1] | (rp[1] << 8));
        __syncthreads();
        uint32_t out[4];
#pragma unroll
        for (int im = 0; im < 2; im++) {
            // ---- record of row y-1 from its own and its neighbours' (grad, raw) ----
            const int g = gp[im], r = rp[im];
            const int lo = sGR[par][im][tl], hi = sGR[par][im][tr];
            int gl = g, gr = g, rl = r, rr = r;
            if (x > 0) { gl = (g + (lo & 0xFF)) >> 1; rl = (r + (lo >> 8)) >> 1; }
            if (x < W - 1) { gr = (g + (hi & 0xFF)) >> 1; rr = (r + (hi >> 8)) >> 1; }
            const int g0 = min(min(gl, gr), g), g1 = max(max(gl, gr), g);
            const int q0 = min(min(rl, rr), r), q1 = max(max(rl, rr), r);
            out[2 * im] = (uint32_t)g | ((uint32_t)g0 << 8) | ((uint32_t)g1 << 16);
            out[2 * im + 1] = (uint32_t)r | ((uint32_t)q0 << 8) | ((uint32_t)q1 << 16);
            // ---- gradient of row y: 2*dh(y) + dh(y-1) + dh(y+1), rows replicated at the image border ----
            const int de = (int)sI[par][im][tr] - (int)sI[par][im][tl];
            const int dup = y > 0 ? dm[im] : d0[im];                        // row y-1 clamps to row 0
            const int ddn = y < H - 1 ? de : d0[im];                        // row y+1 clamps to row H-1
            gp[im] = xin ? min(max(2 * d0[im] + dup + ddn, -ft), ft) + ft : ft;
            rp[im] = xin ? r0[im] : ft;
            dm[im] = d0[im]; d0[im] = de; r0[im] = (ve >> (8 * im)) & 0xFF;
        }
        if (y > ya && t >= 2 && t <= 253 && x < W)
            st_stream(rec + ((size_t)f * H + (y - 1)) * W + x, make_uint4(out[0], out[1], out[2], out[3]));   // 1.1 GB per launch, read by the NEXT kernel: streaming (0.353 -> 0.325 ms)
    }
}

// ------------------------------------------------------------------------------------------------
// a-4 (ii,iii): BT pixel cost + 5x5 box sum -> C.  One workgroup = a strip of 60 output columns
// (+2 halo each side) marching down a band of rows; lane = (column, 8 disparities).
// Per row: BT cost bytes -> LDS row, 5-tap horizontal sum from LDS, 5-row vertical sum in registers.
// ------------------------------------------------------------------------------------------------
template <int NP, int LPP>
__device__ __forceinline__ uint32_t chain_step(const uint32_t (&p)[NP], uint32_t delta, const uint32_t (&c)[NP],
                                               uint32_t (&L)[NP], uint32_t P1pk, uint32_t P2pk, bool first_lane, bool last_lane);

// strip geometry for LPC lanes per column (each lane owns 64/LPC disparities): 512 threads = 512/LPC columns,
// two halo columns each side; the right image needs 64 more staged records than the left
template <int LPC> struct CostGeo {
    static constexpr int EP = 64 / LPC, NP = EP / 2, COLS = 512 / LPC, OUT = COLS - 4, NREC = COLS + 64;
};
#ifndef V3D_COST_LPC
#define V3D_COST_LPC 8
#endif

__device__ __forceinline__ uint32_t bt_pair(uint32_t U, uint32_t U0, uint32_t U1, uint32_t V, uint32_t V0, uint32_t V1)
{
    // min(max(0, u - v1, v0 - u), max(0, v - u1, u0 - v)) on two disparities at once; all operands are 0..255, so
    // unsigned saturating subtracts give the max(0, .) for free: 7 packed ops
    const uint32_t a = pk_max(pk_subu_sat(U, V1), pk_subu_sat(V0, U));
    const uint32_t b = pk_max(pk_subu_sat(V, U1), pk_subu_sat(U0, V));
    return pk_min(a, b);
}

template <int DPL> struct VecT;
template <> struct VecT<8> { typedef uint4 type; };
template <> struct VecT<4> { typedef uint2 type; };
template <int NP> __device__ __forceinline__ void vec_unpack(const uint4& v, uint32_t (&r)[NP]) { r[0] = v.x; r[1] = v.y; r[2] = v.z; r[3] = v.w; }
template <int NP> __device__ __forceinline__ void vec_unpack(const uint2& v, uint32_t (&r)[NP]) { r[0] = v.x; r[1] = v.y; }
__device__ __forceinline__ void vec_repack(uint4& v, const uint32_t (&r)[4]) { v = make_uint4(r[0], r[1], r[2], r[3]); }
__device__ __forceinline__ void vec_repack(uint2& v, const uint32_t (&r)[4]) { v = make_uint2(r[0], r[1]); }
__device__ __forceinline__ uint4 vec_pack4(const uint32_t (&r)[4]) { return make_uint4(r[0], r[1], r[2], r[3]); }
__device__ __forceinline__ uint2 vec_pack2(const uint32_t (&r)[2]) { return make_uint2(r[0], r[1]); }
template <int NP> struct Packer;
template <> struct Packer<4> { static __device__ __forceinline__ uint4 go(const uint32_t (&r)[4]) { return vec_pack4(r); } };
template <> struct Packer<2> { static __device__ __forceinline__ uint2 go(const uint32_t (&r)[2]) { return vec_pack2(r); } };

// ---- a lane's view of C: DPL disparities of one pixel.  CRaw is what it fetches, c_unpack turns it into DPL/2 packed int16 pairs ----
typedef uint32_t v3d_u32x2_a4 __attribute__((ext_vector_type(2), aligned(4)));
typedef uint32_t v3d_u32x3_a4 __attribute__((ext_vector_type(3), aligned(4)));
template <int DPL> struct CRaw;
#if V3D_C12
template <> struct CRaw<8> { typedef v3d_u32x3_a4 type; };      // the lane's 8 x 12 bits: 12 bytes at byte 12 dl of the pixel
template <> struct CRaw<4> { typedef v3d_u32x2_a4 type; };      // 8 bytes from the dword boundary at or below byte 6 dl: the lane's 48 bits start at bit (dl & 1) * 16
template <int DPL> __device__ __forceinline__ int c_lane_off(int dl) { return DPL == 8 ? 12 * dl : (6 * dl) & ~3; }
__device__ __forceinline__ uint32_t unpack12_pair(uint32_t t) { return (t & 0xFFFu) | ((t << 4) & 0x0FFF0000u); }     // bits 0-11 | 12-23 -> two halves
__device__ __forceinline__ uint4 c_unpack(const v3d_u32x3_a4& v, int, uint32_t P2pk)
{
    return make_uint4(pk_add(unpack12_pair(v.x), P2pk), pk_add(unpack12_pair(alignbit(v.y, v.x, 24)), P2pk),
                      pk_add(unpack12_pair(alignbit(v.z, v.y, 16)), P2pk), pk_add(unpack12_pair(v.z >> 8), P2pk));
}
__device__ __forceinline__ uint2 c_unpack(const v3d_u32x2_a4& v, int dl, uint32_t P2pk)
{
    const uint32_t sh = (uint32_t)(dl & 1) * 16u;
    const uint32_t lo = alignbit(v.y, v.x, sh), hi = v.y >> sh;                 // the lane's 48 bits: lo, hi[15:0]
    return make_uint2(pk_add(unpack12_pair(lo), P2pk), pk_add(unpack12_pair(alignbit(hi, lo, 24)), P2pk));
}
#else
template <> struct CRaw<8> { typedef v3d_u32x4 type; };
template <> struct CRaw<4> { typedef v3d_u32x2 type; };
template <int DPL> __device__ __forceinline__ int c_lane_off(int dl) { return dl * DPL * 2; }
__device__ __forceinline__ uint4 c_unpack(const v3d_u32x4& v, int, uint32_t) { return make_uint4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ uint2 c_unpack(const v3d_u32x2& v, int, uint32_t) { return make_uint2(v.x, v.y); }
#endif
// streaming (once-read) load of a lane's field; p = pixel base + c_lane_off
// load of a lane's field; p = pixel base + c_lane_off.  STREAM: non-temporal (k_vdd: every line is touched by one load).  With
// 96-byte pixels k_hfused's consecutive pixel loads share cache lines (pixel k starts at 96 k): there the plain load keeps the
// line in L1 for the next pixel (measured: 4.52 -> 4.30 ms per 34 frames; the same switch costs k_vdd 1.5 %)
template <int DPL, bool STREAM> __device__ __forceinline__ typename CRaw<DPL>::type c_load(const unsigned char* p)
{
    if (V3D_NT && STREAM) return __builtin_nontemporal_load(reinterpret_cast<const typename CRaw<DPL>::type*>(p));
    return *reinterpret_cast<const typename CRaw<DPL>::type*>(p);
}

// (Tried and dropped: letting the vertical SGM path ride along in this kernel -- the lane layout is k_chain's,
// but the unbanded kernel it needs has too few waves to gain anything.)
template <int LPC>
#ifndef V3D_COST_WAVES
#define V3D_COST_WAVES 6
#endif
__global__ __launch_bounds__(512, V3D_COST_WAVES) void k_cost(const uint4* __restrict__ rec,
                                              int W, int H, int W1, int band_h, int P2, unsigned char* __restrict__ C, int xcd_order)
{
    typedef CostGeo<LPC> G;
    constexpr int EP = G::EP, NP = G::NP, COLS = G::COLS, OUT = G::OUT, NREC = G::NREC;
    typedef typename VecT<EP>::type vec_t;
    // right-image planes of one row, per quantity, as REVERSED u16 arrays (index grows with d) in two
    // alignments (copy 1 is copy 0 shifted by one element) so that the packed pair (d, d+1) is always an
    // aligned dword: no byte extraction in the hot loop.  Left-image values are stored pre-broadcast.
    // RCOPY (dwords between the copies) = 3 (mod 4) at LPC 8 / = 0 (mod 4) at LPC 16: the columns of one 32-lane
    // ds_read2_b32 group then fall on distinct LDS bank residues (measured: 48 -> 0 conflict cycles per wave-row).
    constexpr int RROW = (NREC + 4) / 2, RCOPY = 6 * RROW + (LPC == 8 ? 3 : 0), RBUF = 2 * RCOPY;
    __shared__ __attribute__((aligned(8))) uint32_t sRV[2 * RBUF];
    __shared__ __attribute__((aligned(8))) uint32_t sUL[2][COLS][6];
    __shared__ vec_t sPix[2][COLS][LPC];                        // BT cost of EP disparities as packed u16 pairs

    const int tid = threadIdx.x, col = tid / LPC, dq = tid % LPC;
    // XCD-aware tile order (v3d_common.h): neighbouring strips re-read each other's halo records (128 staged columns
    // per 60 outputs); on one XCD those re-reads hit its L2 (k_cost FETCH_SIZE -64 %, 2.2 -> 2.0 ms per 30 frames)
    int bxi = blockIdx.x, byi = blockIdx.y, bzi = blockIdx.z;
    if (xcd_order) xcd_tile(bxi, byi, bzi);
    const int xr0 = bxi * OUT;
    const int ys = byi * band_h, ye = min(ys + band_h, H);
    const int f = bzi;
    const uint32_t* rf = reinterpret_cast<const uint32_t*>(rec + (size_t)f * H * W);   // 4 dwords per pixel
    unsigned char* Cf = C + (size_t)f * c_frame(H, W1);
    static_assert(!V3D_C12 || LPC == 8, "the 12-bit store packs a lane's 8 disparities into one 12-byte field");

    const int xrc = min(max(xr0 - 2 + col, 0), W1 - 1);        // clamped cost-region column of this lane
    // staged record i <-> image column xr0 - 1 + i; reversed element k = NREC-1 - i.  d = EP*dq + j reads record
    // i0 - j with i0 = xrc - xr0 + 65 - EP*dq, i.e. reversed elements k0 + j, k0 = NREC-66 - (xrc - xr0) + EP*dq.
    const int k0 = NREC - 66 - (xrc - xr0) + EP * dq;
    const int rcopy = k0 & 1, rk = k0 - rcopy;                  // even element offset inside copy `rcopy`
    const int rv_off = rcopy * RCOPY + rk / 2;                  // dword offset of this lane's first pair (quantity 0, buffer 0)
    const bool out_col = (col >= 2) && (col < 2 + OUT) && (xr0 - 2 + col < W1);
    const int hc = min(max(col, 2), COLS - 3);                  // centre of the 5-tap window this lane sums
    const int nrows = (ye - ys) + 4;

    // Staging, spread over six of the eight waves (the workgroup moves at the pace of its slowest wave): a thread
    // owns HALF a record (dword 0 = gradient triple, dword 1 = raw triple).  Right image, threads 0..2*NREC-1:
    // record i and its left neighbour i-1 give the packed pair (element k, k+1) of three quantities with one
    // v_perm_b32 each, written as ONE dword to copy (k & 1) -- together the threads fill both copies.  Left
    // image, threads 256..256+2*COLS-1: three pre-broadcast dwords.  Records are fetched two rows ahead of
    // their use so the wait for row k+1's record can leave the youngest loads and the C stores of the last rows
    // in flight (vmcnt counts stores too on CDNA).
    // (A dedicated 9th staging wave was tried: 576-thread blocks drop a workgroup per CU and lose.)
    const int half = tid & 1, ri = tid >> 1, lt = (tid - 256) >> 1;
    const bool ld_right = ri < NREC, ld_left = tid >= 256 && lt < COLS;
    uint32_t ld_a = 0, ld_b = 0;                                // dword offsets inside a record row (uniform row base + these)
    int st_off = 0;                                             // dword offset of this thread's staging writes (buffer 0)
    if (ld_right) {
        ld_a = 4 * min(max(xr0 - 1 + ri, 0), W - 1) + 2 + half;
        ld_b = 4 * min(max(xr0 - 2 + ri, 0), W - 1) + 2 + half;
        const int k = NREC - 1 - ri;
        st_off = (k & 1) * RCOPY + 3 * half * RROW + (k >> 1);
    } else if (ld_left) {
        ld_a = ld_b = 4 * (min(max(xr0 - 2 + lt, 0), W1 - 1) + V3D_D) + half;
        st_off = lt * 6 + 3 * half;
    }
    const bool ld_any = ld_right || ld_left;
    auto stage = [&](int b, uint2 rec) {                        // rec.x = own half-record, rec.y = left neighbour's
        if (ld_right) {
#pragma unroll
            for (int j = 0; j < 3; j++)
                sRV[b * RBUF + st_off + j * RROW] = __builtin_amdgcn_perm(rec.y, rec.x, 0x0c000c00u | (uint32_t)j | ((uint32_t)(4 + j) << 16));
        } else if (ld_left) {
#pragma unroll
            for (int j = 0; j < 3; j++)
                (&sUL[b][0][0])[st_off + j] = __builtin_amdgcn_perm(rec.x, rec.x, 0x0c000c00u | (uint32_t)j | ((uint32_t)j << 16));
        }
    };
    // every VMEM instruction of the row loop is issued unconditionally (v3d_common.h: raw buffer access): threads
    // that stage nothing, halo columns and the warm-up rows are switched off through an out-of-range offset
    const __amdgpu_buffer_rsrc_t rs_rec = buf_rsrc(rf, (uint32_t)H * W * 16u), rs_c = buf_rsrc(Cf, (uint32_t)c_frame(H, W1));
    const uint32_t la = ld_any ? ld_a * 4u : V3D_BUF_OOB, lb = ld_any ? ld_b * 4u : V3D_BUF_OOB;   // + row offset < 2^31: bit 31 survives
    auto fetch = [&](int k) -> uint2 {
        const uint32_t ro = (uint32_t)min(max(ys - 2 + min(k, nrows - 1), 0), H - 1) * W * 16u;
        return make_uint2(buf_load_u32(rs_rec, ro + la), buf_load_u32(rs_rec, ro + lb));
    };
    if (ld_any) stage(0, fetch(0));
    // records in flight: nr[p] holds the row whose index has parity p; a slot is refilled (row + 2) right after the
    // stage that consumed it, so both are statically indexed and each load has two row times to land
    uint2 nr[2];
    nr[1] = fetch(1); nr[0] = fetch(2);
    __syncthreads();

    uint32_t ring[5][NP], vs[NP];                               // last five rows' horizontal sums + their running sum
#pragma unroll
    for (int j = 0; j < NP; j++) { vs[j] = V3D_C12 ? 0u : pk_bcast(P2);    // int16 C: P2 rides in the running sum (C = P2 + box sum); 12-bit C: the box sum alone
#pragma unroll
        for (int i = 0; i < 5; i++) ring[i][j] = 0u; }
    // C store offsets: per-thread part (out-of-range marker for halo columns) + uniform row part
    const uint32_t st_col = out_col ? (uint32_t)((xr0 - 2 + col) * C_PXB + c_lane_off<EP>(dq)) : V3D_BUF_OOB;

    for (int k10 = 0; k10 < nrows; k10 += 10) {
#pragma unroll
      for (int s10 = 0; s10 < 10; s10++) {                      // ring slot and LDS buffer are compile-time: no register
        const int k = k10 + s10;                                // shifts, every LDS address is base + immediate
        if (k >= nrows) break;                                  // uniform
        const int slot = s10 % 5, buf = s10 & 1;

        // ---- BT cost of (xrc, d = EP*dq .. +EP-1) on row clamp(ys - 2 + k): quantities g, g_lo, g_hi, r, r_lo, r_hi ----
        // the two planes (gradient, raw) one after the other, fenced: all 6 x NP right-image dwords in flight at once
        // cost a dozen more registers than the 80 that three workgroups per CU leave
        uint32_t pix[NP];
#pragma unroll
        for (int pl = 0; pl < 2; pl++) {
            uint32_t U[3], V[3][NP];
#pragma unroll
            for (int i = 0; i < 3; i++) {
                U[i] = sUL[buf][col][3 * pl + i];
                const uint32_t* pr = &sRV[buf * RBUF + rv_off + (3 * pl + i) * RROW];
#pragma unroll
                for (int j = 0; j < NP; j++) V[i][j] = pr[j];
            }
#pragma unroll
            for (int j = 0; j < NP; j++) {
                const uint32_t c = bt_pair(U[0], U[1], U[2], V[0][j], V[1][j], V[2][j]);
                pix[j] = pl == 0 ? c : pix[j] + pk_shr_u(c, 2);     // gradient + raw / 4; each half <= 93
            }
            if (pl == 0) { if (NP == 4) asm volatile("" : "+v"(pix[0]), "+v"(pix[1]), "+v"(pix[NP - 2]), "+v"(pix[NP - 1]) :: "memory");
                           else asm volatile("" : "+v"(pix[0]), "+v"(pix[NP - 1]) :: "memory"); }
        }
        sPix[buf][col][dq] = Packer<NP>::go(pix);

        if (k + 1 < nrows && ld_any) stage(buf ^ 1, nr[buf ^ 1]);     // row k+1 has parity buf^1 (k10 is even)
        nr[buf ^ 1] = fetch(k + 3);
        __syncthreads();

        // ---- 5-tap horizontal sum on packed u16 pairs, 5-row vertical running sum ----
        {
            uint32_t h[NP], w[NP];
            vec_unpack<NP>(sPix[buf][hc - 2][dq], h);               // (halo lanes re-sum a neighbour's window; never stored)
#pragma unroll
            for (int t = -1; t <= 2; t++) {
                vec_unpack<NP>(sPix[buf][hc + t][dq], w);
#pragma unroll
                for (int j = 0; j < NP; j++) h[j] += w[j];      // halves <= 5 * 189: no carry
            }
#pragma unroll
            for (int j = 0; j < NP; j++) { vs[j] += h[j] - ring[slot][j]; ring[slot][j] = h[j]; }   // add row k, drop row k - 5
            const uint32_t st_row = k >= 4 ? (uint32_t)c_row(ys + k - 4, W1) : V3D_BUF_OOB;     // uniform
            const uint32_t st_off = __builtin_elementwise_add_sat(st_col, st_row);                        // saturating: marker + marker stays out of range
#if V3D_C12
            {   // 8 x 12 bits -> three dwords, one 12-byte store per lane (a wave's store covers 8 whole pixels: 768 contiguous bytes)
                uint32_t t[NP];
#pragma unroll
                for (int j = 0; j < NP; j++)                                                           // halves < 4096: 24 bits per pair = (vs & 0xFFF) | (vs >> 4 & ~0xFFF):
                    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(t[j]) : "s"(0xFFFu), "v"(vs[j]), "v"(vs[j] >> 4));   //   one shift + one bit-field insert (the compiler's own form takes three ops)
                // four 3-byte values -> three dwords, a v_perm_b32 each
                const v3d_u32x3_a4 pk = { __builtin_amdgcn_perm(t[1], t[0], 0x04020100u), __builtin_amdgcn_perm(t[2], t[1], 0x05040201u),
                                          __builtin_amdgcn_perm(t[NP - 1], t[2], 0x06050402u) };
                __builtin_amdgcn_raw_buffer_store_b96(pk, rs_c, st_off, 0, V3D_NT ? 2 : 0);
            }
#else
            buf_store_stream(rs_c, st_off, Packer<NP>::go(vs));
#endif
        }
      }
    }
}

// ------------------------------------------------------------------------------------------------
// a-5 / a-6: one SGM path direction per launch.  A "chain" is one scanline of the direction
// (a row, a column or a diagonal); a wave runs DPL adjacent chains in lock-step:
// LPP = 64/DPL lanes per pixel, each lane holding DPL consecutive disparities as DPL/2 packed
// int16 pairs.  d+-1 neighbours come from v_alignbit + one DPP row shift each way, the min over
// d from packed mins + a DPP butterfly inside the pixel's lane group: no LDS in the recurrence.
//   MODE 0: S  = L          (first direction)
//   MODE 1: S += L (sat)    (middle directions)
//   MODE 2: S + L -> LDS -> winner-take-all / uniqueness / sub-pixel / right-view keys (last direction)
// ------------------------------------------------------------------------------------------------
struct ChainArgs {
    const unsigned char* C; int16_t* S;
    int W1, H, W, nframes;
    int P1, P2;
    int uniq;                 // uniquenessRatio
    uint32_t* wta;            // MODE 2: [nframes][H][W] WTA records (wta_word); columns < 64 are never written
    int xcd;                  // k_hfused: XCD-contiguous row-group order (V3D_HF_XCD=1).  Measured 4 % slower: off
    int persist;              // k_hfused: 0 = one wave per row group; 1 = the resident number of waves draws row groups from `ticket`
    int* ticket;
};

// minimum of both halves of `mn` over the LPP lanes of a pixel, returned in BOTH halves.  One v_pk_min_u16 with op_sel
// swaps the halves against each other (lo = min(lo, hi), hi = min(hi, lo)); a word with equal halves orders like its
// half as an unsigned 32-bit number, so each butterfly step is ONE v_min_u32 with a DPP operand (packed VOP3P ops cannot
// take DPP) and the result needs no re-broadcast.  Costs are non-negative 15-bit values.
template <int LPP>
__device__ __forceinline__ uint32_t pk_hmin_lanes(uint32_t mn)
{
    uint32_t m1;
    asm("v_pk_min_u16 %0, %1, %1 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(m1) : "v"(mn));
    m1 = min(m1, dpp_xchg<V3D_DPP_QUAD(1, 0, 3, 2)>(m1));
    m1 = min(m1, dpp_xchg<V3D_DPP_QUAD(2, 3, 0, 1)>(m1));
    if (LPP >= 8) m1 = min(m1, dpp_xchg<V3D_DPP_ROW_HALF_MIRROR>(m1));
    if (LPP >= 16) m1 = min(m1, dpp_xchg<V3D_DPP_ROW_MIRROR>(m1));
    return m1;
}

// L[d] = C[d] + min(Lp[d], Lp[d-1]+P1, Lp[d+1]+P1, delta) - delta ; returns delta' = min_d L[d] + P2 (both halves)
template <int NP, int LPP>
__device__ __forceinline__ uint32_t chain_step(const uint32_t (&p)[NP], uint32_t delta, const uint32_t (&c)[NP],
                                               uint32_t (&L)[NP], uint32_t P1pk, uint32_t P2pk, bool first_lane, bool last_lane)
{
    const uint32_t MAXPK = 0x7FFF7FFFu;
    // d-1 / d+1 across the lanes of a pixel: one v_or_b32 with a DPP operand each.  The pixel's edge lanes OR the
    // out-of-range fill in (costs are 15-bit, so x | 0x7FFF7FFF is the fill whatever the shift delivered: the neighbour
    // pixel's lane, or 0 from bound_ctrl where the DPP row ends); the masks are loop-invariant registers.
    const uint32_t fill_prev = first_lane ? MAXPK : 0u, fill_next = last_lane ? MAXPK : 0u;
    const uint32_t prev = dpp_xchg<V3D_DPP_ROW_SHR(1)>(p[NP - 1]) | fill_prev;
    const uint32_t next = dpp_xchg<V3D_DPP_ROW_SHL(1)>(p[0]) | fill_next;
    uint32_t m[NP + 1];
    m[0] = alignbit(p[0], prev, 16);
#pragma unroll
    for (int i = 1; i < NP; i++) m[i] = alignbit(p[i], p[i - 1], 16);
    m[NP] = alignbit(next, p[NP - 1], 16);
    uint32_t mn = MAXPK;
#pragma unroll
    for (int i = 0; i < NP; i++) {
        // C - delta + min(p, n, delta) = C - max(delta - min(p, n), 0): the clamp is the unsigned saturating subtract's
        // (5 packed ops per pair instead of 6; delta - min(p, n) <= P2 <= C, all operands in [0, 32767))
        uint32_t n = pk_add(pk_min(m[i], m[i + 1]), P1pk);
        L[i] = pk_sub(c[i], pk_subu_sat(delta, pk_min(p[i], n)));
        mn = pk_min(mn, L[i]);
    }
    return pk_add(pk_hmin_lanes<LPP>(mn), P2pk);
}

// delta = min_d L[d] + P2 (both halves) recomputed from a path-state vector: lets checkpoints drop the delta word
template <int NP, int LPP>
__device__ __forceinline__ uint32_t chain_delta(const uint32_t (&p)[NP], uint32_t P2pk)
{
    uint32_t mn = p[0];
#pragma unroll
    for (int i = 1; i < NP; i++) mn = pk_min(mn, p[i]);
    return pk_add(pk_hmin_lanes<LPP>(mn), P2pk);
}

#define WTA_ROWB 144   // bytes per pixel row in LDS (128 + 16 pad, keeps 16-B alignment)

// WTA record of one cost-region pixel: [31:17] min S (< 32767), [16:6] d16 + 16 (0 = invalid pixel), [5:0] winning d.
// One plain store per pixel; the right-view map is formed from these records in k_lrcheck_median.
__device__ __forceinline__ uint32_t wta_word(int minS, int d16, int best) { return ((uint32_t)minS << 17) | ((uint32_t)(d16 + 16) << 6) | (uint32_t)best; }
__device__ __forceinline__ int wta_d16(uint32_t w) { return (int)((w >> 6) & 0x7FFu) - 16; }

// winner-take-all for one pixel whose 64 aggregated costs sit in LDS (stereosgbm.cpp per-row tail)
__device__ __forceinline__ void wta_pixel(const unsigned char* srow, bool valid, int x, int y, int frame,
                                          const ChainArgs& a)
{
    uint32_t v[32];
#pragma unroll
    for (int q = 0; q < 8; q++) {
        const uint4 t = *reinterpret_cast<const uint4*>(srow + q * 16);
        v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
    }
    // argmin with the lowest d winning ties, in packed 16-bit arithmetic: (1) min S over the 64 halves; (2) keys
    // (S - minS) * 64 + d with saturation (only keys < 64, i.e. S == minS, can win) and their packed minimum.
    // 4 packed ops per two disparities instead of 6 scalar ones.
    uint32_t mpk = v[0];
#pragma unroll
    for (int i = 1; i < 32; i++) mpk = pk_minu(mpk, v[i]);
    const int minS = (int)min(mpk & 0xFFFFu, mpk >> 16);
    const uint32_t minpk = pk_bcast(minS), k64 = 0x00400040u;
    uint32_t kacc = 0xFFFFFFFFu;
#pragma unroll
    for (int i = 0; i < 32; i++) {
        uint32_t key;
        const uint32_t t = pk_subu_sat(v[i], minpk), dc = (uint32_t)(2 * i) | ((uint32_t)(2 * i + 1) << 16);
        asm("v_pk_mad_u16 %0, %1, %2, %3 clamp" : "=v"(key) : "v"(t), "v"(k64), "s"(dc));
        kacc = pk_minu(kacc, key);
    }
    const int best = (int)(min(kacc & 0xFFFFu, kacc >> 16) & 63u);
    // uniqueness: reject iff exists d, |d-best| > 1, S[d]*(100-uniq) < minS*100  <=>  S[d] < T1
    const int uq = 100 - a.uniq, thr = minS * 100;
    int T1 = uq > 0 ? (thr + uq - 1) / uq : (thr > 0 ? 32768 : 0);
    T1 = min(T1, 32768);
    const uint32_t T1pk = pk_bcast(T1), one = 0x00010001u;
    uint32_t cntpk = 0;
#pragma unroll
    for (int i = 0; i < 32; i++) cntpk += pk_minu(pk_subu_sat(T1pk, v[i]), one);
    const int cnt = (int)(cntpk & 0xFFFFu) + (int)(cntpk >> 16);
    const unsigned short* s16 = reinterpret_cast<const unsigned short*>(srow);
    const int sm = best > 0 ? (int)s16[best - 1] : 0, sp = best < V3D_D - 1 ? (int)s16[best + 1] : 0;
    int cw = (minS < T1) ? 1 : 0;
    if (best > 0 && sm < T1) cw++;
    if (best < V3D_D - 1 && sp < T1) cw++;
    const bool ok = valid && (minS < V3D_MAX_COST) && (cnt <= cw);
    if (!valid) return;
    uint32_t word = 0u;                                        // invalid
    if (ok) {
        int d16 = best * 16;
        if (best > 0 && best < V3D_D - 1) {
            const int den = max(sm + sp - 2 * minS, 1);
            d16 += ((sm - sp) * 16 + den) / (den * 2);
        }
        word = wta_word(minS, d16, best);
    }
    a.wta[((size_t)frame * a.H + y) * a.W + x + V3D_D] = word;
}

template <bool HORIZ, int XS, bool YREV, int MODE, int DPL>
__global__ __launch_bounds__(256) void k_chain(ChainArgs a)
{
    constexpr int NP = DPL / 2, LPP = 64 / DPL, PPW = DPL;     // chains (pixels) per wave = 64 / LPP
    constexpr int PF = 4;                                     // prefetch depth (steps)
    constexpr int BS = 64 / PPW;                              // MODE 2: steps per WTA batch
    typedef typename VecT<DPL>::type Vec;
    static_assert(MODE != 2 || HORIZ, "the WTA tail rides on a horizontal direction");

    __shared__ __attribute__((aligned(16))) unsigned char sS[MODE == 2 ? 4 * 64 * WTA_ROWB : 16];

    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    const int W1 = a.W1, H = a.H;
    const int NC = HORIZ ? H : (XS == 0 ? W1 : W1 + H - 1);
    const int groups = (NC + PPW - 1) / PPW;
    const int gw = blockIdx.x * 4 + wib;
    const int frame = gw / groups, grp = gw - frame * groups;
    if (frame >= a.nframes) return;                           // wave-uniform; no block-wide barriers below

    const int sub = lane / LPP, dl = lane % LPP;
    const int c0 = grp * PPW, c1 = min(c0 + PPW, NC) - 1;
    const int c = c0 + sub;
    const bool cvalid = c <= c1;
    const int cc = min(c, c1);

    int tlo = 0, thi;
    if (HORIZ) thi = W1;
    else if (XS == 0) thi = H;
    else if (XS > 0) { tlo = max(0, H - 1 - c1); thi = min(H, W1 + H - 1 - c0); }
    else { tlo = max(0, c0 - (W1 - 1)); thi = min(H, c1 + 1); }

    const unsigned char* Cf = a.C + (size_t)frame * c_frame(H, W1) + c_lane_off<DPL>(dl);
    int16_t* Sf = a.S + (size_t)frame * vol_frame(H, W1) + dl * DPL;
    const int x0 = HORIZ ? 0 : (XS == 0 ? cc : (XS > 0 ? cc - (H - 1) : cc));

    auto pos = [&](int t, int& x, int& y) {
        if (HORIZ) { x = XS > 0 ? t : W1 - 1 - t; y = cc; }
        else { y = YREV ? H - 1 - t : t; x = x0 + XS * t; }
    };
    auto pix_off = [&](int t) -> int {                         // pixel index of step t inside the frame
        int x, y; pos(t, x, y);
        x = min(max(x, 0), W1 - 1);
        return y * W1 + x;
    };

    const uint32_t P1pk = pk_bcast(a.P1), P2pk = pk_bcast(a.P2);
    const bool first_lane = dl == 0, last_lane = dl == LPP - 1;

    uint32_t p[NP];
#pragma unroll
    for (int i = 0; i < NP; i++) p[i] = 0;
    uint32_t delta = P2pk;                                    // out-of-image predecessor: L = 0, min = 0

    typename CRaw<DPL>::type cq[PF]; Vec sq[PF];
#pragma unroll
    for (int j = 0; j < PF; j++) {
        const size_t o = (size_t)pix_off(min(tlo + j, thi - 1));
        cq[j] = *reinterpret_cast<const typename CRaw<DPL>::type*>(Cf + o * C_PXB);
        if (MODE != 0) sq[j] = *reinterpret_cast<const Vec*>(Sf + o * VOL_PX);
    }

    unsigned char* myS = sS + (MODE == 2 ? wib * 64 * WTA_ROWB : 0);

    for (int tb = tlo; tb < thi; tb += (MODE == 2 ? BS : PF)) {
#pragma unroll
        for (int jj = 0; jj < (MODE == 2 ? BS : PF); jj++) {
            const int j = jj % PF;
            const int t = tb + jj;
            if (t < thi) {
                uint32_t cv[NP], sv[NP], L[NP];
                vec_unpack<NP>(c_unpack(cq[j], dl, P2pk), cv);
                if (MODE != 0) vec_unpack<NP>(sq[j], sv);
                const size_t o = (size_t)pix_off(t) * VOL_PX;
                {   // refill this queue slot with step t + PF
                    const size_t on = (size_t)pix_off(min(t + PF, thi - 1));
                    cq[j] = *reinterpret_cast<const typename CRaw<DPL>::type*>(Cf + on * C_PXB);
                    if (MODE != 0) sq[j] = *reinterpret_cast<const Vec*>(Sf + on * VOL_PX);
                }
                uint32_t nd = chain_step<NP, LPP>(p, delta, cv, L, P1pk, P2pk, first_lane, last_lane);
                bool active = cvalid;
                if (!HORIZ && XS != 0) {
                    int x, y; pos(t, x, y);
                    active = cvalid && ((unsigned)x < (unsigned)W1);
#pragma unroll
                    for (int i = 0; i < NP; i++) L[i] = active ? L[i] : 0u;
                    nd = active ? nd : P2pk;
                }
#pragma unroll
                for (int i = 0; i < NP; i++) p[i] = L[i];
                delta = nd;
                if (MODE == 0) {
                    if (active) *reinterpret_cast<Vec*>(Sf + o) = Packer<NP>::go(L);
                } else {
#pragma unroll
                    for (int i = 0; i < NP; i++) sv[i] = pk_add_sat(sv[i], L[i]);
                    if (MODE == 1) {
                        if (active) *reinterpret_cast<Vec*>(Sf + o) = Packer<NP>::go(sv);
                    } else {
                        *reinterpret_cast<Vec*>(myS + (sub * BS + jj) * WTA_ROWB + dl * DPL * 2) = Packer<NP>::go(sv);
                    }
                }
            }
        }
        if (MODE == 2) {
            // lane = (chain, step-in-batch): one pixel per lane, all 64 costs read back from LDS
            const int wsub = lane / BS, wj = lane % BS;
            const int t = tb + wj;
            const int y = c0 + wsub;
            const int x = XS > 0 ? t : W1 - 1 - t;
            const bool valid = (y <= c1) && (t < thi);
            wta_pixel(myS + lane * WTA_ROWB, valid, x, y, frame, a);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// a-5/a-6, both horizontal paths + WTA in ONE launch (saves an S write, an S read and a C read per
// frame versus k_chain<H0, mode 1> + k_chain<H4, mode 2>).  The final S needs L_left(x) and L_right(x)
// of the same pixel, but the two recurrences run in opposite directions and a row of L (237 KB) fits
// nowhere on chip.  So: phase 1 sweeps left->right reading only C and drops a CHECKPOINT of the path
// state (DPL/2 + 1 registers per lane) every K pixels into a small global buffer; phase 2 walks the
// K-pixel blocks right->left: restore the checkpoint, recompute L_left for the block into registers,
// run L_right backwards over it, form S + L_left + L_right on chip and do the WTA tail.
// Cost: L_left is computed twice (+1 path of VALU), C is read twice, S once, never written.
// ------------------------------------------------------------------------------------------------
template <int DPL> struct HfC {
    typedef typename CRaw<DPL>::type Raw;
    // pointer to the lane's field of pixel (row, x = 0) and the load of pixel x
    static __device__ __forceinline__ const unsigned char* base(const unsigned char* C, int frame, int H, int W1, int row, int dl)
    {
        return C + (size_t)frame * c_frame(H, W1) + c_row(row, W1) + c_lane_off<DPL>(dl);
    }
    static __device__ __forceinline__ Raw load(const unsigned char* b, int x) { return c_load<DPL, !V3D_C12>(b + (size_t)x * C_PXB); }
};

// ---------------- phase 1: left -> right over blocks 0 .. nblk-2, checkpoint at every block start ----------------
template <int DPL>
__device__ __forceinline__ void hf_phase1(const unsigned char* Crow, uint32_t* ck, int nblk, int dl, uint32_t P1pk, uint32_t P2pk)
{
    constexpr int NP = DPL / 2, LPP = 64 / DPL, K = 64 / DPL;
    const bool first_lane = dl == 0, last_lane = dl == LPP - 1;
    uint32_t p[NP];
#pragma unroll
    for (int i = 0; i < NP; i++) p[i] = 0;
    uint32_t delta = P2pk;
    const int xend = (nblk - 1) * K;                       // the last block is recomputed in phase 2 anyway
    for (int xb = 0; xb < xend; xb += K) {
        // a block's K loads go out back to back: per row stream the DRAM sees one 2-KB burst, not 16 scattered lines
        typename HfC<DPL>::Raw cb[K];
#pragma unroll
        for (int jj = 0; jj < K; jj++) cb[jj] = HfC<DPL>::load(Crow, xb + jj);
#pragma unroll
        for (int jj = 0; jj < K; jj++) {
            uint32_t cv[NP], L[NP];
            vec_unpack<NP>(c_unpack(cb[jj], dl, P2pk), cv);
            delta = chain_step<NP, LPP>(p, delta, cv, L, P1pk, P2pk, first_lane, last_lane);
#pragma unroll
            for (int i = 0; i < NP; i++) p[i] = L[i];
        }
        uint32_t* c = ck + (size_t)(xb / K + 1) * NP * 64;
#pragma unroll
        for (int i = 0; i < NP; i++) { if (V3D_CK_NT) __builtin_nontemporal_store(p[i], c + i * 64); else c[i * 64] = p[i]; }
    }
}

// PH: 3 = both phases in one launch; 2 = phase 2 only (k_hscan has dropped the checkpoints before)
template <int DPL, int PH>
__global__ __launch_bounds__(256, 4) void k_hfused(ChainArgs a, uint32_t* __restrict__ ckpt)      // four waves per SIMD: the 128-VGPR budget
{
    constexpr int NP = DPL / 2, LPP = 64 / DPL, PPW = DPL, K = 64 / PPW;
    typedef typename VecT<DPL>::type Vec;
    __shared__ __attribute__((aligned(16))) unsigned char sS[4 * 64 * WTA_ROWB];

    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    const int W1 = a.W1, H = a.H;
    const int groups = (H + PPW - 1) / PPW, total = groups * a.nframes;
    const int sub = lane / LPP, dl = lane % LPP;
    const int nblk = (W1 + K - 1) / K;
    const uint32_t P1pk = pk_bcast(a.P1), P2pk = pk_bcast(a.P2);
    const bool first_lane = dl == 0, last_lane = dl == LPP - 1;
    unsigned char* myS = sS + wib * 64 * WTA_ROWB;
    // Ticketed form (a.persist): exactly the resident number of waves is launched and each draws (frame, row group) tickets
    // until none is left, instead of one wave per row group: no partly filled last "round" of the 4096 wave slots (34 frames
    // are 2.24 rounds).  Measured per 34 / 68 frames: 4.76 -> 4.56 ms / 9.46 -> 8.98 ms; 30 frames (1.98 rounds): unchanged.
    // (Also measured, round 3: letting the odd waves run their left-to-right scan one group AHEAD, so that both phases are on
    //  the chip at all times instead of all waves streaming, then all waves computing -- 4-7 % SLOWER at every batch size: the
    //  kernel does not suffer from its waves marching in step.)
    auto draw = [&]() -> int {
        int g = 0;
        if (lane == 0) g = atomicAdd(a.ticket, 1);
        return __builtin_amdgcn_readfirstlane(g);
    };
    int work = a.persist ? draw() : (a.xcd ? (int)xcd_linear(blockIdx.x, gridDim.x) : (int)blockIdx.x) * 4 + wib;
    for (;;) {
    if (work >= total) break;                                  // wave-uniform; no block-wide barriers anywhere
    {
    const int frame = work / groups, grp = work - frame * groups;
    const int c0 = grp * PPW, c1 = min(c0 + PPW, H) - 1;
    const int cc = min(c0 + sub, c1);
    const unsigned char* Crow = HfC<DPL>::base(a.C, frame, H, W1, cc, dl);
    const int16_t* Srow = a.S + (size_t)frame * vol_frame(H, W1) + vol_row(cc, W1) + dl * DPL;
    uint32_t* ck = ckpt + ((size_t)frame * groups + grp) * nblk * NP * 64 + lane;         // [blk][reg][lane]; delta is recomputed
    if (PH & 1) hf_phase1<DPL>(Crow, ck, nblk, dl, P1pk, P2pk);

    // ---------------- phase 2: right -> left, block by block ----------------
    uint32_t q[NP];
#pragma unroll
    for (int i = 0; i < NP; i++) q[i] = 0;
    uint32_t qdelta = P2pk;
    for (int blk = nblk - 1; blk >= 0; blk--) {
        const int x0 = blk * K;
        // (prefetching the next block's C/S into a second register set was tried: 182 VGPRs halve the
        //  occupancy and the kernel gets slower, 1.84 -> 2.07 ms per 8 frames; other waves cover the latency.
        //  Round 2, 30 frames, same-box A/B: issuing the next block's loads before this block's WTA tail, folding the
        //  recomputed left path into S at once (no L0 array), prefetching C one further block ahead (all 153 VGPRs, 3
        //  waves per SIMD), double- and triple-buffering phase 1's C blocks (119-122 VGPRs) -- every one of them lands on
        //  the same 4.91-4.95 ms as this form; forced to 128 VGPRs the prefetching forms spill and take 6.6-7.1 ms.
        //  Experiment builds that drop work: no WTA tail 4.57 ms, no phase 1 3.29 ms, neither 2.69 ms (= 5.9 TB/s for
        //  phase 2's C + S + checkpoints).  So phase 2 without the WTA streams at the box's read ceiling, phase 1 adds
        //  its 8.2 GB at ~4.5 TB/s and the WTA tail 0.4-0.6 ms that no amount of load scheduling hides: the kernel is
        //  bound by the memory system (24.5 GB at 5.0 TB/s, VALU 56 % busy), not by latency exposure or occupancy.)
        typename HfC<DPL>::Raw craw[K]; Vec cvv[K], svv[K];
#pragma unroll
        for (int j = 0; j < K; j++) {
            const int xj = min(x0 + j, W1 - 1);
            craw[j] = HfC<DPL>::load(Crow, xj);
            svv[j] = ld_stream(reinterpret_cast<const Vec*>(Srow + (size_t)xj * VOL_PX));
        }
        uint32_t p[NP], delta = P2pk;
#pragma unroll
        for (int i = 0; i < NP; i++) p[i] = 0;
        if (blk > 0) {
            const uint32_t* c = ck + (size_t)blk * NP * 64;
#pragma unroll
            for (int i = 0; i < NP; i++) p[i] = V3D_CK_NT ? __builtin_nontemporal_load(c + i * 64) : c[i * 64];
            delta = chain_delta<NP, LPP>(p, P2pk);
        }
        uint32_t L0[K][NP];
#pragma unroll
        for (int j = 0; j < K; j++) {                          // forward recompute of the left path inside the block
            uint32_t cv[NP];
            cvv[j] = c_unpack(craw[j], dl, P2pk);              // unpacked once, as it arrives; the backward pass re-uses the int16 form
            vec_unpack<NP>(cvv[j], cv);
            delta = chain_step<NP, LPP>(p, delta, cv, L0[j], P1pk, P2pk, first_lane, last_lane);
#pragma unroll
            for (int i = 0; i < NP; i++) p[i] = L0[j][i];
        }
#pragma unroll
        for (int jj = 0; jj < K; jj++) {                       // right path, backwards; x >= W1 only in the last block
            const int j = K - 1 - jj;
            if (x0 + j < W1) {                                  // uniform
                uint32_t cv[NP], sv[NP], L[NP];
                vec_unpack<NP>(cvv[j], cv);
                vec_unpack<NP>(svv[j], sv);
                qdelta = chain_step<NP, LPP>(q, qdelta, cv, L, P1pk, P2pk, first_lane, last_lane);
#pragma unroll
                for (int i = 0; i < NP; i++) { q[i] = L[i]; sv[i] = pk_add_sat(pk_add_sat(sv[i], L0[j][i]), L[i]); }
                *reinterpret_cast<Vec*>(myS + (sub * K + jj) * WTA_ROWB + dl * DPL * 2) = Packer<NP>::go(sv);
            }
        }
        {
            const int wsub = lane / K, wj = lane % K;
            const int x = x0 + K - 1 - wj, y = c0 + wsub;
            wta_pixel(myS + lane * WTA_ROWB, (y <= c1) && (x < W1), x, y, frame, a);
        }
    }
    }
    if (!a.persist) break;
    work = draw();
    }
}

// phase 1 of k_hfused as its own launch: it needs a dozen registers where phase 2 needs 119, so on its own it runs at twice
// the occupancy and keeps twice the bytes in flight per CU
template <int DPL>
__global__ __launch_bounds__(256, 8) void k_hscan(ChainArgs a, uint32_t* __restrict__ ckpt)
{
    constexpr int NP = DPL / 2, LPP = 64 / DPL, PPW = DPL, K = 64 / PPW;
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    const int W1 = a.W1, H = a.H;
    const int groups = (H + PPW - 1) / PPW;
    const int gw = (int)blockIdx.x * 4 + wib;
    const int frame = gw / groups, grp = gw - frame * groups;
    if (frame >= a.nframes) return;
    const int sub = lane / LPP, dl = lane % LPP;
    const int c0 = grp * PPW, c1 = min(c0 + PPW, H) - 1;
    const int cc = min(c0 + sub, c1);
    const int nblk = (W1 + K - 1) / K;
    uint32_t* ck = ckpt + ((size_t)frame * groups + grp) * nblk * NP * 64 + lane;
    hf_phase1<DPL>(HfC<DPL>::base(a.C, frame, H, W1, cc, dl), ck, nblk, dl, pk_bcast(a.P1), pk_bcast(a.P2));
}

// ------------------------------------------------------------------------------------------------
// a-5, the three top-down paths r1 = (x-1, y-1), r2 = (x, y-1), r3 = (x+1, y-1) in ONE pass over C
// (SURVEY 8a-5's K_v): reads C once, writes S = L1 + L2 + L3 once -- 2 volumes instead of the
// 7 that three k_chain launches move.
//
// The diagonals couple neighbouring columns row by row, so a column strip cannot run alone.  Here a
// workgroup (1024 threads = 16 waves, 16 lanes x 4 disparities per pixel) owns a strip of 64 columns of
// one frame and marches down the rows in LOCK-STEP with its two neighbour strips:
//   * inside the strip the previous row's (L1, L3) state is exchanged through LDS (one barrier per row);
//   * across strips the edge columns' state travels through global memory as 8-byte {data, tag} granules
//     (relaxed agent-scope atomic stores / loads: sc1, served by L2, no fences -- MI355X_MICROARCH
//     "handoff-1to1"), tag = (call sequence << 12) | (row + 1), 4-row ring per strip edge (stays in L2).
// The coupling is bidirectional (strip k waits for k-1 AND k+1), so the strips of ONE FRAME must be co-resident;
// frames are independent.  Workgroups are dispatched in blockIdx order (per XCD, each XCD taking every 8th), so the
// resident set is always a prefix of the grid = whole frames plus at most one partial frame per XCD skew, and a
// partial frame merely waits (bounded spin) until finished frames free slots for its remaining strips: a launch
// larger than the chip -- or a chip that has lost slots to another tenant -- slows down instead of dead-locking
// (tests: a 768-workgroup launch on 512 slots).  The host still sizes launches to the occupancy query because a
// waiting partial frame costs a whole extra pass.  Every spin is bounded and trips an error flag instead of hanging.
// (Round 3, measured: drawing (frame, strip) from a ticket at workgroup start -- residency order by construction --
//  scatters neighbour strips over the XCDs and costs 3.38 -> 4.56 ms per 30 frames, like the XCD-contiguous order below;
//  the hardware's own blockIdx -> XCD round-robin, neighbours on adjacent XCDs, is the fast placement.)
// ------------------------------------------------------------------------------------------------
#define VDD_RING 4
#define VDD_GRAN 34                      // granules per edge per row: 32 data dwords + delta (+1 pad)
// poll budget of one lane over the whole pass (every poll round is one L2 round trip, ~1 us): a healthy pass spends one
// to three rounds per row, so 64 per row + slack is two orders of magnitude of headroom and still bounds a pass whose
// neighbours never become resident to ~0.1 s at 1080 rows (it was 2^20 rounds, i.e. seconds)
#define VDD_SPIN_PER_ROW 64
#define VDD_SPIN_SLACK 4096

struct VddArgs {
    const unsigned char* C; int16_t* S;
    int W1, H, nframes, nstrips;
    int P1, P2;
    uint32_t seq;
    int spin_limit;                     // poll rounds a lane may spend waiting over the whole pass
    unsigned long long* gran;           // [frame][strip][2 dirs][VDD_RING][VDD_GRAN]
    int* err;
    int xcd;                            // 1: XCD-contiguous strip order.  Measured slower (3.48 -> 4.58 ms per 30 frames): off
};

// wait for N data granules (+ the delta granule if want_d) of one row: all loads of a poll round go out together
// (one L2 round trip per round, not one per granule); bounded by `budget`
template <int N>
__device__ __forceinline__ bool vdd_poll_n(const unsigned long long* g, const unsigned long long* gd, bool want_d,
                                           uint32_t tag, uint32_t (&v)[N], uint32_t& vd, int& budget)
{
    for (;;) {
        unsigned long long x[N], xd = (unsigned long long)tag << 32;
#pragma unroll
        for (int i = 0; i < N; i++) x[i] = __hip_atomic_load(g + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (want_d) xd = __hip_atomic_load(gd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bool ok = (uint32_t)(xd >> 32) == tag;
#pragma unroll
        for (int i = 0; i < N; i++) ok = ok && ((uint32_t)(x[i] >> 32) == tag);
        if (ok) {
#pragma unroll
            for (int i = 0; i < N; i++) v[i] = (uint32_t)x[i];
            if (want_d) vd = (uint32_t)xd;
            return true;
        }
        if (--budget < 0) return false;
        __builtin_amdgcn_s_sleep(1);
    }
}
__device__ __forceinline__ void vdd_put(unsigned long long* g, uint32_t v, uint32_t tag)
{
    __hip_atomic_store(g, ((unsigned long long)tag << 32) | v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// YREV: the same pass bottom-up (MODE_HH's second half: predecessors (x-1,y+1), (x,y+1), (x+1,y+1)), accumulating
// into the S the top-down pass left behind (S += L1 + L2 + L3, saturating).
template <int DPL, bool YREV>
__global__ __launch_bounds__(1024, 8) void k_vdd(VddArgs a)      // 8 waves/SIMD = two workgroups per CU: the second hides the hand-off latency
{
    constexpr int NP = DPL / 2, LPP = 64 / DPL, PPW = DPL, PXS = 16 * PPW;   // PXS = columns per strip (64 / 128)
#ifndef V3D_VDD_PF8
#define V3D_VDD_PF8 2
#endif
    // C prefetch depth in rows.  The 64-VGPR budget of two workgroups per CU binds at DPL = 8: one row with int16 C (4 registers per
    // row), two rows with the 12-bit C (3 registers per row; measured 3.34 -> 3.23 ms per 34 frames); the bottom-up pass also queues S
    constexpr int PF = DPL == 8 ? (YREV || !V3D_C12 ? 1 : V3D_VDD_PF8) : 4;
    typedef typename VecT<DPL>::type Vec;
    // per-pixel exchanged state: LPP lanes x {L1 (NP dwords), L3 (NP dwords)} + per pixel {delta1, delta3}
    __shared__ Vec sL1[2][PXS + 2][LPP];
    __shared__ Vec sL3[2][PXS + 2][LPP];
    __shared__ uint2 sDl[2][PXS + 2];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);    // wave index as an SGPR: edge-wave branches stay scalar
    const int px = wv * PPW + lane / LPP, dl = lane % LPP;      // pixel inside the strip, disparity group
    const int vb = a.xcd ? (int)xcd_linear(blockIdx.x, gridDim.x) : (int)blockIdx.x;
    const int frame = vb / a.nstrips, strip = vb - frame * a.nstrips;
    const int W1 = a.W1, H = a.H;
    const int x = strip * PXS + px;
    const bool colok = x < W1;
    const bool ragged = __builtin_amdgcn_readfirstlane((strip + 1) * PXS > W1);   // this strip sticks out of the image
    const int xc = min(x, W1 - 1);
    const size_t fbase = (size_t)frame * vol_frame(H, W1);
    const unsigned char* Cp = a.C + (size_t)frame * c_frame(H, W1) + (size_t)xc * C_PXB + c_lane_off<DPL>(dl);
    int16_t* Sp = a.S + fbase + (size_t)xc * VOL_PX + dl * DPL;

    const uint32_t P1pk = pk_bcast(a.P1), P2pk = pk_bcast(a.P2);
    const bool first_lane = dl == 0, last_lane = dl == LPP - 1;
    const bool has_left = strip > 0, has_right = strip + 1 < a.nstrips;
    unsigned long long* gme = a.gran + ((size_t)(frame * a.nstrips + strip) * 2) * VDD_RING * VDD_GRAN;
    const unsigned long long* gleft = a.gran + ((size_t)(frame * a.nstrips + strip - 1) * 2 + 1) * VDD_RING * VDD_GRAN;   // left neighbour, right-going
    const unsigned long long* gright = a.gran + ((size_t)(frame * a.nstrips + strip + 1) * 2 + 0) * VDD_RING * VDD_GRAN;  // right neighbour, left-going
    const bool edge_l = has_left && wv == 0;                      // wave-uniform: this wave talks to the left strip
    const bool edge_r = has_right && wv == 15;                     //               ... to the right strip
    const bool lane_l = lane < LPP, lane_r = lane >= 64 - LPP;     // lanes of the strip's first / last pixel
    int budget = a.spin_limit;
    bool failed = a.spin_limit < 0 && tid == 0;                    // spin_limit -1: test hook, every workgroup reports a time-out

    // row -1: every path starts from the out-of-image state (L = 0, delta = P2)
    {
        uint32_t z[NP];
#pragma unroll
        for (int i = 0; i < NP; i++) z[i] = 0u;
        for (int i = tid; i < 2 * (PXS + 2) * LPP; i += 1024) { (&sL1[0][0][0])[i] = Packer<NP>::go(z); (&sL3[0][0][0])[i] = Packer<NP>::go(z); }
        for (int i = tid; i < 2 * (PXS + 2); i += 1024) (&sDl[0][0])[i] = make_uint2(P2pk, P2pk);
    }
    uint32_t p2[NP], d2 = P2pk;
#pragma unroll
    for (int i = 0; i < NP; i++) p2[i] = 0u;

    typename CRaw<DPL>::type cq[PF];
    auto rowy = [&](int y) -> int { const int yc = min(y, H - 1); return YREV ? H - 1 - yc : yc; };
    auto rowof = [&](int y) -> size_t { return vol_row(rowy(y), W1); };
    auto ld_c = [&](int y) { return c_load<DPL, true>(Cp + c_row(rowy(y), W1)); };
    Vec sq[PF];
#pragma unroll
    for (int j = 0; j < PF; j++) { cq[j] = ld_c(j); if (YREV) sq[j] = ld_stream(reinterpret_cast<const Vec*>(Sp + rowof(j))); }
    __syncthreads();

    // The row loop exists twice: waves that own a strip-edge pixel (wave 0 / wave 15 of an inner strip) carry the
    // poll and publish code, the other 14 run a copy without it -- no merge copies of the polled registers, no
    // branch tests.  Every wave still executes one barrier per row.
    auto rows = [&](auto edge_tag) {
    constexpr bool EDGE = decltype(edge_tag)::value;
    for (int y0 = 0; y0 < H; y0 += PF) {
#pragma unroll
        for (int j = 0; j < PF; j++) {
            const int y = y0 + j;
            if (y < H) {                                           // uniform
                const int prev = (y + 1) & 1, cur = y & 1;         // buffer holding row y-1 / receiving row y
                __syncthreads();                                   // row y-1 of the whole strip is in buffer `prev`
                // ---- 1. predecessors: strip neighbours from LDS; the two edge pixels take theirs from the neighbour
                //         strips' granules, polled AFTER the barrier so the other 14 waves compute meanwhile ----
                uint32_t cv[NP], p1[NP], p3[NP];
                uint32_t sold[NP];
                // (12-bit C: unpacking the NEXT row at the end of this one, off the path between the barrier and the recurrences
                //  the neighbour strips wait for, was measured: 3.32 -> 3.49 ms per 34 frames -- one more row of raw fields and
                //  an unpacked row live across the barrier cost more than the ~18 ops they move)
                vec_unpack<NP>(c_unpack(cq[j], dl, P2pk), cv);
                if (YREV) vec_unpack<NP>(sq[j], sold);
                cq[j] = ld_c(y + PF);
                if (YREV) sq[j] = ld_stream(reinterpret_cast<const Vec*>(Sp + rowof(y + PF)));
                vec_unpack<NP>(sL1[prev][px][dl], p1);             // column x-1 (slot px holds pixel px-1)
                vec_unpack<NP>(sL3[prev][px + 2][dl], p3);         // column x+1
                uint32_t d1 = sDl[prev][px].x, d3 = sDl[prev][px + 2].y;
                if constexpr (EDGE) if (y > 0) {
                    const uint32_t tag = (a.seq << 12) | (uint32_t)y;          // row y-1 carries tag (y-1)+1
                    const int slot = (y - 1) & (VDD_RING - 1);
                    if (edge_l) if (lane_l) {                       // my pixel 0: column x0 - 1 lives in the left strip
                        const unsigned long long* g = gleft + slot * VDD_GRAN;
                        uint32_t vd = P2pk;
                        if (!vdd_poll_n<NP>(g + NP * dl, g + 32, true, tag, p1, vd, budget)) { failed = true; budget = 0; }
                        d1 = vd;
                    }
                    if (edge_r) if (lane_r) {                       // my last pixel: column x0 + PXS lives in the right strip
                        const unsigned long long* g = gright + slot * VDD_GRAN;
                        uint32_t vd = P2pk;
                        if (!vdd_poll_n<NP>(g + NP * dl, g + 32, true, tag, p3, vd, budget)) { failed = true; budget = 0; }
                        d3 = vd;
                    }
                }
                // ---- 2. the two diagonal recurrences first: their edge values are what the neighbour strips wait for ----
                uint32_t L1[NP], L2[NP], L3[NP];
                uint32_t nd1 = chain_step<NP, LPP>(p1, d1, cv, L1, P1pk, P2pk, first_lane, last_lane);
                uint32_t nd3 = chain_step<NP, LPP>(p3, d3, cv, L3, P1pk, P2pk, first_lane, last_lane);
                if (ragged) if (!colok) {                           // columns beyond the image (last strip only; `ragged`
#pragma unroll                                                      //  is uniform, so full strips skip the block): out-of-image state
                    for (int i = 0; i < NP; i++) L1[i] = L3[i] = 0u;
                    nd1 = nd3 = P2pk;
                }
                // ---- 3. publish row y as early as possible: granules for the neighbours, LDS for the strip ----
                if constexpr (EDGE) if (y + 1 < H) {
                    const uint32_t tag = (a.seq << 12) | (uint32_t)(y + 1);
                    const int slot = y & (VDD_RING - 1);
                    if (edge_r) if (lane_r) {                       // my last column's L1 goes right
                        unsigned long long* g = gme + (size_t)(1 * VDD_RING + slot) * VDD_GRAN;
#pragma unroll
                        for (int i = 0; i < NP; i++) vdd_put(g + NP * dl + i, L1[i], tag);
                        if (dl == 0) vdd_put(g + 32, nd1, tag);
                    }
                    if (edge_l) if (lane_l) {                       // my first column's L3 goes left
                        unsigned long long* g = gme + (size_t)(0 * VDD_RING + slot) * VDD_GRAN;
#pragma unroll
                        for (int i = 0; i < NP; i++) vdd_put(g + NP * dl + i, L3[i], tag);
                        if (dl == 0) vdd_put(g + 32, nd3, tag);
                    }
                }
                sL1[cur][px + 1][dl] = Packer<NP>::go(L1);
                sL3[cur][px + 1][dl] = Packer<NP>::go(L3);
                if (dl == 0) sDl[cur][px + 1] = make_uint2(nd1, nd3);
                // ---- 4. the vertical recurrence and the sum ----
                const uint32_t nd2 = chain_step<NP, LPP>(p2, d2, cv, L2, P1pk, P2pk, first_lane, last_lane);
#pragma unroll
                for (int i = 0; i < NP; i++) p2[i] = L2[i];
                d2 = nd2;
                if (colok) {
                    uint32_t o[NP];
#pragma unroll
                    for (int i = 0; i < NP; i++) { o[i] = pk_add_sat(pk_add_sat(L1[i], L2[i]), L3[i]); if (YREV) o[i] = pk_add_sat(o[i], sold[i]); }
                    st_stream(reinterpret_cast<Vec*>(Sp + rowof(y)), Packer<NP>::go(o));
                }

            }
        }
    }
    };
    if (edge_l || edge_r) rows(std::true_type{}); else rows(std::false_type{});
    if (failed) atomicAdd(a.err, 1);
}

// ------------------------------------------------------------------------------------------------
// Lock-step guard: the last launch of a compute call that used k_vdd.  If any strip of this handle has timed out
// since the counter was last cleared, the caller must never consume the disparities: every output pixel becomes
// INVALID and a flag lands in host-visible memory (the next API call on the handle then returns V3D_ERR_LOCKSTEP).
// Healthy path: one dword load per workgroup.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_vdd_guard(const int* __restrict__ err, volatile int* err_host, int16_t* __restrict__ out, size_t n)
{
    const int e = __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (e == 0) return;
    if (blockIdx.x == 0 && threadIdx.x == 0) { *err_host = e; __threadfence_system(); }
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = (int16_t)V3D_INVALID16;
}

// ------------------------------------------------------------------------------------------------
// a-8: medianBlur(3) on int16 with replicated borders (the invalid value takes part like any other).
// ------------------------------------------------------------------------------------------------
#define V3D_SORT2(a, b) { const int _lo = min(a, b), _hi = max(a, b); a = _lo; b = _hi; }
__global__ __launch_bounds__(256) void k_median3x3(const int16_t* __restrict__ src, int W, int H, int16_t* __restrict__ dst)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, f = blockIdx.z;
    if (x >= W) return;
    const int16_t* s = src + (size_t)f * H * W;
    const int xm = max(x - 1, 0), xp = min(x + 1, W - 1);
    const int16_t* r0 = s + (size_t)max(y - 1, 0) * W;
    const int16_t* r1 = s + (size_t)y * W;
    const int16_t* r2 = s + (size_t)min(y + 1, H - 1) * W;
    int p0 = r0[xm], p1 = r0[x], p2 = r0[xp], p3 = r1[xm], p4 = r1[x], p5 = r1[xp], p6 = r2[xm], p7 = r2[x], p8 = r2[xp];
    V3D_SORT2(p1, p2); V3D_SORT2(p4, p5); V3D_SORT2(p7, p8); V3D_SORT2(p0, p1);
    V3D_SORT2(p3, p4); V3D_SORT2(p6, p7); V3D_SORT2(p1, p2); V3D_SORT2(p4, p5);
    V3D_SORT2(p7, p8); V3D_SORT2(p0, p3); V3D_SORT2(p5, p8); V3D_SORT2(p4, p7);
    V3D_SORT2(p3, p6); V3D_SORT2(p1, p4); V3D_SORT2(p2, p5); V3D_SORT2(p4, p7);
    V3D_SORT2(p4, p2); V3D_SORT2(p6, p4); V3D_SORT2(p4, p2);
    dst[(size_t)f * H * W + (size_t)y * W + x] = (int16_t)p4;
}

// ------------------------------------------------------------------------------------------------
// a-6 (right-view map) + a-7 + a-8: disp2, L-R check and 3x3 median in one launch.
//
// The WTA tail leaves ONE 32-bit record per cost-region pixel (wta_word): min S, the sub-pixel disparity and the
// winning d.  OpenCV's right-view map -- disp2[x2] = the d of the cheapest pixel x with x - d == x2, later-processed
// (smaller) x losing ties -- is the minimum of the keys (min S << 6 | 63 - d) over the 64 source pixels x2 .. x2 + 63.
// Rounds 1-2 formed it with a global atomicMin per pixel inside the WTA tail (60 M L2 atomics per 30 frames: 0.3 ms of
// k_hfused's 4.9, measured by a build without them); now a block of this kernel stages the records of 256 source
// columns x 18 rows ONCE (coalesced dword loads instead of two gathers per pixel), min-scatters their keys into an
// LDS row of right-view targets (ds_min_u32), and checks / medians out of LDS.  Same minimum over the same key set:
// bit-identical, schedule-independent.
// Tile: 128 x 16 outputs + a one-pixel ring; sources x0 - 64 .. x0 + 191 (one per thread), targets x0 - 64 .. x0 + 128.
// ------------------------------------------------------------------------------------------------
#define LRM_TX 128
#define LRM_TY 16
#define LRM_NS (LRM_TX + 2 * V3D_D)        // source columns per row  (256 = one per thread)
#define LRM_NT (LRM_TX + V3D_D + 1)        // right-view targets per row
template <bool MED>
__global__ __launch_bounds__(256) void k_lrcheck_median(const uint32_t* __restrict__ wta, int W, int H, int d12, int16_t* __restrict__ out)
{
    static_assert(LRM_NS == 256, "one source column per thread");
    constexpr int NR = LRM_TY + 2;
    __shared__ uint32_t sW[NR][LRM_NS];
    __shared__ uint32_t sD2[NR][LRM_NT + 3];
    __shared__ short sT[NR][LRM_TX + 2];
    const int t = threadIdx.x, f = blockIdx.z;
    const int x0 = blockIdx.x * LRM_TX, y0 = blockIdx.y * LRM_TY;
    const size_t fo = (size_t)f * H * W;
    const int xbase = x0 - V3D_D;                                            // image column of source / target index 0
    // ---- 1. this thread's source column, all rows in flight together (unconditional loads from clamped addresses;
    //         columns left of the cost region were never written: masked below) ----
    const int xs = xbase + t;
    const bool src_in = xs >= V3D_D && xs < W;
    const int xsc = min(max(xs, 0), W - 1);
    uint32_t wv[NR];
#pragma unroll
    for (int r = 0; r < NR; r++) wv[r] = wta[fo + (size_t)min(max(y0 - 1 + r, 0), H - 1) * W + xsc];   // replicated image border
    for (int i = t; i < NR * (LRM_NT + 3); i += 256) (&sD2[0][0])[i] = 0xFFFFFFFFu;
    __syncthreads();
    // ---- 2. records -> LDS, keys -> min-scatter at target x - d ----
#pragma unroll
    for (int r = 0; r < NR; r++) {
        const uint32_t v = src_in ? wv[r] : 0u;
        sW[r][t] = v;
        const int best = (int)(v & 63u), i = t - best;                        // target column xs - best
        if ((v & 0x1FFC0u) != 0u && i >= 0 && i < LRM_NT) atomicMin(&sD2[r][i], ((v >> 17) << 6) | (uint32_t)(63 - best));
    }
    __syncthreads();
    // ---- 3. L-R check of the tile + ring (stereosgbm.cpp: both roundings of the disparity must disagree) ----
    constexpr int NIT = (NR * (LRM_TX + 2) + 255) / 256;
#pragma unroll
    for (int it = 0; it < NIT; it++) {
        const int i = t + 256 * it;
        if (i < NR * (LRM_TX + 2)) {
            const int ty = i / (LRM_TX + 2), tx = i - ty * (LRM_TX + 2);
            const int x = min(max(x0 - 1 + tx, 0), W - 1);
            int d1 = V3D_INVALID16;
            if (x >= V3D_D) {
                d1 = wta_d16(sW[ty][x - xbase]);
                if (d1 != V3D_INVALID16) {
                    const int da = d1 >> 4, db = (d1 + 15) >> 4;              // x - da, x - db lie in [x - 63, x]: always inside the row
                    const uint32_t ka = sD2[ty][x - da - xbase], kb = sD2[ty][x - db - xbase];
                    const bool bad = (ka != 0xFFFFFFFFu) && (abs(63 - (int)(ka & 63u) - da) > d12) &&
                                     (kb != 0xFFFFFFFFu) && (abs(63 - (int)(kb & 63u) - db) > d12);
                    if (bad) d1 = V3D_INVALID16;
                }
            }
            sT[ty][tx] = (short)d1;
        }
    }
    __syncthreads();
    // ---- 4. 3x3 median (19-exchange network), 8 outputs per thread ----
    const int tx = t & (LRM_TX - 1), x = x0 + tx;
    if (x >= W) return;
#pragma unroll
    for (int ty = t >> 7; ty < LRM_TY; ty += 2) {
        const int y = y0 + ty;
        if (y >= H) break;
        if (!MED) { out[fo + (size_t)y * W + x] = sT[ty + 1][tx + 1]; continue; }
        int p0 = sT[ty][tx], p1 = sT[ty][tx + 1], p2 = sT[ty][tx + 2], p3 = sT[ty + 1][tx], p4 = sT[ty + 1][tx + 1],
            p5 = sT[ty + 1][tx + 2], p6 = sT[ty + 2][tx], p7 = sT[ty + 2][tx + 1], p8 = sT[ty + 2][tx + 2];
        V3D_SORT2(p1, p2); V3D_SORT2(p4, p5); V3D_SORT2(p7, p8); V3D_SORT2(p0, p1);
        V3D_SORT2(p3, p4); V3D_SORT2(p6, p7); V3D_SORT2(p1, p2); V3D_SORT2(p4, p5);
        V3D_SORT2(p7, p8); V3D_SORT2(p0, p3); V3D_SORT2(p5, p8); V3D_SORT2(p4, p7);
        V3D_SORT2(p3, p6); V3D_SORT2(p1, p4); V3D_SORT2(p2, p5); V3D_SORT2(p4, p7);
        V3D_SORT2(p4, p2); V3D_SORT2(p6, p4); V3D_SORT2(p4, p2);
        out[fo + (size_t)y * W + x] = (int16_t)p4;
    }
}

// ------------------------------------------------------------------------------------------------
// The same three steps as a ROW MARCH (round 3; the default for even W <= 4096).  The tile form above stages 256 source columns x 18
// rows for 128 x 16 outputs: every record is fetched 2.25 times, and a block's 36 KB of LDS leave few blocks per CU.  Here
// a 256-thread block owns a band of rows at the full image width and marches down it: per row each thread loads its own
// records (coalesced, every record read once per band + 2 halo rows per band), the right-view keys are min-scattered into ONE
// LDS row, the checked disparities go into a three-row LDS ring and the median of the previous row comes out of it.  10 bytes
// of LDS per column (19 KB at 1920), the next row's records fly while the current row is processed.  Same minimum over the
// same key set, same median network (run on two adjacent outputs at once in packed int16): bit-identical to the tile form.
// ------------------------------------------------------------------------------------------------
#define LRR_BAND 15
#define V3D_SORT2PK(a, b) { const uint32_t _lo = pk_min(a, b), _hi = pk_max(a, b); a = _lo; b = _hi; }
template <bool MED, int NPP>      // NPP: pixel PAIRS per thread and row (columns 2t, 2t+1, 2t + 512, ...): 4 covers W <= 2048, 8 W <= 4096; W even
__global__ __launch_bounds__(256) void k_lrcheck_median_rows(const uint32_t* __restrict__ wta, int W, int H, int d12, int16_t* __restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lrr_smem[];
    uint32_t* sD2 = reinterpret_cast<uint32_t*>(lrr_smem);                    // [W] right-view keys of the current row
    uint32_t* sT = reinterpret_cast<uint32_t*>(lrr_smem + (size_t)W * 4);     // [3][W/2] checked disparities, two per word: ring of rows
    const int t = threadIdx.x, f = blockIdx.z, W2 = W >> 1;
    const int ya = blockIdx.x * LRR_BAND, yb = min(ya + LRR_BAND, H);
    const size_t fo = (size_t)f * H * W;
    auto load_row = [&](int y, uint2 (&r)[NPP]) {                              // records of image row clamp(y); columns < 64 were never written
        const uint2* src = reinterpret_cast<const uint2*>(wta + fo + (size_t)min(max(y, 0), H - 1) * W);
#pragma unroll
        for (int i = 0; i < NPP; i++) r[i] = src[min(t + 256 * i, W2 - 1)];
    };
    uint2 nx[NPP];
    const int y_first = MED ? ya - 1 : ya, y_last = MED ? yb : yb - 1;
    load_row(y_first, nx);
    // rows ya-1 .. yb (median needs a row above and below; replicated at the image border = the clamped load)
    for (int y = y_first; y <= y_last; y++) {
        uint32_t rec[2 * NPP];
#pragma unroll
        for (int i = 0; i < NPP; i++) {
            const int x = 2 * (t + 256 * i);
            rec[2 * i] = (x >= V3D_D && x < W) ? nx[i].x : 0u; rec[2 * i + 1] = (x >= V3D_D && x < W) ? nx[i].y : 0u;
        }
        load_row(y + 1, nx);                                                   // the next row's records fly during this row (clamped: always in range)
        for (int x = t; x < W; x += 256) sD2[x] = 0xFFFFFFFFu;
        __syncthreads();
        // ---- right-view keys: the d of the cheapest source pixel of every target column (ties: larger d), by LDS min-scatter ----
#pragma unroll
        for (int i = 0; i < 2 * NPP; i++) {
            const uint32_t v = rec[i];
            const int best = (int)(v & 63u);
            if ((v & 0x1FFC0u) != 0u) atomicMin(&sD2[2 * (t + 256 * (i >> 1)) + (i & 1) - best], ((v >> 17) << 6) | (uint32_t)(63 - best));
        }
        __syncthreads();
        // ---- L-R check (stereosgbm.cpp: both roundings of the disparity must disagree) ----
        uint32_t* row = sT + (size_t)((y + 3) % 3) * W2;
#pragma unroll
        for (int i = 0; i < NPP; i++) {
            const int x0 = 2 * (t + 256 * i);
            if (x0 < W) {
                int dd[2];
#pragma unroll
                for (int n = 0; n < 2; n++) {
                    const int x = x0 + n;
                    int d1 = V3D_INVALID16;
                    if (x >= V3D_D) {
                        d1 = wta_d16(rec[2 * i + n]);
                        if (d1 != V3D_INVALID16) {
                            const int da = d1 >> 4, db = (d1 + 15) >> 4;
                            const uint32_t ka = sD2[x - da], kb = sD2[x - db];
                            const bool bad = (ka != 0xFFFFFFFFu) && (abs(63 - (int)(ka & 63u) - da) > d12) &&
                                             (kb != 0xFFFFFFFFu) && (abs(63 - (int)(kb & 63u) - db) > d12);
                            if (bad) d1 = V3D_INVALID16;
                        }
                    }
                    dd[n] = d1;
                }
                const uint32_t w = ((uint32_t)dd[0] & 0xFFFFu) | ((uint32_t)dd[1] << 16);
                if (MED) row[x0 >> 1] = w; else *reinterpret_cast<uint32_t*>(out + fo + (size_t)y * W + x0) = w;
            }
        }
        __syncthreads();
        if (!MED) continue;
        // ---- 3x3 median of row y-1 from ring rows y-2, y-1, y: two adjacent outputs per 19-exchange network in packed int16 ----
        const int yo = y - 1;
        if (yo >= ya && yo < yb) {                                             // uniform
            // at the image border the missing row is the replicated one: row -1 was loaded as row 0, row H as row H-1
            const uint32_t* rr[3] = { sT + (size_t)((yo - 1 + 3) % 3) * W2, sT + (size_t)((yo + 3) % 3) * W2, sT + (size_t)((yo + 1 + 3) % 3) * W2 };
#pragma unroll
            for (int i = 0; i < NPP; i++) {
                const int xw = t + 256 * i;                                    // word index: outputs 2 xw, 2 xw + 1
                if (xw < W2) {
                    uint32_t p[9];
#pragma unroll
                    for (int r = 0; r < 3; r++) {
                        const uint32_t w0 = rr[r][xw];
                        const uint32_t wl = xw > 0 ? rr[r][xw - 1] : (w0 << 16);              // column -1 replicates column 0
                        const uint32_t wr = xw + 1 < W2 ? rr[r][xw + 1] : (w0 >> 16);         // column W replicates column W-1
                        p[3 * r] = alignbit(w0, wl, 16); p[3 * r + 1] = w0; p[3 * r + 2] = alignbit(wr, w0, 16);   // (x-1, x), (x, x+1), (x+1, x+2)
                    }
                    V3D_SORT2PK(p[1], p[2]); V3D_SORT2PK(p[4], p[5]); V3D_SORT2PK(p[7], p[8]); V3D_SORT2PK(p[0], p[1]);
                    V3D_SORT2PK(p[3], p[4]); V3D_SORT2PK(p[6], p[7]); V3D_SORT2PK(p[1], p[2]); V3D_SORT2PK(p[4], p[5]);
                    V3D_SORT2PK(p[7], p[8]); V3D_SORT2PK(p[0], p[3]); V3D_SORT2PK(p[5], p[8]); V3D_SORT2PK(p[4], p[7]);
                    V3D_SORT2PK(p[3], p[6]); V3D_SORT2PK(p[1], p[4]); V3D_SORT2PK(p[2], p[5]); V3D_SORT2PK(p[4], p[7]);
                    V3D_SORT2PK(p[4], p[2]); V3D_SORT2PK(p[6], p[4]); V3D_SORT2PK(p[4], p[2]);
                    *reinterpret_cast<uint32_t*>(out + fo + (size_t)yo * W + 2 * xw) = p[4];
                }
            }
        }
        // (no barrier here: the next trip first refills sD2 -- its last readers finished before the barrier above -- and overwrites
        //  ring row (y+1) % 3, the row this median read as its first, only behind its own two barriers)
    }
}

static int launch_lrcheck_median(const uint32_t* wta, int W, int H, int n, int d12, int16_t* out, bool med, int tiles, hipStream_t st)
{
    // the row march reads record pairs and writes disparity pairs: even widths, 8-byte aligned buffers
    const bool rows_ok = !tiles && W <= 4096 && (W & 1) == 0 && (reinterpret_cast<uintptr_t>(wta) & 7) == 0 && (reinterpret_cast<uintptr_t>(out) & 3) == 0;
    if (rows_ok) {
        const dim3 grid(v3d_cdiv(H, LRR_BAND), 1, n);
        const size_t smem = (size_t)W * 4 + (size_t)3 * (W / 2) * 4;
        if (W <= 2048) {
            if (med) hipLaunchKernelGGL((k_lrcheck_median_rows<true, 4>), grid, dim3(256), smem, st, wta, W, H, d12, out);
            else hipLaunchKernelGGL((k_lrcheck_median_rows<false, 4>), grid, dim3(256), smem, st, wta, W, H, d12, out);
        } else {
            if (med) hipLaunchKernelGGL((k_lrcheck_median_rows<true, 8>), grid, dim3(256), smem, st, wta, W, H, d12, out);
            else hipLaunchKernelGGL((k_lrcheck_median_rows<false, 8>), grid, dim3(256), smem, st, wta, W, H, d12, out);
        }
    } else {
        const dim3 grid(v3d_cdiv(W, LRM_TX), v3d_cdiv(H, LRM_TY), n);
        if (med) hipLaunchKernelGGL(k_lrcheck_median<true>, grid, dim3(256), 0, st, wta, W, H, d12, out);
        else hipLaunchKernelGGL(k_lrcheck_median<false>, grid, dim3(256), 0, st, wta, W, H, d12, out);
    }
    V3D_LAUNCH_CHECK();
    return V3D_OK;
}

// ------------------------------------------------------------------------------------------------
// a-8: filterSpeckles as run-based connected-component labelling.  Components are the 4-connected
// sets of valid pixels joined where |a - b| <= maxDiff; components of at most maxSpeckleSize pixels
// are invalidated.  (1) every row is cut into horizontal runs by a block-wide scan (no atomics);
// a run is named by the index of its first pixel, carries its length, and is appended to its row's
// RUN LIST.  (2) runs of adjacent rows are joined with a lock-free union-find, one union per overlapping
// run pair instead of one per pixel.  (3) run lengths are added at the roots, (4) small components are
// erased -- (3) and (4) walk the run lists (tens of runs per row), not the pixels.
// The outcome is schedule-independent: union-find yields the same partition in any order, and
// sizes are only ever compared against the threshold.
//   lab  [n]: run start for non-start pixels (constant); parent pointer for run starts; -1 invalid   (dense)
//   runs [n]: per row, the run starts of that row in x order, ended by -1 if the row has fewer than W runs
//   csz  [n]: at run starts only: the run's length, and at a root the running size of its component
// Dense traffic per pixel: img read + lab write (k_ccl_runs), two img rows read (k_ccl_vmerge); the round-1 form also
// wrote and re-read dense length / size planes and re-read lab per pixel (~40 B per pixel, 0.81 ms per 30 frames).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int ccl_ld(const int* L, int i) { return __hip_atomic_load(L + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int ccl_find(const int* L, int i)
{
    int p = ccl_ld(L, i);
    while (p != i) { i = p; p = ccl_ld(L, i); }
    return i;
}
// find with path halving: every visited node is re-pointed at its grandparent.  Safe next to concurrent
// atomicMin hooks: a node is only ever re-pointed at one of its own ancestors, never at a slot seen as a root.
__device__ __forceinline__ int ccl_find_halve(int* L, int i)
{
    for (;;) {
        const int p = ccl_ld(L, i);
        if (p == i) return i;
        const int gp = ccl_ld(L, p);
        if (gp == p) return p;
        __hip_atomic_store(L + i, gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        i = gp;
    }
}
__device__ __forceinline__ void ccl_union(int* L, int a, int b)
{
    for (;;) {
        a = ccl_find_halve(L, a); b = ccl_find_halve(L, b);
        if (a == b) return;
        if (a < b) { const int t = a; a = b; b = t; }          // a > b: hang the larger root under the smaller
        const int old = atomicMin(L + a, b);
        if (old == a) return;
        a = old;                                               // a was re-rooted meanwhile: carry on from its old parent
    }
}
__device__ __forceinline__ bool ccl_conn(int a, int b, int newVal, int maxDiff) { return a != newVal && b != newVal && abs(a - b) <= maxDiff; }

// one WAVE per image row (four rows per block), 256 pixels per step -- FOUR consecutive pixels per lane (one 8-byte load,
// one 16-byte label store): "latest run start at or before x" is a 3-step max inside the lane + an inclusive max-scan of
// the lane totals over the wave (DPP row shifts + row broadcasts, no LDS), "run starts before x" four ballots + popcounts;
// the carry from step to step rides in SGPRs, no barrier.  A 1920-pixel row is 8 steps.  (Round 2's one-pixel-per-lane
// form ran 30 steps per row with six ds_bpermute exchanges each: 0.157 ms per 30 frames.)
#define V3D_DPP_ROW_BCAST15 0x142
#define V3D_DPP_ROW_BCAST31 0x143
__device__ __forceinline__ uint32_t wave_incl_max_u32(uint32_t v)      // inclusive max-scan over the 64 lanes, identity 0
{
    v = max(v, dpp_mov<V3D_DPP_ROW_SHR(1)>(0u, v));
    v = max(v, dpp_mov<V3D_DPP_ROW_SHR(2)>(0u, v));
    v = max(v, dpp_mov<V3D_DPP_ROW_SHR(4)>(0u, v));
    v = max(v, dpp_mov<V3D_DPP_ROW_SHR(8)>(0u, v));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, V3D_DPP_ROW_BCAST15, 0xA, 0xF, false));   // rows 1, 3 take lane 15 / 47
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, V3D_DPP_ROW_BCAST31, 0xC, 0xF, false));   // rows 2, 3 take lane 31
    return v;
}
__global__ __launch_bounds__(256) void k_ccl_runs(const int16_t* __restrict__ img, int W, int H, int newVal, int maxDiff,
                                                  int* __restrict__ lab, int* __restrict__ runs, int* __restrict__ csz)
{
    const int y = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (y >= H) return;                                        // wave-uniform
    const size_t fo = (size_t)blockIdx.z * W * H + (size_t)y * W;
    const int16_t* row = img + fo;
    const bool vec = (W & 3) == 0 && ((reinterpret_cast<uintptr_t>(img) & 7) | (reinterpret_cast<uintptr_t>(lab) & 15)) == 0;   // rows (and frames) start aligned: vector loads / stores
    int carry = 0, nrun = 0;                                   // (latest run start so far) + 1, runs so far (wave-uniform)
    int last_v = newVal;                                       // value of the pixel left of this step's first one
    auto ld4 = [&](int x0, int (&v)[4]) {                      // pixels x0 .. x0+3 (newVal beyond the row)
        if (vec) {
            if (x0 < W) { const uint2 t = *reinterpret_cast<const uint2*>(row + x0);
                          v[0] = (short)(t.x & 0xFFFFu); v[1] = (short)(t.x >> 16); v[2] = (short)(t.y & 0xFFFFu); v[3] = (short)(t.y >> 16); }
            else { v[0] = v[1] = v[2] = v[3] = newVal; }
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++) v[k] = x0 + k < W ? (int)row[x0 + k] : newVal;
        }
    };
    int nx[4];
    ld4(4 * lane, nx);
    for (int xs = 0; xs < W; xs += 256) {
        const int x0 = xs + 4 * lane;
        int v[4] = { nx[0], nx[1], nx[2], nx[3] };
        ld4(x0 + 256, nx);                                     // next step's pixels fly during this step
        // neighbours across the lane boundary: left of v[0] = lane-1's v[3], right of v[3] = lane+1's v[0]
        int pv = __shfl_up(v[3], 1), nv = __shfl_down(v[0], 1);
        const int vn0 = __builtin_amdgcn_readfirstlane(nx[0]); // first pixel of the next step
        pv = lane == 0 ? last_v : pv;
        nv = lane == 63 ? vn0 : nv;
        last_v = __builtin_amdgcn_readlane(v[3], 63);
        bool valid[4], start[4];
        uint32_t c[4];                                         // inclusive (latest start + 1) inside the lane
#pragma unroll
        for (int k = 0; k < 4; k++) {
            valid[k] = x0 + k < W && v[k] != newVal;
            start[k] = valid[k] && !ccl_conn(k ? v[k - 1] : pv, v[k], newVal, maxDiff);
            const uint32_t m = start[k] ? (uint32_t)(x0 + k + 1) : 0u;
            c[k] = k ? max(c[k - 1], m) : m;
        }
        const uint32_t incl = wave_incl_max_u32(c[3]);
        uint32_t excl = (uint32_t)__shfl_up((int)incl, 1);
        excl = max(lane == 0 ? 0u : excl, (uint32_t)carry);    // latest start + 1 left of this lane's pixels
        carry = max(carry, (int)__builtin_amdgcn_readlane((int)incl, 63));
        int before = nrun;                                     // run starts left of this lane's pixels
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const unsigned long long sm = __builtin_amdgcn_ballot_w64(start[k]);
            before += __popcll(sm & ((1ull << lane) - 1ull));
            nrun += __popcll(sm);
        }
        int labv[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int cur = (int)max(c[k], excl) - 1;          // run start of pixel x0+k (if valid)
            labv[k] = valid[k] ? y * W + cur : -1;
            if (start[k]) { runs[fo + before] = y * W + x0 + k; before++; }
            const int right = k < 3 ? v[k + 1] : nv;
            if (valid[k] && !(x0 + k + 1 < W && ccl_conn(v[k], right, newVal, maxDiff))) csz[fo + cur] = x0 + k - cur + 1;   // the run's last pixel: its length
        }
        if (vec) { if (x0 < W) *reinterpret_cast<int4*>(lab + fo + x0) = make_int4(labv[0], labv[1], labv[2], labv[3]); }
        else {
#pragma unroll
            for (int k = 0; k < 4; k++) if (x0 + k < W) lab[fo + x0 + k] = labv[k];
        }
    }
    if (lane == 0 && nrun < W) runs[fo + nrun] = -1;           // end of the row's run list
}

// Two launches: LEVEL 0 joins the row pairs inside bands of VM_BAND rows (trees at most VM_BAND deep), LEVEL 1 the
// band boundaries.  The partition is the same in any order; what changes is the depth of the parent chains the
// racing unions build, i.e. how many dependent global loads a find costs.
// A thread tests EIGHT consecutive pixels of a row pair (two 16-byte loads + the pair left of them); one wave covers 512
// columns.  (Round 2's one-pixel-per-thread form launched a million 30-instruction waves per batch: 0.21 ms per 30 frames,
// bound by wave launch, not by its loads.)
#ifndef VM_BAND
#define VM_BAND 16
#endif
template <int LEVEL>
__global__ __launch_bounds__(64) void k_ccl_vmerge(const int16_t* __restrict__ img, int W, int H, int newVal, int maxDiff, int* __restrict__ lab)
{
    const int x0 = (blockIdx.x * 64 + threadIdx.x) * 8;
    const int y = LEVEL == 0 ? blockIdx.y + blockIdx.y / (VM_BAND - 1) : blockIdx.y * VM_BAND + VM_BAND - 1;
    if (x0 >= W || y + 1 >= H) return;
    const size_t fo = (size_t)blockIdx.z * W * H;
    const int16_t* im = img + fo; int* L = lab + fo;
    const int i0 = y * W + x0;
    int v[9], u[9];                                            // [0] = the pixel pair left of this thread's eight (itself at x0 = 0)
    v[0] = im[i0 - (x0 > 0 ? 1 : 0)]; u[0] = im[i0 + W - (x0 > 0 ? 1 : 0)];
    if ((W & 7) == 0 && (reinterpret_cast<uintptr_t>(img) & 15) == 0) {   // rows start 16-byte aligned
        const uint4 a = *reinterpret_cast<const uint4*>(im + i0), b = *reinterpret_cast<const uint4*>(im + i0 + W);
        const uint32_t aw[4] = { a.x, a.y, a.z, a.w }, bw[4] = { b.x, b.y, b.z, b.w };
#pragma unroll
        for (int k = 0; k < 4; k++) {
            v[1 + 2 * k] = (short)(aw[k] & 0xFFFFu); v[2 + 2 * k] = (short)(aw[k] >> 16);
            u[1 + 2 * k] = (short)(bw[k] & 0xFFFFu); u[2 + 2 * k] = (short)(bw[k] >> 16);
        }
    } else {
#pragma unroll
        for (int k = 0; k < 8; k++) { const int xc = min(x0 + k, W - 1) - x0; v[1 + k] = im[i0 + xc]; u[1 + k] = im[i0 + W + xc]; }
    }
#pragma unroll
    for (int k = 0; k < 8; k++) {
        if (x0 + k >= W) break;
        if (!ccl_conn(v[1 + k], u[1 + k], newVal, maxDiff)) continue;
        // the pixel to my left joins the same two runs: it (or one further left) does the union
        if (x0 + k > 0 && ccl_conn(v[k], u[k], newVal, maxDiff) && ccl_conn(v[k], v[1 + k], newVal, maxDiff) && ccl_conn(u[k], u[1 + k], newVal, maxDiff)) continue;
        ccl_union(L, L[i0 + k], L[i0 + k + W]);
    }
}

// (3) and (4): one WAVE per image row walks that row's run list, 64 runs per step.
// count: every non-root run adds its length to its root (a root's own length is already there).  Only "<= maxSize or
// not" matters: stop adding once the root is known to be large.
__global__ __launch_bounds__(256) void k_ccl_count(int W, int H, int maxSize, int* __restrict__ lab, const int* __restrict__ runs, int* __restrict__ csz)
{
    const int y = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (y >= H) return;                                        // wave-uniform
    const size_t fo = (size_t)blockIdx.z * W * H;
    int* L = lab + fo; int* C = csz + fo;
    const int* rl = runs + fo + (size_t)y * W;
    for (int k0 = 0; k0 < W; k0 += 64) {
        const int k = k0 + lane;
        const int s = k < W ? rl[k] : -1;
        // entries behind the end marker are stale: a lane counts only if every entry before it in this step is a run
        const unsigned long long endm = __builtin_amdgcn_ballot_w64(s < 0);
        const int first_end = endm ? __builtin_ctzll(endm) : 64;
        if (lane < first_end) {
            const int r = ccl_find(L, s);
            if (r != s) {
                L[s] = r;                                      // path compression (the forest is final here)
                if (__hip_atomic_load(C + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) <= maxSize) atomicAdd(C + r, C[s]);
            }
        }
        if (first_end < 64) break;                             // wave-uniform
    }
}

// apply: a run whose component is small is overwritten pixel by pixel (at most maxSize of them: the loop is short and rare)
__global__ __launch_bounds__(256) void k_ccl_apply(int16_t* __restrict__ img, int W, int H, int newVal, int maxSize,
                                                   const int* __restrict__ lab, const int* __restrict__ runs, const int* __restrict__ csz)
{
    const int y = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (y >= H) return;                                        // wave-uniform
    const size_t fo = (size_t)blockIdx.z * W * H;
    const int* L = lab + fo;
    const int* rl = runs + fo + (size_t)y * W;
    const int row_end = (y + 1) * W;
    for (int k0 = 0; k0 < W; k0 += 64) {
        const int k = k0 + lane;
        const int s = k < W ? rl[k] : -1;
        const unsigned long long endm = __builtin_amdgcn_ballot_w64(s < 0);
        const int first_end = endm ? __builtin_ctzll(endm) : 64;
        if (lane < first_end) {
            const int r = ccl_find(L, s);
            if (csz[fo + r] <= maxSize) {
                img[fo + s] = (int16_t)newVal;
                for (int i = s + 1; i < row_end && L[i] == s; i++) img[fo + i] = (int16_t)newVal;    // non-start pixels carry their run's start
            }
        }
        if (first_end < 64) break;                             // wave-uniform
    }
}

// the five launches; ws = 3 * n_pixels * frames int32
static int launch_speckles(int16_t* img, int W, int H, int frames, int newVal, int maxSize, int maxDiff, int32_t* ws, hipStream_t st)
{
    const int px = W * H;
    int* lab = ws; int* runs = ws + (size_t)px * frames; int* csz = ws + (size_t)px * frames * 2;
    hipLaunchKernelGGL(k_ccl_runs, dim3(v3d_cdiv(H, 4), 1, frames), dim3(256), 0, st, img, W, H, newVal, maxDiff, lab, runs, csz);
    // rows y with (y % VM_BAND) != VM_BAND-1 first (blockIdx.y enumerates them), then the band boundaries
    hipLaunchKernelGGL(k_ccl_vmerge<0>, dim3(v3d_cdiv(W, 512), H - H / VM_BAND, frames), dim3(64), 0, st, img, W, H, newVal, maxDiff, lab);
    if (H / VM_BAND > 0) hipLaunchKernelGGL(k_ccl_vmerge<1>, dim3(v3d_cdiv(W, 512), H / VM_BAND, frames), dim3(64), 0, st, img, W, H, newVal, maxDiff, lab);
    hipLaunchKernelGGL(k_ccl_count, dim3(v3d_cdiv(H, 4), 1, frames), dim3(256), 0, st, W, H, maxSize, lab, runs, csz);
    hipLaunchKernelGGL(k_ccl_apply, dim3(v3d_cdiv(H, 4), 1, frames), dim3(256), 0, st, img, W, H, newVal, maxSize, lab, runs, csz);
    V3D_LAUNCH_CHECK();
    return V3D_OK;
}

// parity-test export of C as int16 (v3d_sgbm_debug_cost_volume): 8 lanes per pixel, 8 disparities each
__global__ __launch_bounds__(256) void k_c_export(const unsigned char* __restrict__ C, size_t npx, int P2, int16_t* __restrict__ out)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, px = i >> 3;
    const int dl = (int)(i & 7);
    if (px >= npx) return;
    const uint4 v = c_unpack(*reinterpret_cast<const typename CRaw<8>::type*>(C + px * C_PXB + c_lane_off<8>(dl)), dl, pk_bcast(P2));
    *reinterpret_cast<uint4*>(out + px * V3D_D + 8 * dl) = v;
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
struct v3d_sgbm {
    v3d_sgbm_params prm;
    int device, maxW, maxH, maxB;
    int P1, P2, ftzero, uniq, d12;
    int dpl;                                    // disparities per lane in k_chain (4 or 8)
    uint4* rec;
    unsigned char* C;                           // cost volume, C_PXB bytes per pixel
    int16_t* S;
    uint32_t* wta;                              // WTA records, one per pixel
    uint32_t* ckpt;                             // k_hfused checkpoints
    unsigned long long* gran;                   // k_vdd edge granules
    size_t gran_bytes;
    int* vdd_err;
    uint32_t vdd_seq;
    int vdd_mode;                               // 0 off, 1 on
    int vdd_dpl;                                // forced k_vdd mapping (4 / 8), 0 = choose per call
    int cost_band;                              // rows per k_cost workgroup
    int vdd_xcd, cost_xcd, hf_xcd;
    int lrm_tiles;                              // 1: L-R check + median as 128 x 16 tiles (the round-2 form) instead of the row march
    int hf_persist;                             // k_hfused: 0 one wave per row group, 1 resident waves draw row groups from a ticket counter
    int* hf_ticket;
    int vdd_mf4, vdd_mf8;                       // frames per launch of each mapping at maxW (reported by get_option)
    int vdd_occ4, vdd_occ8, ncu;                // occupancy query results the bounds are derived from
    int reserve_cus;                            // CUs left to other streams' kernels (e.g. an RCCL collective) when sizing a lock-step launch
    int vdd_launch_frames;                      // 0 = size launches from the occupancy query; > 0: frames per launch (tests: over-sized launches)
    int vdd_spin_limit;                         // 0 = derive from the row count
    int* err_host;                              // pinned, device-visible: lock-step time-outs seen by k_vdd_guard
    hipEvent_t vdd_done_ev;                     // recorded behind the last lock-step launch of a compute call
    bool vdd_ev_recorded;
    bool hfused;
    int hsplit;                                 // 1: left->right scan of the horizontal pass as its own launch (k_hscan)
    int32_t* labels;
    size_t bytes;
    // optional per-stage HIP-event timing (v3d_sgbm_profile): events live on the caller's stream
    bool prof_on;
    int prof_calls;
    std::vector<hipEvent_t> prof_ev;            // [call][V3D_NSTAGE + 1]
};

enum { ST_PREFILTER = 0, ST_COST, ST_V2, ST_D1, ST_D3, ST_H0, ST_V2R, ST_D1R, ST_D3R, ST_H4_WTA, ST_LRCHECK, ST_MEDIAN, ST_SPECKLE, V3D_NSTAGE };
static const char* const g_stage_names[V3D_NSTAGE] = { "prefilter", "cost", "chain_v2", "chain_d1", "chain_d3", "chain_h0",
    "chain_v2r", "chain_d1r", "chain_d3r", "chain_h4_wta", "lrcheck", "median", "speckles" };
#define V3D_PROF_MAX_CALLS 512

// record the event that closes stage `st` (and opens st+1); stages that are skipped record nothing
static inline void prof_mark(v3d_sgbm* h, int slot, hipStream_t stm)
{
    if (!h->prof_on || h->prof_calls >= V3D_PROF_MAX_CALLS) return;
    (void)hipEventRecord(h->prof_ev[(size_t)h->prof_calls * (V3D_NSTAGE + 1) + slot], stm);
}

extern "C" void v3d_sgbm_default_params(v3d_sgbm_params* p)
{
    p->minDisparity = 0; p->numDisparities = 64; p->blockSize = 5; p->P1 = 8 * 3 * 25; p->P2 = 32 * 3 * 25;
    p->disp12MaxDiff = 1; p->preFilterCap = 0; p->uniquenessRatio = 10; p->speckleWindowSize = 100;
    p->speckleRange = 32; p->mode = V3D_MODE_SGBM;
}

template <typename T> static int ws_alloc(T** p, size_t n, size_t* total)
{
    const size_t b = ((n * sizeof(T)) + 255) & ~(size_t)255;
    V3D_HIP_CHECK(hipMalloc((void**)p, b));
    *total += b;
    return V3D_OK;
}

// frames one lock-step launch should hold at cost-region width W1: the workgroup slots the occupancy query reports
// (minus the CUs the host says other streams keep busy, two slots each) over the strips of one frame.  No safety
// margin: an over-sized launch is slow, not wrong (k_vdd: in-order dispatch keeps whole frames resident).
static int vdd_frames_per_launch(const v3d_sgbm* h, int dpl, int W1)
{
    const int cus = h->ncu - h->reserve_cus > 0 ? h->ncu - h->reserve_cus : 0;
    return ((dpl == 8 ? h->vdd_occ8 : h->vdd_occ4) * cus) / v3d_cdiv(W1, 16 * dpl);
}
static void vdd_size_launches(v3d_sgbm* h)
{
    h->vdd_mf4 = vdd_frames_per_launch(h, 4, h->maxW - V3D_D);
    h->vdd_mf8 = vdd_frames_per_launch(h, 8, h->maxW - V3D_D);
}
static inline bool vdd_usable(const v3d_sgbm* h) { return h->vdd_mode && h->vdd_mf4 >= 1 && h->vdd_mf8 >= 1; }

// Tuning switches of a handle (A/B work, tests, tools): the defaults are the measured best, nothing here changes results.
//   "lockstep"       1/0  top-down paths in one co-resident lock-step pass (k_vdd) / one launch per direction (k_chain)
//   "hfused"         1/0  both horizontal paths + WTA in one launch (k_hfused) / two k_chain launches
//   "chain_dpl"      4/8  disparities per lane in k_chain and k_hfused
//   "vdd_dpl"        0/4/8  k_vdd strip mapping (0 = choose per call from the batch size)
//   "cost_band"      >= 8 rows per k_cost workgroup
//   "cost_xcd", "vdd_xcd", "hf_xcd"   1/0  XCD-contiguous workgroup order of that kernel
//   "hsplit"         1/0  k_hfused's left-to-right scan as its own launch (measured: no gain; kept for A/B)
//   "lrm_tiles"      0/1  L-R check + median as a row march over full-width bands / as 128 x 16 tiles (round-2 form; also the form for W > 4096)
//   "hf_persist"     1/0  k_hfused as the resident number of waves drawing row groups from a ticket counter / one wave per row group
//   "reserve_cus"    CUs other streams keep busy while a lock-step pass runs (shrinks the frames per launch)
//   "vdd_launch_frames"  frames per lock-step launch (0 = from the occupancy query); larger than the chip holds is safe, slow
//   "vdd_spin_limit" poll rounds a lane may wait in a lock-step pass (0 = 64 per row + 4096; -1 = test hook: every
//                    workgroup reports a time-out, which drives the guard / V3D_ERR_LOCKSTEP path deterministically)
extern "C" int v3d_sgbm_set_option(v3d_sgbm* h, const char* key, int value)
{
    if (!h || !key) { v3d_set_error("null argument"); return V3D_ERR_ARG; }
    auto is = [&](const char* k) { return strcmp(key, k) == 0; };
    auto bad = [&]() { v3d_set_error("option %s: value %d out of range", key, value); return V3D_ERR_ARG; };
    if (is("lockstep")) { if (value != 0 && value != 1) return bad(); h->vdd_mode = value; }
    else if (is("hfused")) { if (value != 0 && value != 1) return bad(); h->hfused = value != 0; }
    else if (is("chain_dpl")) { if (value != 4 && value != 8) return bad(); h->dpl = value; }
    else if (is("hsplit")) { if (value != 0 && value != 1) return bad(); h->hsplit = value; }
    else if (is("vdd_dpl")) { if (value != 0 && value != 4 && value != 8) return bad(); h->vdd_dpl = value; }
    else if (is("cost_band")) { if (value < 8 || value > 65536) return bad(); h->cost_band = value; }
    else if (is("cost_xcd")) { if (value != 0 && value != 1) return bad(); h->cost_xcd = value; }
    else if (is("vdd_xcd")) { if (value != 0 && value != 1) return bad(); h->vdd_xcd = value; }
    else if (is("hf_xcd")) { if (value != 0 && value != 1) return bad(); h->hf_xcd = value; }
    else if (is("hf_persist")) { if (value != 0 && value != 1) return bad(); h->hf_persist = value; }
    else if (is("lrm_tiles")) { if (value != 0 && value != 1) return bad(); h->lrm_tiles = value; }
    else if (is("reserve_cus")) { if (value < 0 || value > h->ncu) return bad(); h->reserve_cus = value; vdd_size_launches(h); }
    else if (is("vdd_spin_limit")) { if (value < -1) return bad(); h->vdd_spin_limit = value; }
    else if (is("vdd_launch_frames")) { if (value < 0) return bad(); h->vdd_launch_frames = value; }
    else { v3d_set_error("unknown option %s", key); return V3D_ERR_ARG; }
    return V3D_OK;
}
extern "C" int v3d_sgbm_get_option(const v3d_sgbm* h, const char* key, int* value)
{
    if (!h || !key || !value) { v3d_set_error("null argument"); return V3D_ERR_ARG; }
    auto is = [&](const char* k) { return strcmp(key, k) == 0; };
    if (is("lockstep")) *value = vdd_usable(h) ? 1 : 0;
    else if (is("hfused")) *value = h->hfused ? 1 : 0;
    else if (is("chain_dpl")) *value = h->dpl;
    else if (is("hsplit")) *value = h->hsplit;
    else if (is("vdd_dpl")) *value = h->vdd_dpl;
    else if (is("cost_band")) *value = h->cost_band;
    else if (is("cost_xcd")) *value = h->cost_xcd;
    else if (is("vdd_xcd")) *value = h->vdd_xcd;
    else if (is("hf_xcd")) *value = h->hf_xcd;
    else if (is("hf_persist")) *value = h->hf_persist;
    else if (is("lrm_tiles")) *value = h->lrm_tiles;
    else if (is("reserve_cus")) *value = h->reserve_cus;
    else if (is("vdd_spin_limit")) *value = h->vdd_spin_limit;
    else if (is("vdd_launch_frames")) *value = h->vdd_launch_frames;
    else if (is("vdd_frames_per_launch_dpl4")) *value = h->vdd_mf4;       // read-only: the co-residency bounds in force
    else if (is("vdd_frames_per_launch_dpl8")) *value = h->vdd_mf8;
    else { v3d_set_error("unknown option %s", key); return V3D_ERR_ARG; }
    return V3D_OK;
}

extern "C" int v3d_sgbm_create(const v3d_sgbm_params* prm, int device, int maxW, int maxH, int maxB, v3d_sgbm** out)
{
    if (!prm || !out) { v3d_set_error("null argument"); return V3D_ERR_ARG; }
    if (prm->minDisparity != 0 || prm->numDisparities != V3D_D || prm->blockSize != 5) {
        v3d_set_error("this build supports minDisparity=0, numDisparities=64, blockSize=5 (got %d, %d, %d)",
                      prm->minDisparity, prm->numDisparities, prm->blockSize);
        return V3D_ERR_UNSUPPORTED;
    }
    if (prm->mode != V3D_MODE_SGBM && prm->mode != V3D_MODE_HH) { v3d_set_error("unsupported mode %d", prm->mode); return V3D_ERR_UNSUPPORTED; }
    if (maxW <= V3D_D + 4 || maxH < 1 || maxB < 1) { v3d_set_error("bad geometry %dx%d batch %d", maxW, maxH, maxB); return V3D_ERR_ARG; }
    // one frame's cost volume must stay below 2 GiB: k_cost addresses it through a range-checked buffer whose
    // out-of-range marker is bit 31 of the byte offset
    if ((size_t)maxW * maxH * V3D_D >= ((size_t)1 << 30)) { v3d_set_error("frame too large for 31-bit volume byte offsets"); return V3D_ERR_UNSUPPORTED; }
    V3D_HIP_CHECK(hipSetDevice(device));
    v3d_sgbm* h = new v3d_sgbm();
    h->prof_on = false; h->prof_calls = 0;
    h->prm = *prm; h->device = device; h->maxW = maxW; h->maxH = maxH; h->maxB = maxB;
    h->P1 = prm->P1 > 0 ? prm->P1 : 2;
    h->P2 = prm->P2 > 0 ? prm->P2 : 5; if (h->P2 < h->P1 + 1) h->P2 = h->P1 + 1;
    h->ftzero = (prm->preFilterCap > 15 ? prm->preFilterCap : 15) | 1;
    h->uniq = prm->uniquenessRatio >= 0 ? prm->uniquenessRatio : 10;
    h->d12 = prm->disp12MaxDiff > 0 ? prm->disp12MaxDiff : 1;
    // int16 headroom of the packed recurrence: L <= C <= P2 + 25*(2*ftzero + 63) and delta = min L + P2
    // must stay below 32767 (OpenCV forms delta in int32; the reference's P2 = 2400 is far inside)
    if (2 * h->P2 + 25 * (2 * h->ftzero + 63) >= 32767 || h->ftzero > 31) {   // ftzero <= 31: BT bytes add pairwise without carry in k_cost
        const int p2 = h->P2; delete h;
        v3d_set_error("P2=%d / preFilterCap exceed the int16 range of the packed SGM recurrence (need 2*P2 + 25*(2*ftzero+63) < 32767)", p2);
        return V3D_ERR_UNSUPPORTED;
    }
    h->dpl = 4;
    const size_t px = (size_t)maxW * maxH * maxB, vol = vol_frame(maxH, maxW - V3D_D) * maxB;
    h->bytes = 0;
    int rc = 0;
    rc |= ws_alloc(&h->rec, px, &h->bytes);
    rc |= ws_alloc(&h->C, c_frame(maxH, maxW - V3D_D) * maxB, &h->bytes);   rc |= ws_alloc(&h->S, vol, &h->bytes);
    rc |= ws_alloc(&h->wta, px, &h->bytes); rc |= ws_alloc(&h->labels, px * 3, &h->bytes);
    {   // checkpoints: per frame, per wave (DPL rows), per K-pixel block: 64 lanes x (DPL/2 + 1) dwords
        const int W1m = maxW - V3D_D;
        const size_t c4 = (size_t)v3d_cdiv(maxH, 4) * v3d_cdiv(W1m, 16) * 64 * 3, c8 = (size_t)v3d_cdiv(maxH, 8) * v3d_cdiv(W1m, 8) * 64 * 5;
        rc |= ws_alloc(&h->ckpt, (c4 > c8 ? c4 : c8) * maxB, &h->bytes);
    }
    h->hfused = true; h->hsplit = V3D_X_SPLIT;
    {
        h->vdd_dpl = 0;
        const int nstrips_max = v3d_cdiv(maxW - V3D_D, 64);          // granule ring sized for the narrower strips
        const size_t ng = (size_t)maxB * nstrips_max * 2 * VDD_RING * VDD_GRAN;
        rc |= ws_alloc(&h->gran, ng, &h->bytes); h->gran_bytes = ng * sizeof(unsigned long long);
        rc |= ws_alloc(&h->vdd_err, 64, &h->bytes);
        h->hf_ticket = h->vdd_err ? h->vdd_err + 32 : nullptr;       // k_hfused's ticket counter shares the small allocation
        if (!rc) { (void)hipMemset(h->gran, 0, ng * sizeof(unsigned long long)); (void)hipMemset(h->vdd_err, 0, 64 * sizeof(int)); }
        h->err_host = nullptr;
        if (!rc && hipHostMalloc((void**)&h->err_host, 64, hipHostMallocDefault) != hipSuccess) { h->err_host = nullptr; rc = 1; }
        if (h->err_host) *h->err_host = 0;
        h->vdd_done_ev = nullptr; h->vdd_ev_recorded = false;
        if (!rc && hipEventCreateWithFlags(&h->vdd_done_ev, hipEventDisableTiming) != hipSuccess) { h->vdd_done_ev = nullptr; rc = 1; }
        h->vdd_seq = 1;
        h->vdd_mode = 1;                                  // lock-step pass on; option "lockstep" = 0 falls back to three k_chain launches
        h->reserve_cus = 0; h->vdd_spin_limit = 0; h->vdd_launch_frames = 0;
        // all strips of a launch must be resident together: bound frames per launch by the occupancy query, with margin
        int b4 = 0, b8 = 0, ncu = 0;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&b4, k_vdd<4, true>, 1024, 0);
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&b8, k_vdd<8, true>, 1024, 0);
        (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device);
        h->vdd_occ4 = b4 > 2 ? 2 : b4; h->vdd_occ8 = b8 > 2 ? 2 : b8; h->ncu = ncu;
        vdd_size_launches(h);
    }
    h->lrm_tiles = 0;
    h->hf_persist = 1;                                   // measured: -4 % at 34 / 68 frames, neutral at 30
    h->vdd_xcd = 0; h->hf_xcd = 0; h->cost_xcd = 1;      // measured: XCD-contiguous order pays for k_cost only
    h->cost_band = 90;
    if (rc) { v3d_sgbm_destroy(h); return V3D_ERR_HIP; }
    *out = h;
    return V3D_OK;
}

extern "C" void v3d_sgbm_destroy(v3d_sgbm* h)
{
    if (!h) return;
    (void)hipSetDevice(h->device);
    void* ptrs[] = { h->rec, h->C, h->S, h->wta, h->labels, h->ckpt, h->gran, h->vdd_err };
    for (void* p : ptrs) if (p) (void)hipFree(p);
    if (h->err_host) (void)hipHostFree(h->err_host);
    if (h->vdd_done_ev) (void)hipEventDestroy(h->vdd_done_ev);
    for (hipEvent_t e : h->prof_ev) (void)hipEventDestroy(e);
    delete h;
}

extern "C" size_t v3d_sgbm_workspace_bytes(const v3d_sgbm* h) { return h ? h->bytes : 0; }

template <bool HORIZ, int XS, bool YREV, int MODE>
static void launch_chain(const v3d_sgbm* h, const ChainArgs& a, hipStream_t st)
{
    const int NC = HORIZ ? a.H : (XS == 0 ? a.W1 : a.W1 + a.H - 1);
    if (h->dpl == 4) {
        const int groups = v3d_cdiv(NC, 4), waves = groups * a.nframes;
        hipLaunchKernelGGL((k_chain<HORIZ, XS, YREV, MODE, 4>), dim3(v3d_cdiv(waves, 4)), dim3(256), 0, st, a);
    } else {
        const int groups = v3d_cdiv(NC, 8), waves = groups * a.nframes;
        hipLaunchKernelGGL((k_chain<HORIZ, XS, YREV, MODE, 8>), dim3(v3d_cdiv(waves, 4)), dim3(256), 0, st, a);
    }
}

static int check_geometry(const v3d_sgbm* h, int n, int W, int H, int pitch)
{
    if (!h) { v3d_set_error("null handle"); return V3D_ERR_ARG; }
    if (n < 1 || n > h->maxB || W > h->maxW || H > h->maxH || (size_t)W * H > (size_t)h->maxW * h->maxH) {
        v3d_set_error("frame %dx%d x%d exceeds the handle's workspace (%dx%d x%d)", W, H, n, h->maxW, h->maxH, h->maxB);
        return V3D_ERR_ARG;
    }
    if (W <= V3D_D + 4 || H < 1 || pitch < W) { v3d_set_error("bad frame geometry W=%d H=%d pitch=%d (need W > 68)", W, H, pitch); return V3D_ERR_ARG; }
    return V3D_OK;
}

// one lock-step pass over the three top-down (or, rev, bottom-up) paths; frames per launch bounded by co-residency.
// mapping: 4 disparities per lane (64-column strips) while the whole batch fits one co-resident launch, else
// 8 per lane (128-column strips: ~30 % fewer instructions per element, twice the frames per launch)
static void launch_vdd(v3d_sgbm* h, int n, int W1, int H, bool rev, hipStream_t st)
{
    // sized from THIS call's width (a handle made for 4K frames holds more 1080p frames per launch)
    const int mf4 = vdd_frames_per_launch(h, 4, W1), mf8 = vdd_frames_per_launch(h, 8, W1);
    const int dpl = h->vdd_dpl ? h->vdd_dpl : (n <= mf4 ? 4 : 8);
    const int mf = h->vdd_launch_frames > 0 ? h->vdd_launch_frames : dpl == 8 ? (mf8 > 0 ? mf8 : 1) : (mf4 > 0 ? mf4 : 1);
    const int nl = v3d_cdiv(n, mf), per = v3d_cdiv(n, nl);           // equal shares: two launches of 20, not 34 + 6
    for (int f0 = 0; f0 < n; f0 += per) {
        VddArgs v;
        const int nf = n - f0 < per ? n - f0 : per;
        v.C = h->C + (size_t)f0 * c_frame(H, W1); v.S = h->S + (size_t)f0 * vol_frame(H, W1);
        v.W1 = W1; v.H = H; v.nframes = nf; v.nstrips = v3d_cdiv(W1, 16 * dpl); v.P1 = h->P1; v.P2 = h->P2;
        v.seq = (h->vdd_seq++) & 0xFFFFFu;
        if (v.seq == 0) {                                   // the 20-bit launch sequence wrapped: sweep the stale tags (once per 2^20 launches)
            (void)hipMemsetAsync(h->gran, 0, h->gran_bytes, st);
            v.seq = (h->vdd_seq++) & 0xFFFFFu;
        }
        v.gran = h->gran; v.err = h->vdd_err; v.xcd = h->vdd_xcd;
        v.spin_limit = h->vdd_spin_limit != 0 ? h->vdd_spin_limit : VDD_SPIN_PER_ROW * H + VDD_SPIN_SLACK;
        const dim3 grid(v.nstrips * nf), block(1024);
        if (dpl == 8) { if (rev) hipLaunchKernelGGL((k_vdd<8, true>), grid, block, 0, st, v); else hipLaunchKernelGGL((k_vdd<8, false>), grid, block, 0, st, v); }
        else { if (rev) hipLaunchKernelGGL((k_vdd<4, true>), grid, block, 0, st, v); else hipLaunchKernelGGL((k_vdd<4, false>), grid, block, 0, st, v); }
    }
}

// A lock-step pass of an earlier call timed out (k_vdd_guard raised the host flag): that call's output was
// invalidated on the device, and the handle refuses further work until the host has reacted --
// v3d_sgbm_set_lockstep(h, 0 or 1) clears the state.  No synchronisation here: one read of pinned memory.
static int lockstep_state(const v3d_sgbm* h)
{
    if (h->err_host && *(volatile int*)h->err_host != 0) {
        v3d_set_error("a lock-step SGM pass of an earlier call timed out (%d workgroups): its output was set to INVALID; "
                      "call v3d_sgbm_set_lockstep(h, 0) and recompute", *(volatile int*)h->err_host);
        return V3D_ERR_LOCKSTEP;
    }
    return V3D_OK;
}

// stages: 1 = cost volume, 2 = aggregation + WTA + LR check (raw), 3 = median + speckles (final)
static int run_sgbm(v3d_sgbm* h, const uint8_t* left, const uint8_t* right, int n, int W, int H, int pitch,
                    size_t frame_stride, int16_t* out, int last_stage, hipStream_t st)
{
    int rc = check_geometry(h, n, W, H, pitch);
    if (rc) return rc;
    if (!left || !right || !out) { v3d_set_error("null image pointer"); return V3D_ERR_ARG; }
    if ((rc = lockstep_state(h)) != V3D_OK) return rc;
    const int W1 = W - V3D_D;
    const int px = W * H;

    prof_mark(h, ST_PREFILTER, st);
    hipLaunchKernelGGL(k_prefilter, dim3(v3d_cdiv(W, 252), v3d_cdiv(H, PF_BAND), n), dim3(256), 0, st, left, right, W, H, pitch, frame_stride, h->ftzero, h->rec);
    prof_mark(h, ST_COST, st);
    constexpr int COST_OUT = CostGeo<V3D_COST_LPC>::OUT;
    hipLaunchKernelGGL((k_cost<V3D_COST_LPC>), dim3(v3d_cdiv(W1, COST_OUT), v3d_cdiv(H, h->cost_band), n), dim3(512), 0, st, h->rec, W, H, W1, h->cost_band, h->P2, h->C, h->cost_xcd);
    V3D_LAUNCH_CHECK();
    prof_mark(h, ST_V2, st);
    if (last_stage == 1) return V3D_OK;

    ChainArgs a;
    a.C = h->C; a.S = h->S; a.W1 = W1; a.H = H; a.W = W; a.nframes = n; a.P1 = h->P1; a.P2 = h->P2; a.uniq = h->uniq;
    a.wta = h->wta; a.xcd = h->hf_xcd; a.persist = 0; a.ticket = h->hf_ticket;
    // direction order is free (sums commute; saturation of non-negative addends is order-independent)
    const bool use_vdd = vdd_usable(h) && H < 4095;
    if (use_vdd) {
        // r1 + r2 + r3 in one lock-step pass (k_vdd); frames per launch bounded by co-residency
        // mapping: 4 disparities per lane (64-column strips) while the whole batch fits one co-resident launch, else
        // 8 per lane (128-column strips: ~30 % fewer instructions per element, twice the frames per launch)
        launch_vdd(h, n, W1, H, false, st);
        prof_mark(h, ST_D1, st);
        prof_mark(h, ST_D3, st);
        V3D_HIP_CHECK(hipEventRecord(h->vdd_done_ev, st));       // v3d_sgbm_stream_wait_lockstep: other streams may order behind the pass
        h->vdd_ev_recorded = true;
    } else {
    launch_chain<false, 0, false, 0>(h, a, st);         // r2: (x, y-1)
    prof_mark(h, ST_D1, st);
    launch_chain<false, 1, false, 1>(h, a, st);         // r1: (x-1, y-1)
    prof_mark(h, ST_D3, st);
    launch_chain<false, -1, false, 1>(h, a, st);        // r3: (x+1, y-1)
    }
    prof_mark(h, ST_H0, st);
    if (!h->hfused) launch_chain<true, 1, false, 1>(h, a, st);          // r0: (x-1, y)
    prof_mark(h, ST_V2R, st);
    if (h->prm.mode == V3D_MODE_HH) {
        if (use_vdd) {
            launch_vdd(h, n, W1, H, true, st);              // (x-1,y+1), (x,y+1), (x+1,y+1) in one bottom-up lock-step pass
            prof_mark(h, ST_D1R, st); prof_mark(h, ST_D3R, st);
            V3D_HIP_CHECK(hipEventRecord(h->vdd_done_ev, st));
        } else {
            launch_chain<false, 0, true, 1>(h, a, st);      // (x, y+1)
            prof_mark(h, ST_D1R, st);
            launch_chain<false, -1, true, 1>(h, a, st);     // (x+1, y+1)
            prof_mark(h, ST_D3R, st);
            launch_chain<false, 1, true, 1>(h, a, st);      // (x-1, y+1)
        }
    } else { prof_mark(h, ST_D1R, st); prof_mark(h, ST_D3R, st); }
    prof_mark(h, ST_H4_WTA, st);
    if (h->hfused) {                                    // r0 + r4 + WTA tail in one launch
        const dim3 g4(v3d_cdiv(v3d_cdiv(H, 4) * n, 4)), g8(v3d_cdiv(v3d_cdiv(H, 8) * n, 4));
        if (h->hsplit) {
            if (h->dpl == 4) { hipLaunchKernelGGL(k_hscan<4>, g4, dim3(256), 0, st, a, h->ckpt); hipLaunchKernelGGL((k_hfused<4, 2>), g4, dim3(256), 0, st, a, h->ckpt); }
            else { hipLaunchKernelGGL(k_hscan<8>, g8, dim3(256), 0, st, a, h->ckpt); hipLaunchKernelGGL((k_hfused<8, 2>), g8, dim3(256), 0, st, a, h->ckpt); }
        } else {
            dim3 l4 = g4, l8 = g8;
            if (h->hf_persist) {                            // resident waves only: 4 workgroups of 4 waves per CU (LDS / 119 VGPRs)
                a.persist = h->hf_persist;
                V3D_HIP_CHECK(hipMemsetAsync(h->hf_ticket, 0, sizeof(int), st));
                const unsigned res = (unsigned)h->ncu * 4u;
                if (l4.x > res) l4.x = res;
                if (l8.x > res) l8.x = res;
            }
            if (h->dpl == 4) hipLaunchKernelGGL((k_hfused<4, 3>), l4, dim3(256), 0, st, a, h->ckpt);
            else hipLaunchKernelGGL((k_hfused<8, 3>), l8, dim3(256), 0, st, a, h->ckpt);
            a.persist = 0;
        }
    } else
        launch_chain<true, -1, false, 2>(h, a, st);     // r4: (x+1, y), + WTA tail
    V3D_LAUNCH_CHECK();
    prof_mark(h, ST_LRCHECK, st);
    if (last_stage == 2) {
        if ((rc = launch_lrcheck_median(h->wta, W, H, n, h->d12, out, false, h->lrm_tiles, st)) != V3D_OK) return rc;
        if (use_vdd) hipLaunchKernelGGL(k_vdd_guard, dim3(256), dim3(256), 0, st, h->vdd_err, h->err_host, out, (size_t)px * n);
        V3D_LAUNCH_CHECK();
        prof_mark(h, ST_MEDIAN, st);
        return V3D_OK;
    }
    prof_mark(h, ST_MEDIAN, st);
    if ((rc = launch_lrcheck_median(h->wta, W, H, n, h->d12, out, true, h->lrm_tiles, st)) != V3D_OK) return rc;
    V3D_LAUNCH_CHECK();
    prof_mark(h, ST_SPECKLE, st);
    if (h->prm.speckleWindowSize > 0) {
        const int newVal = (h->prm.minDisparity - 1) * 16, maxDiff = 16 * h->prm.speckleRange, maxSize = h->prm.speckleWindowSize;
        rc = launch_speckles(out, W, H, n, newVal, maxSize, maxDiff, h->labels, st);
        if (rc) return rc;
    }
    if (use_vdd) {                                      // time-outs of this (or an earlier, uncleared) pass: poison `out`, raise the host flag
        hipLaunchKernelGGL(k_vdd_guard, dim3(256), dim3(256), 0, st, h->vdd_err, h->err_host, out, (size_t)px * n);
        V3D_LAUNCH_CHECK();
    }
    prof_mark(h, V3D_NSTAGE, st);
    if (h->prof_on && h->prof_calls < V3D_PROF_MAX_CALLS) h->prof_calls++;
    return V3D_OK;
}

// ---- per-stage timing: enable (resets the counters), run any number of compute calls, synchronise the
// stream, then read.  Events are recorded on the caller's stream between the stage launches. ----
// synchronises the device and returns how many k_vdd workgroups gave up waiting for a neighbour (0 = healthy)
extern "C" int v3d_sgbm_sync_errors(v3d_sgbm* h)
{
    if (!h) { v3d_set_error("null handle"); return V3D_ERR_ARG; }
    int e = 0;
    V3D_HIP_CHECK(hipSetDevice(h->device));
    V3D_HIP_CHECK(hipDeviceSynchronize());
    V3D_HIP_CHECK(hipMemcpy(&e, h->vdd_err, sizeof(int), hipMemcpyDeviceToHost));
    return e;
}

extern "C" int v3d_sgbm_set_lockstep(v3d_sgbm* h, int enable)
{
    if (!h) { v3d_set_error("null handle"); return V3D_ERR_ARG; }
    V3D_HIP_CHECK(hipSetDevice(h->device));
    V3D_HIP_CHECK(hipDeviceSynchronize());
    V3D_HIP_CHECK(hipMemset(h->vdd_err, 0, sizeof(int)));
    *(volatile int*)h->err_host = 0;
    h->vdd_mode = enable ? 1 : 0;
    return V3D_OK;
}

// non-blocking: > 0 once k_vdd_guard of a finished call has seen time-outs (the flag travels with the call's last launch)
extern "C" int v3d_sgbm_poll_errors(const v3d_sgbm* h)
{
    if (!h) { v3d_set_error("null handle"); return V3D_ERR_ARG; }
    return h->err_host ? *(volatile int*)h->err_host : 0;
}

// make `stream` wait until the lock-step pass of the most recent compute call on this handle has finished (no-op if
// none ran).  For hosts that run a collective on a side stream: RCCL workgroups that land on a CU take a slot the
// co-resident pass was sized with, so order the collective BEHIND the pass (it then overlaps the horizontal pass,
// the post-filters and the upscale instead) and the next compute call behind the collective.
extern "C" int v3d_sgbm_stream_wait_lockstep(v3d_sgbm* h, void* stream)
{
    if (!h) { v3d_set_error("null handle"); return V3D_ERR_ARG; }
    if (!h->vdd_ev_recorded) return V3D_OK;
    V3D_HIP_CHECK(hipStreamWaitEvent((hipStream_t)stream, h->vdd_done_ev, 0));
    return V3D_OK;
}

extern "C" int v3d_sgbm_profile(v3d_sgbm* h, int enable)
{
    if (!h) { v3d_set_error("null handle"); return V3D_ERR_ARG; }
    if (enable && h->prof_ev.empty()) {
        V3D_HIP_CHECK(hipSetDevice(h->device));
        h->prof_ev.resize((size_t)V3D_PROF_MAX_CALLS * (V3D_NSTAGE + 1));
        for (auto& e : h->prof_ev) V3D_HIP_CHECK(hipEventCreate(&e));
    }
    h->prof_on = enable != 0;
    h->prof_calls = 0;
    return V3D_OK;
}
extern "C" int v3d_sgbm_profile_stage_count(void) { return V3D_NSTAGE; }
extern "C" const char* v3d_sgbm_profile_stage_name(int i) { return (i >= 0 && i < V3D_NSTAGE) ? g_stage_names[i] : ""; }
// total_ms[i] = summed duration of stage i over the recorded calls; returns the number of calls (or < 0)
extern "C" int v3d_sgbm_profile_read(v3d_sgbm* h, double* total_ms, int n)
{
    if (!h || !total_ms || n < V3D_NSTAGE) { v3d_set_error("bad argument"); return V3D_ERR_ARG; }
    for (int i = 0; i < V3D_NSTAGE; i++) total_ms[i] = 0.0;
    for (int c = 0; c < h->prof_calls; c++)
        for (int i = 0; i < V3D_NSTAGE; i++) {
            float ms = 0.f;
            const hipEvent_t a = h->prof_ev[(size_t)c * (V3D_NSTAGE + 1) + i], b = h->prof_ev[(size_t)c * (V3D_NSTAGE + 1) + i + 1];
            V3D_HIP_CHECK(hipEventElapsedTime(&ms, a, b));
            total_ms[i] += ms;
        }
    return h->prof_calls;
}

extern "C" int v3d_sgbm_compute(v3d_sgbm* h, const uint8_t* l, const uint8_t* r, int W, int H, int pitch, int16_t* out, void* stream)
{
    return run_sgbm(h, l, r, 1, W, H, pitch, 0, out, 3, (hipStream_t)stream);
}
extern "C" int v3d_sgbm_compute_batch(v3d_sgbm* h, const uint8_t* l, const uint8_t* r, int n, int W, int H, int pitch,
                                      size_t frame_stride, int16_t* out, void* stream)
{
    return run_sgbm(h, l, r, n, W, H, pitch, frame_stride, out, 3, (hipStream_t)stream);
}
extern "C" int v3d_sgbm_debug_cost_volume(v3d_sgbm* h, const uint8_t* l, const uint8_t* r, int W, int H, int pitch, int16_t* C_out, void* stream)
{
    int16_t dummy;
    int rc = run_sgbm(h, l, r, 1, W, H, pitch, 0, &dummy, 1, (hipStream_t)stream);
    if (rc) return rc;
    // the export is int16 [H][W-64][64] with P2 folded in, whatever the storage form
    const size_t npx = (size_t)(W - V3D_D) * H;
    hipLaunchKernelGGL(k_c_export, dim3((unsigned)((npx * 8 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, h->C, npx, h->P2, C_out);
    V3D_LAUNCH_CHECK();
    return V3D_OK;
}
extern "C" int v3d_sgbm_debug_raw(v3d_sgbm* h, const uint8_t* l, const uint8_t* r, int W, int H, int pitch, int16_t* out, int16_t* S_out, void* stream)
{
    int rc = run_sgbm(h, l, r, 1, W, H, pitch, 0, out, 2, (hipStream_t)stream);
    if (rc) return rc;
    // S holds sum of all directions but the last (the last one is only ever formed on-chip)
    if (S_out) V3D_HIP_CHECK(hipMemcpyAsync(S_out, h->S, (size_t)(W - V3D_D) * H * V3D_D * sizeof(int16_t), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return V3D_OK;
}

extern "C" int v3d_median3x3_i16(const int16_t* src, int W, int H, int16_t* dst, void* stream)
{
    if (!src || !dst || W < 1 || H < 1) { v3d_set_error("bad argument"); return V3D_ERR_ARG; }
    hipLaunchKernelGGL(k_median3x3, dim3(v3d_cdiv(W, 256), H, 1), dim3(256), 0, (hipStream_t)stream, src, W, H, dst);
    V3D_LAUNCH_CHECK();
    return V3D_OK;
}

extern "C" int v3d_filter_speckles(int16_t* img, int W, int H, int newVal, int maxSize, int maxDiff, int32_t* ws, void* stream)
{
    if (!img || !ws || W < 1 || H < 1) { v3d_set_error("bad argument"); return V3D_ERR_ARG; }
    return launch_speckles(img, W, H, 1, newVal, maxSize, maxDiff, ws, (hipStream_t)stream);
}
