/*
 * v3d_oracle.h -- CPU restatement of the per-frame hot path (TEST INFRASTRUCTURE ONLY).
 *
 * This library is the parity oracle and the "port" CPU baseline.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the product
 * (video_3d_pipeline + libv3d_hip.so) never does.
 *
 * PARITY UNPINNED: the arithmetic of this path lives in opencv-python 4.11.0.86
 * (reference uv.lock:1164-1165), which is neither vendored under /root/reference nor
 * installed in the build container, and the reference holds no tests/golden vectors
 * (SURVEY.md section 4, 8c).  Everything below restates the *published* OpenCV
 * algorithms (calib3d StereoSGBM MODE_SGBM / MODE_HH, imgproc resize INTER_LANCZOS4,
 * cvtColor, medianBlur, filterSpeckles) as the reference invokes them:
 *   depth.py:250-268  split_sbs_frame      -> orc_sbs_to_gray (Lanczos4 unsqueeze)
 *   depth.py:274-275, 337-338  cvtColor    -> orc_sbs_to_gray (BGR->gray)
 *   depth.py:315-325, 341  StereoSGBM      -> orc_sgbm_compute
 *   depth.py:341, 374      /16, clamp      -> orc_disp_to_depth
 *   depth.py:344-374       DPT blend       -> orc_mono_blend (resize INTER_LINEAR + min-max + 0.7/0.3 + clamp)
 *   depth.py:397-406       save_depth_map  -> orc_depth_to_u16
 *   upscale.py:21-73 (re-specified as a guided filter) -> orc_guided_upscale
 *   CREStereo-style group correlation (no reference code) -> orc_corr_lookup
 */
#ifndef V3D_ORACLE_H
#define V3D_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    int minDisparity;      /* must be 0 */
    int numDisparities;    /* multiple of 16 */
    int blockSize;         /* odd */
    int P1, P2;
    int disp12MaxDiff;
    int preFilterCap;
    int uniquenessRatio;
    int speckleWindowSize;
    int speckleRange;
    int mode;              /* 0 = MODE_SGBM (5 paths), 1 = MODE_HH (8 paths) */
} orc_sgbm_params;

/* the parameter block depth.py:315-325 builds */
void orc_sgbm_default_params(orc_sgbm_params* p);

/* full StereoSGBM::compute: raw SGM + median 3x3 + speckle filter.  disp16: H*W int16 (x16, -16 invalid) */
int orc_sgbm_compute(const orc_sgbm_params* p, const uint8_t* left, const uint8_t* right,
                     int W, int H, int16_t* disp16);

/* stage exports (for kernel-by-kernel parity) */
/* cost volume C[y][xr][d] (P2 folded in), xr in [0, W-D) */
int orc_sgbm_cost_volume(const orc_sgbm_params* p, const uint8_t* left, const uint8_t* right,
                         int W, int H, int16_t* C);
/* raw disparity after WTA/uniqueness/subpixel/LR-check, before median+speckle.
   S_out (optional, may be NULL): final aggregated S[y][xr][d] */
int orc_sgbm_raw(const orc_sgbm_params* p, const uint8_t* left, const uint8_t* right,
                 int W, int H, int16_t* disp16, int16_t* S_out);
void orc_median3x3_i16(const int16_t* src, int W, int H, int16_t* dst);
void orc_filter_speckles(int16_t* img, int W, int H, int newVal, int maxSpeckleSize, int maxDiff);

/* depth.py:250-268 + 274-275 + 337-338: SBS BGR u8 [H][W][3] -> left/right gray u8.
   unsqueeze!=0: each half (W/2 wide) is Lanczos4-resized to W wide (out W x H);
   unsqueeze==0: out (W/2) x H.  returns 0, or -1 if W is odd. */
int orc_sbs_to_gray(const uint8_t* sbs_bgr, int W, int H, int unsqueeze, uint8_t* L, uint8_t* R);
/* the BGR halves themselves (what split_sbs_frame returns) */
int orc_split_sbs(const uint8_t* sbs_bgr, int W, int H, int unsqueeze, uint8_t* Lbgr, uint8_t* Rbgr);
void orc_bgr_to_gray(const uint8_t* bgr, int n, uint8_t* gray);
/* the 8 int16 Lanczos taps (scale 2^11) for fractional offset fx */
void orc_lanczos4_taps(float fx, int16_t taps[8]);

/* depth.py:341 + 374 */
void orc_disp_to_depth(const int16_t* disp16, int n, float* out);
/* depth.py:397-406 */
void orc_depth_to_u16(const float* depth, int n, uint16_t* out);

/* guided-filter joint upsampling, float64.
   depth_lo: [Hlo][Wlo] f32; guide: [Hhi][Whi] u8 luma; out: [Hhi][Whi] f64 */
int orc_guided_upscale(const float* depth_lo, int Wlo, int Hlo, const uint8_t* guide, int Whi, int Hhi,
                       int r, double eps, double* out);
/* the bilinear resample alone (align_corners=False, edge clamp) */
void orc_bilinear_resize(const float* src, int Ws, int Hs, int Wd, int Hd, double* dst);

/* depth.py:344-374: cv2.resize(mono) INTER_LINEAR float32, min-max to [0, 64], w_stereo*disp + w_mono*mono, clamp <= 0 */
void orc_resize_linear_f32(const float* src, int Ws, int Hs, int Wd, int Hd, float* dst);
void orc_mono_blend(const int16_t* disp16, int W, int H, const float* mono, int mw, int mh,
                    float w_stereo, float w_mono, float* out);

/* CREStereo-style local group correlation, fp32 (SURVEY Appendix B.2 form A).
   fl, fr: [C][h][w] f32; flow: [2][h][w] f32 (x then y); out: [G*9][h][w] f32.
   C must be divisible by G; pattern 0: 1x9 (dx=-4..4), pattern 1: 3x3. */
int orc_corr_lookup(const float* fl, const float* fr, const float* flow, int C, int h, int w,
                    int G, int pattern, float* out);

#ifdef __cplusplus
}
#endif
#endif
