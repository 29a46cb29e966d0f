/*
 * v3d_hip.h -- C ABI of libv3d_hip.so: the MI355X (gfx950) implementation of the per-frame
 * hot path of video_3d_pipeline (SBS frame -> disparity -> 4K depth).
 *
 * The reference has no FFI of its own (pure Python calling OpenCV / ffmpeg); each entry point
 * below replaces one OpenCV / NumPy / ffmpeg call site of the reference and is what a ctypes
 * binding in the reference's depth.py / upscale.py would bind:
 *
 *   v3d_sbs_to_gray        depth.py:250-268 split_sbs_frame (cv2.resize INTER_LANCZOS4) +
 *                          depth.py:274-275, 337-338 cvtColor BGR->RGB->GRAY
 *   v3d_sgbm_create        depth.py:315-325 cv2.StereoSGBM_create(...)
 *   v3d_sgbm_compute[_batch]  depth.py:341 stereo.compute(left_gray, right_gray) -> int16 x16
 *   v3d_disp_to_depth      depth.py:341 .astype(float32)/16.0 and depth.py:374 clamp <=0 -> 0
 *   v3d_mono_blend         depth.py:344-374 the "hybrid" blend: cv2.resize(mono) INTER_LINEAR, min-max to [0, 64],
 *                          0.7 * disparity + 0.3 * mono, clamp <= 0 -> 0 (the mono map comes from the host: DPT or any provider)
 *   v3d_depth_to_u16       depth.py:397-406 save_depth_map min-max normalisation to uint16
 *   v3d_guided_upscale     upscale.py:21-73 upscale_depth_maps_ffmpeg (`scale` filter), re-specified
 *                          as guided-filter joint upsampling
 *   v3d_corr_lookup        CREStereo recurrent correlation lookup (BASELINE.json config 4; the
 *                          reference only names it: depth.py:1, CREStereo_model.txt)
 *
 * Conventions
 *  - every image/volume pointer is a DEVICE pointer owned by the caller (e.g. a torch tensor's
 *    data_ptr()); nothing here allocates on the steady-state path: workspaces belong to the handle
 *    and are sized at create time;
 *  - `stream` is a hipStream_t passed as void*; calls enqueue work and do NOT synchronise;
 *  - return 0 on success, negative on error; v3d_last_error() returns the thread-local message;
 *  - a handle is bound to one device and is not thread-safe.
 */
#ifndef V3D_HIP_H
#define V3D_HIP_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define V3D_OK 0
#define V3D_ERR_ARG (-1)
#define V3D_ERR_HIP (-2)
#define V3D_ERR_UNSUPPORTED (-3)
#define V3D_ERR_LOCKSTEP (-4)   /* a lock-step SGM pass timed out on an over-subscribed GPU: see v3d_sgbm_set_lockstep */

#define V3D_MODE_SGBM 0   /* 5 paths, single pass: cv2.STEREO_SGBM_MODE_SGBM (the reference's default) */
#define V3D_MODE_HH   1   /* 8 paths, two passes: cv2.STEREO_SGBM_MODE_HH */

typedef struct v3d_sgbm v3d_sgbm;

/* mirrors the keyword arguments of cv2.StereoSGBM_create (depth.py:315-325) */
typedef struct {
    int minDisparity;       /* must be 0 */
    int numDisparities;     /* must be 64 in this build */
    int blockSize;          /* must be 5 in this build */
    int P1, P2;
    int disp12MaxDiff;
    int preFilterCap;
    int uniquenessRatio;
    int speckleWindowSize;
    int speckleRange;
    int mode;               /* V3D_MODE_SGBM / V3D_MODE_HH */
} v3d_sgbm_params;

/* fills the parameter block depth.py:315-325 uses */
void v3d_sgbm_default_params(v3d_sgbm_params* p);

/* create a matcher whose workspaces hold up to max_batch frames of max_width x max_height */
int v3d_sgbm_create(const v3d_sgbm_params* params, int device, int max_width, int max_height,
                    int max_batch, v3d_sgbm** out);
void v3d_sgbm_destroy(v3d_sgbm* h);
/* bytes of device workspace the handle owns */
size_t v3d_sgbm_workspace_bytes(const v3d_sgbm* h);

/* one frame: left/right gray u8 [H][pitch], disp16 out int16 [H][W] (value x16, -16 invalid) */
int v3d_sgbm_compute(v3d_sgbm* h, const uint8_t* left_gray, const uint8_t* right_gray,
                     int W, int H, int pitch, int16_t* disp16_out, void* stream);
/* n frames: frame f at left_gray + f*frame_stride (bytes), output frame f at disp16_out + f*W*H */
int v3d_sgbm_compute_batch(v3d_sgbm* h, const uint8_t* left_gray, const uint8_t* right_gray,
                           int n, int W, int H, int pitch, size_t frame_stride,
                           int16_t* disp16_out, void* stream);

/* Lock-step pass and an over-subscribed GPU.  The three top-down SGM paths run as ONE pass whose workgroups must all
   be resident together (sized from the occupancy query at create time).  If another process or stream holds CUs, a
   workgroup's bounded wait for its neighbour strip gives up.  The library then NEVER hands out those disparities:
     - the last launch of the call sets every output pixel of the call to INVALID (-16) and raises a host-visible flag;
     - every later v3d_sgbm_compute* on the handle returns V3D_ERR_LOCKSTEP until the host reacts;
     - v3d_sgbm_poll_errors (no synchronisation) / v3d_sgbm_sync_errors (device synchronise) return the number of
       workgroups that timed out since the state was last cleared (0 = healthy);
     - v3d_sgbm_set_lockstep(h, 0) synchronises, clears the state and makes later calls use one launch per direction
       (same bits, ~2x the SGM time); (h, 1) clears and keeps the lock-step pass.  Then recompute the batch.
   Both paths are GPU paths; there is no CPU fallback. */
int v3d_sgbm_sync_errors(v3d_sgbm* h);
int v3d_sgbm_poll_errors(const v3d_sgbm* h);
int v3d_sgbm_set_lockstep(v3d_sgbm* h, int enable);
/* make `stream` (hipStream_t) wait until the lock-step pass of the latest compute call on `h` has finished.  A host
   that runs a collective (RCCL) on a side stream orders it behind the pass with this call, and the next compute call
   behind the collective: the collective's workgroups then never take CU slots the pass was sized with. */
int v3d_sgbm_stream_wait_lockstep(v3d_sgbm* h, void* stream);

/* Tuning switches of a handle (defaults = the measured best; results never change): "lockstep" 0/1, "hfused" 0/1,
   "chain_dpl" 4/8, "hsplit" 0/1, "hf_persist" 0/1, "lrm_tiles" 0/1, "vdd_dpl" 0/4/8, "cost_band" >= 8, "cost_xcd" / "vdd_xcd" / "hf_xcd" 0/1, "reserve_cus" (CUs
   other streams keep busy during a lock-step pass), "vdd_spin_limit" (poll rounds per lane; 0 = derived from the row
   count), "vdd_launch_frames" (frames per lock-step launch; 0 = sized from the occupancy query and the call's width).
   get also knows the read-only "vdd_frames_per_launch_dpl4" / "_dpl8".  Unknown key or bad value: V3D_ERR_ARG.
   The library reads no environment variables. */
int v3d_sgbm_set_option(v3d_sgbm* h, const char* key, int value);
int v3d_sgbm_get_option(const v3d_sgbm* h, const char* key, int* value);
/* library-wide switches of the handle-less entry points: "gf_fused" 1/0 (guided filter as one launch with a/b kept in LDS /
   as two sweeps with a/b through HBM: same bits), "gf_band" (rows per workgroup of the fused kernel, default 432; 0 = chosen per launch from its round count -- then a frame's last bit may depend on the batch size), "gf_cols" 256/512 (its
   strip width), "gf_int1" 1/0 (int16 disparity + exact 2x: stage 1 of the fused kernel in exact integers / in f64: same bits), "gf_band1", "gf_band2" (rows per workgroup of the two sweeps), "gf_tiled" 0/1 (force the LDS-tiled guided
   kernel), "corr_fused" 1/0 (1x9 correlation as one gather-GEMM launch with the warped features staged in LDS / as a warp
   kernel + a GEMM kernel: same bits), "corr_gather" 0/1 (register-only gather-GEMM) */
int v3d_set_option(const char* key, int value);
int v3d_get_option(const char* key, int* value);

/* per-stage HIP-event timing on the caller's stream (what bench.py's `roofline` object reads):
   v3d_sgbm_profile(h, 1) resets and enables; run compute calls; synchronise the stream;
   v3d_sgbm_profile_read fills total_ms[stage] (n >= v3d_sgbm_profile_stage_count()) and returns the
   number of recorded calls.  Only full v3d_sgbm_compute[_batch] calls are recorded. */
int v3d_sgbm_profile(v3d_sgbm* h, int enable);
int v3d_sgbm_profile_stage_count(void);
const char* v3d_sgbm_profile_stage_name(int stage);
int v3d_sgbm_profile_read(v3d_sgbm* h, double* total_ms, int n);

/* stage exports used by the parity tests (same inputs as v3d_sgbm_compute, one frame) */
/* cost volume C[y][x-64][d] int16, P2 folded in */
int v3d_sgbm_debug_cost_volume(v3d_sgbm* h, const uint8_t* left_gray, const uint8_t* right_gray,
                               int W, int H, int pitch, int16_t* C_out, void* stream);
/* raw disparity before median/speckle (after WTA, uniqueness, sub-pixel, L-R check) and,
   if S_out != NULL, the aggregated volume S[y][x-64][d] */
int v3d_sgbm_debug_raw(v3d_sgbm* h, const uint8_t* left_gray, const uint8_t* right_gray,
                       int W, int H, int pitch, int16_t* disp16_out, int16_t* S_out, void* stream);
int v3d_median3x3_i16(const int16_t* src, int W, int H, int16_t* dst, void* stream);
/* labels_ws: device scratch of 3*W*H int32 */
int v3d_filter_speckles(int16_t* img, int W, int H, int newVal, int maxSpeckleSize, int maxDiff,
                        int32_t* labels_ws, void* stream);

/* SBS BGR u8 [H][pitch] (W*3 payload bytes per row) -> gray left/right.
   unsqueeze != 0: outputs are W x H (Lanczos4 x2 horizontal); else (W/2) x H. W must be even. */
int v3d_sbs_to_gray(const uint8_t* sbs_bgr, int W, int H, int pitch, int unsqueeze,
                    uint8_t* left_gray, uint8_t* right_gray, void* stream);
/* n frames in one launch: frame f at sbs_bgr + f*frame_stride bytes; outputs packed [n][H][outW] */
int v3d_sbs_to_gray_batch(const uint8_t* sbs_bgr, int n, int W, int H, int pitch, size_t frame_stride,
                          int unsqueeze, uint8_t* left_gray, uint8_t* right_gray, void* stream);
/* the BGR halves themselves (split_sbs_frame's return value), [H][outW][3] */
int v3d_split_sbs(const uint8_t* sbs_bgr, int W, int H, int pitch, int unsqueeze,
                  uint8_t* left_bgr, uint8_t* right_bgr, void* stream);

int v3d_disp_to_depth(const int16_t* disp16, size_t n, float* depth_out, void* stream);
/* minmax_ws: device scratch of >= 2 floats */
int v3d_depth_to_u16(const float* depth, size_t n, uint16_t* out, float* minmax_ws, void* stream);

/* the upscaled depth as the 16-bit sample the PNG sink stores (stands where upscale.py:47-59 hands gray16 frames to the
   encoder): out = clamp(rint(depth), 0, 65535), round-half-to-even */
int v3d_round_to_u16(const float* depth, size_t n, uint16_t* out, void* stream);

/* depth.py:344-374: depth_out[H][W] = clamp0(w_stereo * disp16/16 + w_mono * (resize(mono) - min) / (max - min) * 64), float32
   arithmetic in the reference's order (bit-identical to the NumPy expression); max == min leaves the stereo disparity.
   mono: f32 [mh][mw] of any size (cv2.resize INTER_LINEAR semantics; same size = no resize).  depth.py uses 0.7 / 0.3.
   ws: device scratch of v3d_mono_blend_ws_bytes(n) bytes.  Batch: frame f at disp16 + f*W*H, mono + f*mono_stride (floats),
   depth_out + f*W*H. */
size_t v3d_mono_blend_ws_bytes(int n);
int v3d_mono_blend(const int16_t* disp16, int W, int H, const float* mono, int mw, int mh,
                   float w_stereo, float w_mono, float* depth_out, void* ws, void* stream);
int v3d_mono_blend_batch(const int16_t* disp16, int n, int W, int H, const float* mono, int mw, int mh, size_t mono_stride,
                         float w_stereo, float w_mono, float* depth_out, void* ws, void* stream);

/* guided-filter joint upsampling: depth_lo f32 [Hlo][Wlo], guide u8 luma [Hhi][Whi] -> out f32 [Hhi][Whi].
   ws: device scratch of v3d_guided_upscale_ws_bytes(Whi, Hhi) bytes */
size_t v3d_guided_upscale_ws_bytes(int Whi, int Hhi);
int v3d_guided_upscale(const float* depth_lo, int Wlo, int Hlo, const uint8_t* guide, int Whi, int Hhi,
                       int r, float eps, float* out, void* ws, void* stream);
/* n frames in one launch: frame f at depth_lo + f*depth_stride (floats), guide + f*guide_stride (bytes),
   out + f*Whi*Hhi; ws must hold n * v3d_guided_upscale_ws_bytes(Whi, Hhi) bytes */
int v3d_guided_upscale_batch(const float* depth_lo, int Wlo, int Hlo, size_t depth_stride, const uint8_t* guide,
                             int Whi, int Hhi, size_t guide_stride, int n, int r, float eps, float* out,
                             void* ws, void* stream);
/* the same filter fed with the matcher's int16 disparity (x16, <= 0 = invalid): depth.py:341 `.astype(float32)/16.0` and
   depth.py:374 `disparity[disparity <= 0] = 0` happen as the values are loaded, so the stereo-only pipeline
   (sgbm -> upscale) never writes or re-reads the float32 depth plane.  Bit-identical to v3d_disp_to_depth followed by
   v3d_guided_upscale_batch.  Frame f at disp16 + f*disp_stride (int16 elements).
   Domain: disparities up to 16 * 113 = 1820 (this build's matcher, numDisparities = 64, never exceeds 1023).  For an exact 2x
   upscale the filter's first stage then runs in exact int32 sums (sum of g * 256 p over a window stays below 2^31); a caller
   feeding larger values must switch that off first: v3d_set_option("gf_int1", 0). */
int v3d_guided_upscale_disp16_batch(const int16_t* disp16, int Wlo, int Hlo, size_t disp_stride, const uint8_t* guide,
                                    int Whi, int Hhi, size_t guide_stride, int n, int r, float eps, float* out,
                                    void* ws, void* stream);
/* BGR [H][W][3] u8 -> luma u8 with the same weights as cvtColor */
int v3d_bgr_to_gray(const uint8_t* bgr, size_t n_pixels, uint8_t* gray, void* stream);

/* CREStereo-style local group correlation on the matrix cores.
   fl, fr: bf16 [h][w][C] (channel-last), flow: f32 [2][h][w], out: f32 [G*9][h][w];
   C = 64*G; pattern 0 = 1x9, 1 = 3x3.  ws: scratch of v3d_corr_ws_bytes(C,h,w) bytes */
size_t v3d_corr_ws_bytes(int C, int h, int w);
int v3d_corr_lookup(const uint16_t* fl_bf16, const uint16_t* fr_bf16, const float* flow,
                    int C, int h, int w, int G, int pattern, float* out, void* ws, void* stream);

const char* v3d_last_error(void);
const char* v3d_version(void);

#ifdef __cplusplus
}
#endif
#endif
