/*
 * cnm_engine.h -- C ABI of the MI355X (gfx950) CNMNet depth engine.
 *
 * Drop-in boundary for ONE hot path of xxlong0/CNMNet: plane-sweep warp + L1 cost
 * volume, the encoder/decoder inverse-depth regression, the two-view fusion net,
 * depth->normal and the depth-based inverse warp.  The reference has no FFI of its
 * own (it is pure Python on torch ops); each entry point below names the reference
 * function (file:line under the reference checkout) whose arithmetic it replaces.
 * The Python shim in cnmnet_amd/ binds these with ctypes.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in _host;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); all work is
 *     enqueued on it, nothing synchronises, nothing allocates;
 *   - the caller owns every buffer, including workspaces (sizes via *_floats());
 *   - return value: CNM_OK (0) or a negative cnm_status; never throws;
 *   - re-entrant: the data path keeps no process-global mutable state (contrast the
 *     reference's module global pixel_coords, depthnet/inverse_warp.py:5); every buffer,
 *     workspace and stream comes from the caller, so DataParallel worker threads (one per
 *     device / stream) may call concurrently.  Two things ARE process-wide, both outside the
 *     data path: the cnm_tune_* knobs (DEBUG / A-B switches for benches and tests: set them
 *     before the first forward and not while other threads are inside the engine; the
 *     defaults are the product) and the engine status word behind cnm_engine_status().
 *
 * Activation layout inside the engine ("c4"): [N][G][H][W][4] floats, channel
 * c = 4*g + j, G = ceil(C/4), padded channels are zero.  A c4 view is described by
 * (base pointer, G_total = groups per image in the underlying buffer, g0 = first
 * group of the view), which lets producers write straight into channel slices of a
 * consumer's concatenated input (torch.cat at depthNet_model.py:233,242,245,250,255,260
 * is never materialised as a copy).
 */
#ifndef CNM_ENGINE_H
#define CNM_ENGINE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CNM_ABI_VERSION 6                    /* 6 [r6]: cnm_layer_weights gains u4q (the four-wave F(4x4,3x3) kernel's filter); 5 [r5]: engine status per device; CNM_ERR_LAUNCH is sticky per device after a hand-off time-out; sync-workspace flag
                                               words are no longer guaranteed zero after a FAILED call (stale generations, harmless to later launches on the
                                               same queue; callers that acknowledge a failure should zero their sync workspaces, as cnmnet_amd.ops.engine_status
                                               does); _cpu host twins, cnm_engine_status, cnm_tune_sync_spin_limit exported since 4 */
#define CNM_WINO4_MIN_WORKGROUPS 384
#define CNM_UPSAMPLED_MIN_PIXELS 196608      /* 16 images x 96 x 128 */
#define CNM_UPSAMPLED_MIN_PIXELS_F16 262144  /* [r6] the fp16 engine fuses from MORE output pixels on: 16 x 96 x 128 runs unfused there */
/* Engine status.  The persistent stream-K convolution kernels hand partial outputs from one workgroup to another inside a
 * launch; a hand-off that does not complete within its spin bound (it cannot, unless workgroups of one launch are not
 * co-resident for seconds) does not hang and does not pass silently: the kernel records it in a pinned host word, the
 * outputs of that launch are wrong, and every later call of a staged-kernel entry point ON THAT DEVICE returns CNM_ERR_LAUNCH
 * without launching until the failure has been acknowledged (one status word per device [r5]: other devices of the process keep
 * working).  cnm_engine_status(clear) speaks for the CURRENT device: CNM_OK or CNM_ERR_LAUNCH (a hand-off timed out there since
 * the last clear) and, with clear != 0, acknowledges it; it also allocates the device's status word if no launch has yet, so a
 * caller that queries the status once before capturing a hipGraph gets time-outs inside replays reported.  It reads host memory only: synchronise
 * the stream first when the launch in question may still be running.  The sync workspaces need no repair afterwards
 * (a flag counts only if it carries the generation -- the dispatch id -- of the launch that polls it, csrc/sync_ws.h).
 * Replaying a captured hipGraph runs none of these entry points, so nothing refuses there: call cnm_engine_status() after
 * synchronising a replay (cnmnet_amd/trainer.py does at the step's loss read-back, bench.py after its timed region). */
int cnm_engine_status(int clear);
/* DEBUG / TEST: writes the hand-off generation the kernels of ONE launch see (the queue's dispatch id, made odd) to out[0 .. nblocks-1],
 * one value per workgroup of a probe launch on `stream`: the same for every workgroup of a launch, different from launch to launch,
 * eager or hipGraph replay (tests/test_gpu_parity.py::test_sync_generation_differs_from_launch_to_launch). */
int cnm_debug_sync_generation(unsigned* out, int nblocks, void* stream);
/* DEBUG / TEST ONLY: polls before a hand-off gives up (0 = the default, 2^24: about five seconds; looked at every 4096 polls);
 * bit 31 injects the fault the bound exists for (every wait fails at once).  Returns the previous value. */
unsigned cnm_tune_sync_spin_limit(unsigned polls);

/* Tuning knobs -- DEBUG / A-B switches, process-wide (see "re-entrant" above).  Each returns the previous value.
 * wino4_min_workgroups: workgroup count from which the fp32 executors prefer the F(4x4,3x3) kernel; n <= 0 only queries.
 * refine_side_stream: 1 (default) runs DepthRefineNet's probability decoder on an engine-owned side stream, forked from
 *   and joined back into the caller's stream with events (the two decoders of depthNet_model.py:341-365 are independent);
 *   0 keeps every launch on the caller's stream; on < 0 only queries. */
int cnm_tune_wino4_min_workgroups(int n);
int cnm_tune_refine_side_stream(int on);
/* upsampled_min_pixels: an up_conv_layer (bilinear x2 + 3x3, depthNet_model.py:89-112) runs as ONE fused pass over its
 *   low-resolution input (cnm_conv3x3_upsampled_winograd4_c4_f32 + ring pass) when its output has at least this many
 *   pixels over the batch and at most 256 input channels -- below that the ring pass costs more than the upsampled
 *   tensor's round trip; n <= 0 only queries, INT_MAX switches the fused path off. */
int cnm_tune_upsampled_min_pixels(int n);
int cnm_tune_upsampled_min_pixels_f16(int n);   /* [r6] the same switch of the fp16 engine (default CNM_UPSAMPLED_MIN_PIXELS_F16) */
/* glds_tile: tile (couts x pixels) of the LDS-DMA implicit-GEMM kernel: 0 (default) = chosen per layer from its
 *   workgroup count, stride and reduction depth; 1 .. 5 force 128x256 / 64x512 / 64x128 / 128x512 / 256x256 (falling back
 *   to 64x512 when Cout is not a multiple of 128, to 128x256 when 256x256 does not divide it); any other n only queries. */
int cnm_tune_glds_tile(int n);
/* gldsx: 1 (default) = fp16 stride-1 convolutions whose shape qualifies (W a power of two in 32 .. 256, H W a multiple of 256, at
 *   most one left-over channel group beyond a multiple of 8) run on the row-extended kernel, which stages the pixel operand of a
 *   filter row once per k x k filter row instead of once per tap; 0 = the tap-by-tap LDS-DMA kernel everywhere (A/B); a forced
 *   glds_tile also selects the latter.  Returns the previous value. */
int cnm_tune_gldsx(int n);
/* wino36_staged: 1 (default) lets the F(4x4,3x3) entry points (plain, concatenated input, fused up_conv) run the
 *   LDS-staged persistent kernel (conv_winograd4s.hip: 128 output channels x 16 tiles per workgroup, input patch by
 *   LDS-DMA) where the output-channel count is a multiple of 128, the image is at least six tiles wide and the units fill
 *   whole rounds of one workgroup per CU; 2 runs it wherever the shape is eligible; 0 keeps the gather-fed kernel
 *   everywhere.  Results are bit-identical in all three settings; any other value only queries. */
int cnm_tune_wino36_staged(int on);
/* rows_wide: the row-wise Winograd kernels with four outputs per tile (conv1.0 F(4,7), the stride-2 column-phase layers) run
 *   128 output channels per workgroup -- eight waves sharing one set of transformed tiles, one workgroup per CU -- for the
 *   layer where that was measured faster (7x7 stride 2 with Cout a multiple of 128 and whole rounds of workgroups: 1,
 *   default), wherever Cout % 128 == 0 (2), or never (0: 64 output channels, two workgroups per CU).  Bit-identical
 *   results; any other value only queries. */
int cnm_tune_rows_wide(int on);
/* wino4_small: 1 (default) lets the fp32 executors run 3x3 stride-1 layers below the wino4_min_workgroups switch on the
 *   staged F(4x4,3x3) kernel too (Cout a multiple of 128, at least 3 x 3 tiles per image: 4 x 4 tile blocks) -- with the
 *   executors' sync workspace it spreads any unit count evenly over the CUs; 0 keeps F(2x2,3x3) there.  Other values query. */
int cnm_tune_wino4_small(int on);

typedef enum cnm_status {
    CNM_OK = 0,
    CNM_ERR_BAD_ARG = -1,       /* null pointer, non-positive size, unsupported parameter */
    CNM_ERR_BAD_SHAPE = -2,     /* H/W not a multiple of 32 for the nets (reference: torch.cat fails) */
    CNM_ERR_BAD_SCALE = -3,     /* idepth_scale not 2.0 or 3.0 (reference: UnboundLocalError, depthNet_model.py:186-191) */
    CNM_ERR_LAUNCH = -4,        /* hipGetLastError() != hipSuccess after a launch, or an unacknowledged device-side failure (cnm_engine_status) */
    CNM_ERR_WORKSPACE = -5      /* workspace smaller than *_workspace_floats() */
} cnm_status;

int cnm_abi_version(void);
const char* cnm_status_string(int status);

/* ---------------------------------------------------------------- geometry prep
 * Replaces get_pixel_coordinates + process_camera_parameters
 * (depthnet/depth_util.py:13-56): per (ref,src) pair the 3x3 homography factor
 * Hm = K_r R K_l^-1 and KT = K_r T, 12 floats; the [B,3,H*W] KRKiUV tensor of
 * depth_util.py:43 is never materialised.
 * ref_cam [B,2,4,4], src_cam [B,S,2,4,4] -> hmkt [B*S,12] (row-major Hm, then KT). */
int cnm_homography_terms_f32(const float* ref_cam, const float* src_cam, float* hmkt,
                             int B, int S, void* stream);

/* Inverse-depth sweep range for a given idepth_scale (depthNet_model.py:186-191).
 * Host-side helper, returns CNM_ERR_BAD_SCALE for anything but 2.0 / 3.0. */
int cnm_idepth_range_host(double idepth_scale, double* idepth_min, double* idepth_max);

/* ---------------------------------------------------------------- plane sweep (K1)
 * Replaces depthNet.getVolume (depthnet/depthNet_model.py:185-224):
 *   cost[p,d,y,x] = sum_c | bilinear_zero(src[p,c], u'-0.5, v'-0.5) - ref[b,c,y,x] |
 *   (u',v') = (t0,t1)/(t2+1e-6),  t = Hm (x,y,1)^T z_d + KT,
 *   z_d = 1/(idepth_min + d (idepth_max-idepth_min)/(D-1)),  p = b*S + s.
 * ref [B,3,H,W], src [B,S,3,H,W] (NCHW fp32), hmkt [B*S,12].
 * _nchw : volume [B*S,D,H,W]            -- the drop-in for getVolume's return value.
 * _c4   : x [B*S][G][H][W][4], G = D/4+1 -- groups 0..D/4-1 = cost planes, last group =
 *         (ref r,g,b,0): the concatenated conv1 input of depthNet_model.py:233 with the
 *         three image channels rotated to the end.  D must be a multiple of 4.
 * ws: 16-byte aligned scratch of cnm_planesweep_workspace_floats(B,S,H,W) floats: the tile queue of the
 *     persistent sweep (tickets drawn, workgroups gone).  The words must be ZERO when a call starts; the
 *     call leaves them zero, so a workspace is zeroed once, when it is allocated, and may then be reused by
 *     any number of calls that are ordered on one stream (two calls in flight at once need two workspaces).
 *     Nothing on the device checks this: with stale words tickets start offset and tiles are skipped, so a
 *     workspace whose last call did not run to completion (failed launch, abandoned capture) must be zeroed
 *     again.  The Python modules keep one workspace per net and order calls from different streams on it
 *     (depthnet/depthNet_model.py _EngineNet._workspace).
 *     ws == NULL is allowed: tiles are then dealt to the workgroups with a fixed stride (no scratch, slower
 *     when tiles differ in cost). */
size_t cnm_planesweep_workspace_floats(int B, int S, int H, int W);
/* sweep_store: cache policy of the plane sweep's output stores (0 plain, 2 non-temporal); the two write identical bytes.  Which one
 * is faster inside a step depends on the machine (the launch displaces the dirty lines its predecessors left in the memory-side
 * cache).  [r6] The policy is a per-device DECISION, never a side effect of launching: a launch reads one atomic, in this order --
 * forced by cnm_tune_sweep_store(0 / 2, .); CNM_SWEEP_STORE = plain | nt | 0 | 2 in the environment (read once); calibrated by
 * cnm_calibrate_sweep_store; the default, non-temporal.  No launch records events or takes a lock for it, launches under stream capture
 * use the same policy as all others.  The calibrated policy applies to launches whose output fits the 256 MB memory-side cache (what it
 * was measured on); a larger volume (640 x 480 x 96 planes x 16 pairs = 2 GB) is always written with non-temporal stores unless forced.
 * cnm_calibrate_sweep_store(scratch, scratch_floats, stream, median_us) is the measurement: 24 launches of the 16-pair 192 x 256 x 64
 * shape on the caller's scratch (cnm_calibrate_sweep_store_floats() floats, ~630 MB, contents irrelevant), alternating policies, each
 * behind a 400 MB fill and between two fence-free events on `stream`; BLOCKING (synchronises the stream), refused under stream capture
 * (CNM_ERR_BAD_ARG).  Returns the policy chosen for the current device (0 / 2) or a negative status; median_us (may be NULL) receives
 * [plain, nt] in microseconds.  The Python modules call it once per device and process when a depthNet first allocates its workspace.
 * cnm_tune_sweep_store(n, median_us): n = 0 / 2 forces; n = -1 drops the forced policy and the calibration; other n only query.
 * Returns the policy in force on the current device, -1 if nothing has decided yet (launches then use non-temporal stores). */
int cnm_tune_sweep_store(int n, float* median_us);
size_t cnm_calibrate_sweep_store_floats(void);
int cnm_calibrate_sweep_store(float* scratch, size_t scratch_floats, void* stream, float* median_us);
/* A caller that has measured the launch inside its own step (both policies forced in turn, the launch timed between its real neighbours
 * with cnm_debug_sweep_timing_arm -- cnmnet_amd/pipeline.py does this on its first call) records the decision for the current device; it
 * replaces the scratch calibration's.  policy 0 / 2; median_us (may be NULL) = [plain, nt] microseconds as measured. */
int cnm_decide_sweep_store(int policy, const float* median_us);
/* DEBUG / MEASUREMENT: cnm_debug_sweep_timing_arm(n) makes each of the next n plane-sweep launches of the process (any entry point,
 * including the one inside cnm_depthnet_forward_*; not those under stream capture) record a HIP event before and after itself on its
 * launch stream (events without the system-scope fence: the closing one does not wait for the launch's output to be written back);
 * n <= 0 disarms and frees.  cnm_debug_sweep_timing_read(ms, n) waits for the launches recorded so far and writes their elapsed
 * milliseconds; returns the count.  How bench.py times the launch between its real neighbours of a step.  [r6] Thread-safe: the launch
 * path reads one atomic while nothing is armed, everything else is under the hook's own mutex. */
int cnm_debug_sweep_timing_arm(int n);
int cnm_debug_sweep_timing_read(float* ms, int n);
int cnm_planesweep_volume_nchw_f32(const float* ref, const float* src, const float* hmkt, float* volume,
                                   float* ws, size_t ws_floats, int B, int S, int H, int W, int D,
                                   double idepth_min, double idepth_max, void* stream);
int cnm_planesweep_cat_c4_f32(const float* ref, const float* src, const float* hmkt, float* x,
                              float* ws, size_t ws_floats, int B, int S, int H, int W, int D,
                              double idepth_min, double idepth_max, void* stream);
/* [r6] ... and with the images / cameras as VIEWS of the caller's frame tensors: *_bstride = floats between consecutive frames (0 = dense:
 * 3 H W, S 3 H W, 32, 32 S).  ref = frames[:, 0] and src = frames[:, 1:] of frames [B][1 + S][3][H][W] (cams [B][1 + S][2][4][4] likewise) are read
 * where they lie; the reference slices its batches the same way (eval.py:440-447) and every consumer makes a contiguous copy.  The image strides
 * must be whole images (multiples of 3 H W, less than 65536 images; the sweep carries them as one packed word), the camera strides are free. */
int cnm_homography_terms_strided_f32(const float* ref_cam, long long ref_cam_bstride, const float* src_cam, long long src_cam_bstride, float* hmkt,
                                     int B, int S, void* stream);
int cnm_planesweep_cat_strided_c4_f32(const float* ref, long long ref_bstride, const float* src, long long src_bstride, const float* hmkt, float* x,
                                      float* ws, size_t ws_floats, int B, int S, int H, int W, int D, double idepth_min, double idepth_max, void* stream);
int cnm_planesweep_cat_strided_c8_f16(const float* ref, long long ref_bstride, const float* src, long long src_bstride, const float* hmkt, void* x,
                                      float* ws, size_t ws_floats, int B, int S, int H, int W, int D, double idepth_min, double idepth_max, void* stream);

/* ---------------------------------------------------------------- conv stack (K2-K5)
 * Weight packing.  Folds eval-mode BatchNorm (depthNet_model.py:19-79, eps/affine per
 * torch defaults) into the convolution and re-tiles OIHW weights for the MFMA kernel:
 *   w_packed [Kpad/16][Cout][16], k = (ky*ks+kx)*4*Gin + cpacked, cpacked = (ci+Cin-rot)%Cin
 *   b_packed [Cout] = beta - mean*gamma/sqrt(var+eps)   (or `bias`, or 0).
 * bn_* may be NULL (no BN); bias may be NULL.  Gin = ceil(Cin/4).
 * cnm_packed_conv_floats gives the size of w_packed. */
size_t cnm_packed_conv_floats(int Cout, int Cin, int ksize);
int cnm_pack_conv_bn_f32(const float* w_oihw, const float* bn_gamma, const float* bn_beta,
                         const float* bn_mean, const float* bn_var, const float* bias, float eps,
                         int Cout, int Cin, int ksize, int rot,
                         float* w_packed, float* b_packed, void* stream);

/* Conv2d(k, stride, pad=(k-1)/2, bias folded) + optional ReLU on c4 views, fp32 MFMA
 * implicit GEMM (v_mfma_f32_32x32x2_f32).  Cout % 64 == 0.  in: N x (Gin groups) x H x W. */
int cnm_conv2d_c4_f32(const float* in, int Gin_total, int gin0, int Gin,
                      float* out, int Gout_total, int gout0, int Cout,
                      const float* w_packed, const float* b_packed,
                      int N, int H, int W, int ksize, int stride, int relu, void* stream);

/* Same, reading torch.cat((a, b), 1) without materialising it: channel groups [0,Ga) come
 * from view a, [Ga,Ga+Gb) from view b (used where one skip tensor feeds two decoders,
 * depthNet_model.py:343,346 and :357,360). */
int cnm_conv2d_cat2_c4_f32(const float* in_a, int Ga_total, int ga0, int Ga,
                           const float* in_b, int Gb_total, int gb0, int Gb,
                           float* out, int Gout_total, int gout0, int Cout,
                           const float* w_packed, const float* b_packed,
                           int N, int H, int W, int ksize, int stride, int relu, void* stream);

/* Winograd F(2x2,3x3) twin of cnm_conv2d_cat2_c4_f32 for ksize 3, stride 1 (the reference's
 * nn.Conv2d(.., 3, 1, 1) layers: depthNet_model.py:77-86 conv_layer, :90-98 upconv, DepthRefineNet :248-285):
 * 2.25x fewer multiplies, fp32 data and accumulation, transforms in {0,+-1,+-1/2}.  u_packed comes from
 * cnm_pack_winograd_bn_f32 (G g G^T with the folded BatchNorm scale, MFMA operand order); b_packed is the
 * bias vector cnm_pack_conv_bn_f32 produces.  Gb = 0 -> single input view. */
size_t cnm_packed_winograd_floats(int Cout, int Cin);
int cnm_pack_winograd_bn_f32(const float* w_oihw, const float* bn_gamma, const float* bn_var, float eps,
                             int Cout, int Cin, int rot, float* u_packed, void* stream);
int cnm_conv3x3_winograd_c4_f32(const float* in_a, int Ga_total, int ga0, int Ga,
                                const float* in_b, int Gb_total, int gb0, int Gb,
                                float* out, int Gout_total, int gout0, int Cout,
                                const float* u_packed, const float* b_packed,
                                int N, int H, int W, int relu, void* stream);
/* 3x3 stride-2 (pad 1) twin on the same kernel and the same packed filter: keeps element (0,0) of every 2x2 tile
 * (out [N][Gout_total][ceil(H/2)][ceil(W/2)][4]).  The executors use it where the implicit-GEMM kernel would launch
 * fewer than 256 workgroups (depthNet conv5.3 at 192x256). */
int cnm_conv3x3_s2_winograd_c4_f32(const float* in, int Gin_total, int gin0, int Gin,
                                   float* out, int Gout_total, int gout0, int Cout,
                                   const float* u_packed, const float* b_packed,
                                   int N, int H, int W, int relu, void* stream);

/* Winograd F(4x4,3x3) variant of cnm_conv3x3_winograd_c4_f32: 36 instead of 64 multiplies per 16 outputs (1.78x fewer
 * than F(2x2,3x3)); fp32 data and accumulation, transform constants up to 8: per-layer error 1-3e-5 on O(1) outputs.
 * Workgroups cover 16 tiles of 4x4 outputs: meant for layers with >= 8192 tiles per cout block of 64.
 * u_packed from cnm_pack_winograd4_bn_f32. */
size_t cnm_packed_winograd4_floats(int Cout, int Cin);
int cnm_pack_winograd4_bn_f32(const float* w_oihw, const float* bn_gamma, const float* bn_var, float eps,
                              int Cout, int Cin, int rot, float* u_packed, void* stream);
/* Packed F(4x4,3x3) / F(2x2,5x5) filter of the DATA GRADIENT of a stride-1 convolution with weight w [Cw_out][Cw_in][k][k]
 * (training, train.py:307-310 through autograd): w'[ci][co] = w[co][ci] rotated by 180 degrees, i.e. Cw_in output channels
 * (a multiple of 64) and Cw_out input channels, cnm_packed_winograd4_floats(Cw_in, Cw_out) floats; read straight from w. */
int cnm_pack_winograd4_dgrad_f32(const float* w_oihw, int Cw_out, int Cw_in, int ksize, float* u_packed, void* stream);
/* [r6] Many 3x3 filters in ONE launch (training re-packs every filter every step, forward and data-gradient form): jobs_dev = device
 * array of njobs records { const float* w; float* out; int Cout, Cin, rot, nchunks, dgrad, first_block; } (40 bytes each; Cout / Cin as
 * the single-filter calls see them: the data-gradient form of w [Cw_out][Cw_in][3][3] has Cout = Cw_in, Cin = Cw_out, dgrad = 1, rot = 0;
 * nchunks = ceil(4 ceil(Cin / 4) / 16); a job owns nchunks * Cout / 16 blocks from first_block on, jobs sorted by first_block);
 * total_blocks = their sum.  No BatchNorm fold.  Bit-identical to the single-filter calls. */
int cnm_pack_winograd4_batch_f32(const void* jobs_dev, int njobs, int total_blocks, void* stream);
int cnm_conv3x3_winograd4_c4_f32(const float* in_a, int Ga_total, int ga0, int Ga,
                                 const float* in_b, int Gb_total, int gb0, int Gb,
                                 float* out, int Gout_total, int gout0, int Cout,
                                 const float* u_packed, const float* b_packed,
                                 int N, int H, int W, int relu, void* stream);
/* The same convolution with a sync workspace for the LDS-staged persistent kernel (conv_winograd4s.hip: 128 output
 * channels x 16 tiles per workgroup, one workgroup per CU).  sync_ws = cnm_wino36_sync_floats() floats: 4096 bytes of flag
 * words -- zero before the first use and re-armed by every completed hand-off; a failed one (cnm_engine_status) leaves nothing
 * a later call could mistake for its own -- followed by one 128 KB partial-output slot per CU; it must not be shared by
 * launches that can run concurrently.  With it the kernel splits the layer's (unit, 16-channel chunk)
 * phases into equal contiguous ranges, one per CU, whatever the unit count: a unit cut by a range boundary is finished by
 * the range that holds its first chunk, which adds the other ranges' partial outputs in range order (write-through
 * stores + flag, agent-scope acquire).  Results are bit-reproducible from run to run; they differ from the unsplit
 * evaluation in the last bits (the fp32 accumulation chain of a cut unit is summed in two or more pieces).  Shapes the
 * staged kernel does not take (Cout not a multiple of 128, fewer than six tile columns) run as
 * cnm_conv3x3_winograd4_c4_f32.  sync_ws = NULL is allowed and means exactly that function. */
size_t cnm_wino36_sync_floats(void);
int cnm_conv3x3_winograd4_sync_c4_f32(const float* in_a, int Ga_total, int ga0, int Ga,
                                      const float* in_b, int Gb_total, int gb0, int Gb,
                                      float* out, int Gout_total, int gout0, int Cout,
                                      const float* u_packed, const float* b_packed,
                                      int N, int H, int W, int relu, float* sync_ws, size_t sync_floats, void* stream);

/* [r6] The four-wave F(4x4,3x3) kernel (csrc/conv_winograd4q.hip): 64 output channels x 32 tiles per workgroup, ONE wave per SIMD with
 * 288 accumulator registers, so that every weight fragment streamed out of L2 feeds two groups of MFMAs (half the weight stream per
 * flop of the eight-wave kernel above) and Cout only has to be a multiple of 64 (the three iconv1 layers, depthNet_model.py:110-112,
 * 278-308).  The filter is the 36-point packed filter of cnm_pack_winograd4_bn_f32 re-ordered for 8-channel phases -- a permutation,
 * cnm_packed_winograd4_quad_floats(Cout, Cin) floats.  Shapes: cnm_conv3x3_winograd4q_ok (at least 12 x 2 or 6 x 3 tiles per image);
 * others return CNM_ERR_BAD_ARG.  Same sync-workspace contract as cnm_conv3x3_winograd4_sync_c4_f32 (NULL / 0 allowed: ranges then end
 * on unit boundaries).  Results are bit-reproducible run to run and differ from the eight-wave kernel's in the last bits (the reduction
 * over input channels is grouped differently).
 * cnm_tune_wino36_quad: 0 = the fp32 executors never use it, 1 (default) = on the layers where it was measured faster, 2 = wherever
 * the shape is eligible and the quad filter was supplied (cnm_layer_weights.u4q); other values query.  Returns the previous value. */
size_t cnm_packed_winograd4_quad_floats(int Cout, int Cin);
int cnm_repack_winograd4_quad_f32(const float* u_packed, int Cout, int Cin, float* u_packed_quad, void* stream);
int cnm_conv3x3_winograd4q_ok(int Cout, int H, int W);
int cnm_conv3x3_winograd4q_sync_c4_f32(const float* in_a, int Ga_total, int ga0, int Ga,
                                       const float* in_b, int Gb_total, int gb0, int Gb,
                                       float* out, int Gout_total, int gout0, int Cout,
                                       const float* u_packed_quad, const float* b_packed,
                                       int N, int H, int W, int relu, float* sync_ws, size_t sync_floats, void* stream);
int cnm_tune_wino36_quad(int mode);

/* Winograd F(2x2,5x5) for the 5x5 stride-1 layer (conv2.0 = nn.Conv2d(128, 256, 5, 1, 2), depthNet_model.py:141-144):
 * the 36-point machine of the F(4x4,3x3) kernel with 2x2 output tiles; 9 instead of 25 multiplies per output (the
 * row-wise kernel: 15).  u_packed from cnm_pack_winograd5x5_bn_f32, size cnm_packed_winograd4_floats(Cout, Cin). */
int cnm_pack_winograd5x5_bn_f32(const float* w_oihw, const float* bn_gamma, const float* bn_var, float eps,
                                int Cout, int Cin, int rot, float* u_packed, void* stream);
int cnm_conv5x5_winograd_c4_f32(const float* in_a, int Ga_total, int ga0, int Ga,
                                const float* in_b, int Gb_total, int gb0, int Gb,
                                float* out, int Gout_total, int gout0, int Cout,
                                const float* u_packed, const float* b_packed,
                                int N, int H, int W, int relu, void* stream);
/* ... with the sync workspace of cnm_conv3x3_winograd4_sync_c4_f32 (same contract): the LDS-staged persistent kernel with
 * 2 x 2 output tiles. */
int cnm_conv5x5_winograd_sync_c4_f32(const float* in_a, int Ga_total, int ga0, int Ga,
                                     const float* in_b, int Gb_total, int gb0, int Gb,
                                     float* out, int Gout_total, int gout0, int Cout,
                                     const float* u_packed, const float* b_packed,
                                     int N, int H, int W, int relu, float* sync_ws, size_t sync_floats, void* stream);

/* Phase-scatter form of four 3x3 stride-1 convolutions: out[2y + a][2x + b] = conv3x3(in, w_phase[2a + b])[y][x] (zero
 * padding), in [N][G][H][W][4] -> out [N][Gout][2H][2W][4], ONE launch on the kernel of the fused up_conv layers with the
 * four filters packed as 4*Cout output channels, phase major (cnm_pack_winograd4_bn_f32 of the [4*Cout, Cin, 3, 3]
 * tensor; b_packed: 4*Cout values or NULL).  It is the data gradient of a stride-2 convolution with a 3x3 or 5x5 filter
 * (reference train.py:164-310 via autograd of depthNet_model.py:19-43 conv_layer): the dX phases are stride-1
 * convolutions of dY.  Sync workspace as for cnm_conv3x3_winograd4_sync_c4_f32 (NULL / 0 allowed). */
int cnm_conv3x3_phase_scatter_winograd4_sync_c4_f32(const float* in, int Gin_total, int gin0, int Gin,
                                                    float* out, int Gout_total, int gout0, int Cout,
                                                    const float* u_packed, const float* b_packed,
                                                    int N, int H, int W, int relu, float* sync_ws, size_t sync_floats, void* stream);

/* ... with 4x4 phase filters (taps at offsets -1 .. 2) on F(3x3,4x4): the data gradient of a stride-2 7x7 convolution (conv1.3).
 * u_packed = cnm_pack_winograd36_f32(ksize 4) of the [4*Cout, Cin, 4, 4] tensor.  Staged kernel only (Cout % 32 == 0, at least
 * 6 x 2 tiles of 3 x 3 pixels per image): CNM_ERR_BAD_ARG otherwise. */
int cnm_conv4x4_phase_scatter_winograd_sync_c4_f32(const float* in, int Gin_total, int gin0, int Gin,
                                                   float* out, int Gout_total, int gout0, int Cout,
                                                   const float* u_packed, const float* b_packed,
                                                   int N, int H, int W, int relu, float* sync_ws, size_t sync_floats, void* stream);
/* 36-point pack of a plain [Cout, Cin, k, k] filter, k = 3 / 4 / 5 (F(4x4,3x3) / F(3x3,4x4) / F(2x2,5x5)); no BatchNorm fold. */
int cnm_pack_winograd36_f32(const float* w_oihw, int Cout, int Cin, int ksize, float* u_packed, void* stream);

/* The STRIDE-2 5x5 / 7x7 (and, where they would run the implicit GEMM, 3x3) layers (conv2.3 = nn.Conv2d(256, 256, 5, 2, 2), conv1.3 = nn.Conv2d(128, 128, 7, 2, 3),
 * depthNet_model.py:136-139,145-148 through conv_layer :19-43) as a stride-1 convolution of the four pixel phases of the
 * input (space to depth, never materialised) on the LDS-staged 36-point kernel: 5x5 -> four 3x3 phase filters,
 * F(4x4,3x3), 9 multiplies per output instead of 25 (row-wise phase kernel: 15); 3x3 -> four (at most) 2x2 phase filters in
 * 3x3 slots, F(4x4,3x3): the direct count of 9, at the staged kernel's efficiency; 7x7 -> four 4x4 phase filters,
 * F(3x3,4x4) on the same six points, 16 instead of 49 (22.75).  H, W (even) = INPUT size, out = [N][Gout][H/2][W/2][4].
 * u_packed from cnm_pack_winograd4_s2_bn_f32 (cnm_packed_winograd4_s2_floats floats).  Staged kernel only: needs the
 * sync workspace of cnm_conv3x3_winograd4_sync_c4_f32 (same contract) and a shape cnm_conv_s2_winograd4_ok() accepts
 * (Cout % 128 == 0, even H and W, at least 3 x 3 tiles of 4 x 4 (5x5) / 3 x 3 (7x7) outputs per image): CNM_ERR_BAD_ARG otherwise.
 * cnm_tune_wino4_s2(0) makes the fp32 executors keep the row-wise phase kernel for these layers (default 1; other values query). */
size_t cnm_packed_winograd4_s2_floats(int Cout, int Cin);
int cnm_pack_winograd4_s2_bn_f32(const float* w_oihw, const float* bn_gamma, const float* bn_var, float eps,
                                 int Cout, int Cin, int ksize, int rot, float* u_packed, void* stream);
int cnm_conv_s2_winograd4_ok(int Cout, int H, int W, int ksize);
int cnm_conv_s2_winograd4_sync_c4_f32(const float* in_a, int Ga_total, int ga0, int Ga,
                                      const float* in_b, int Gb_total, int gb0, int Gb,
                                      float* out, int Gout_total, int gout0, int Cout,
                                      const float* u_packed, const float* b_packed,
                                      int N, int H, int W, int ksize, int relu, float* sync_ws, size_t sync_floats, void* stream);
int cnm_tune_wino4_s2(int on);

/* nn.Upsample(scale_factor=2, mode='bilinear') followed by Conv2d(3x3, pad 1) + folded BatchNorm + ReLU -- the
 * reference's up_conv_layer (depthNet_model.py:89-112) -- as ONE pass over the LOW-resolution input: upsample-then-3x3
 * equals, per output row / column parity, a 3x3 filter on the low-resolution image; the four composed filters are packed
 * (cnm_pack_winograd4_bn_f32 with 4*Cout output channels, phase major; bias replicated four times) and the F(4x4,3x3)
 * kernel writes its 4*Cout virtual channels pixel-shuffled.  in [N][Gin_total][H][W][4] -> out [N][Gout_total][2H][2W][4].
 * The composition corresponds to REPLICATE padding of the upsampled image; the reference zero-pads, which differs on the
 * one-pixel output ring only: with_ring = 1 leaves the ring pre-activation and cnm_conv3x3_upsampled_ring_c4_f32
 * (w_ring from cnm_pack_upsampled_ring_f32) finishes it; with_ring = 0 returns the replicate-padding result. */
int cnm_conv3x3_upsampled_winograd4_c4_f32(const float* in, int Gin_total, int gin0, int Gin,
                                           float* out, int Gout_total, int gout0, int Cout,
                                           const float* u_packed, const float* b_packed,
                                           int N, int H, int W, int relu, int with_ring, void* stream);
/* ... with the sync workspace of cnm_conv3x3_winograd4_sync_c4_f32 (same contract). */
int cnm_conv3x3_upsampled_winograd4_sync_c4_f32(const float* in, int Gin_total, int gin0, int Gin,
                                                float* out, int Gout_total, int gout0, int Cout,
                                                const float* u_packed, const float* b_packed,
                                                int N, int H, int W, int relu, int with_ring,
                                                float* sync_ws, size_t sync_floats, void* stream);

/* Ring pass of the fused upsample + 3x3 convolution: subtracts, on the one-pixel output ring, the filter taps that the
 * replicate-padded composition added beyond the reference's zero padding, then applies bias and ReLU there.
 * w_ring = the plain 3x3 filter with the folded BatchNorm scale in MFMA operand order
 * (cnm_packed_upsampled_ring_floats floats); b_packed = the layer's folded bias (Cout floats). */
size_t cnm_packed_upsampled_ring_floats(int Cout, int Cin);
int cnm_pack_upsampled_ring_f32(const float* w_oihw, const float* bn_gamma, const float* bn_var, float eps,
                                int Cout, int Cin, float* w_ring, void* stream);
int cnm_conv3x3_upsampled_ring_c4_f32(const float* in, int Gin_total, int gin0, int Gin,
                                      float* out, int Gout_total, int gout0, int Cout,
                                      const float* w_ring, const float* b_packed,
                                      int N, int H, int W, int relu, void* stream);

/* fp16 twins (c8 tensors, BASELINE config 5): the composed phase filters packed with cnm_pack_conv_bn_f16 as 4*Cout output
 * channels (phase major; bias replicated four times) run on the LDS-DMA implicit-GEMM kernel with clamped window samples
 * and the pixel-shuffling epilogue; the ring pass reads / rewrites half data and keeps its arithmetic, w_ring and b_packed
 * in fp32.  Gin = 16-byte channel groups (8 channels each). */
int cnm_conv3x3_upsampled_c8_f16(const void* in, int Gin_total, int gin0, int Gin,
                                 void* out, int Gout_total, int gout0, int Cout,
                                 const void* w_packed_f16, const float* b_packed,
                                 int N, int H, int W, int relu, int with_ring, void* stream);
int cnm_conv3x3_upsampled_ring_c8_f16(const void* in, int Gin_total, int gin0, int Gin,
                                      void* out, int Gout_total, int gout0, int Cout,
                                      const float* w_ring, const float* b_packed,
                                      int N, int H, int W, int relu, void* stream);

/* Row-wise Winograd twin of cnm_conv2d_cat2_c4_f32 for ksize R = 5 or 7, stride 1 or 2 (the reference's
 * conv1 = nn.Conv2d(3+D, 128, 7, 1, 3) / (128, 128, 7, 2, 3) and conv2 = nn.Conv2d(128, 256, 5, 1, 2) /
 * (256, 256, 5, 2, 2), depthNet_model.py:137-148 via conv_layer :77-86): the transform runs along image rows, the R
 * kernel rows stay in the GEMM reduction.  Stride 1: F(2,R), (R+1)/2 instead of R multiplies per output and kernel
 * row; stride 2: the two column phases of the input are F(2,ceil(R/2)) correlations accumulated together,
 * ceil(R/2)+1 instead of R multiplies.  tile = outputs per tile along the row: 2, or 4 (not for ksize 5 stride 1):
 * F(4,7) -- 10 multiplies per 4 outputs and kernel row, 1.6x fewer than F(2,7), measured error 3e-4 on O(1) outputs
 * against 5e-5 for tile 2 -- and F(4,4) / F(4,3) column phases for stride 2 (1.4x / 1.33x fewer than tile 2).  The
 * inference executors use tile 4, the training path keeps tile 2.
 * u_packed from cnm_pack_winograd_rows_bn_f32 (same ksize, stride and tile).
 * Also ksize 3 with stride 2 and tile 4: two F(4,2) column phases (5 multiplies per 4 outputs, phase and kernel row
 * instead of 6); in cnm_layer_weights such a filter goes into u4 of a 3x3 stride-2 layer (u holds the F(2x2,3x3) filter). */
size_t cnm_packed_winograd_rows_floats(int Cout, int Cin, int ksize, int stride, int tile);
int cnm_pack_winograd_rows_bn_f32(const float* w_oihw, const float* bn_gamma, const float* bn_var, float eps,
                                  int Cout, int Cin, int ksize, int stride, int tile, int rot, float* u_packed, void* stream);
int cnm_conv_rows_winograd_c4_f32(const float* in_a, int Ga_total, int ga0, int Ga,
                                  const float* in_b, int Gb_total, int gb0, int Gb,
                                  float* out, int Gout_total, int gout0, int Cout,
                                  const float* u_packed, const float* b_packed,
                                  int N, int H, int W, int ksize, int stride, int tile, int relu, void* stream);
/* The same with a sync workspace (cnm_wino36_sync_floats() floats, zero before the first use, one per
 * stream): the 7x7 stride-1 four-output shape with Cout % 128 == 0 runs the LDS-staged persistent kernel
 * (csrc/conv_rows_staged.hip) with equal shares of the reduction per CU; other shapes ignore the workspace.
 * cnm_tune_rows7_staged(0) routes that shape back to the gather-fed kernel (returns the previous setting). */
int cnm_conv_rows_winograd_sync_c4_f32(const float* in_a, int Ga_total, int ga0, int Ga,
                                       const float* in_b, int Gb_total, int gb0, int Gb,
                                       float* out, int Gout_total, int gout0, int Cout,
                                       const float* u_packed, const float* b_packed,
                                       int N, int H, int W, int ksize, int stride, int tile, int relu,
                                       float* sync_ws, size_t sync_floats, void* stream);
int cnm_tune_rows7_staged(int on);

/* nn.Upsample(scale_factor=2, mode='bilinear') with align_corners=False
 * (depthNet_model.py:94,105) on a c4 view: [N,G,H,W,4] -> [N,G,2H,2W,4]. */
int cnm_upsample2x_c4_f32(const float* in, int Gin_total, int gin0,
                          float* out, int Gout_total, int gout0,
                          int N, int G, int H, int W, void* stream);

/* depth_layer + scale (depthNet_model.py:82-84,246,251,256,261,351,365):
 *   disp[n,y,x] = scale * sigmoid(conv3x3(in)[n,y,x] + bias);  w_head [9][C] (tap-major).
 * If up_out != NULL also writes F.upsample(disp, 2) (nearest, :247,252,257) as the c4
 * group `up_g` of the [N,up_Gtotal,2H,2W,4] buffer: (value,0,0,0). */
int cnm_pack_head_f32(const float* w_oihw, int C, float* w_head, void* stream);
int cnm_head_sigmoid_c4_f32(const float* in, int Gin_total, int gin0, int C,
                            const float* w_head, const float* bias, float scale,
                            float* disp, float* up_out, int up_Gtotal, int up_g,
                            int N, int H, int W, void* stream);

/* DepthRefineNet input assembly (depthNet_model.py:332-333), channels rotated so the
 * 64 feature channels come first: x = [iconv01+iconv02 (C), id1, id2, |id1-id2|, 0].
 * idepth0x: image n starts at idepth0x + n*idepth_stride floats (H*W when contiguous;
 * 2*H*W when the two sides are interleaved pairs of one depthnet call). */
int cnm_refine_assemble_c4_f32(const float* idepth01, const float* idepth02, long long idepth_stride,
                               const float* f1, int G1_total, int g1,
                               const float* f2, int G2_total, int g2,
                               float* x, int N, int C, int H, int W, void* stream);

/* Multi-source variant (eval.py:656-663 for S=4, :917-929 for S=6): idepth_pairs [B*S,H,W] and
 * feat_pairs_c4 [B*S,C/4,H,W,4] are one depthnet call's outputs, pair p = b*S + s; even sources are
 * averaged into side 1, odd into side 2 ((a+c)*0.5, (a+c+e)/3.), then assembled as above. S even. */
int cnm_refine_assemble_multi_c4_f32(const float* idepth_pairs, const float* feat_pairs_c4, float* x,
                                     int B, int S, int C, int H, int W, void* stream);

/* Layout converters at the module boundary (NCHW torch tensors <-> c4 views). */
int cnm_nchw_to_c4_f32(const float* nchw, float* c4, int G_total, int g0, int N, int C, int H, int W, void* stream);
int cnm_c4_to_nchw_f32(const float* c4, int G_total, int g0, float* nchw, int N, int C, int H, int W, void* stream);

/* ---------------------------------------------------------------- whole networks
 * Layer tables: the engine's own description of the two nets (names follow the
 * reference's state_dict prefixes, e.g. "conv1.0"/"conv1.1" = conv / its BN). */
typedef struct cnm_layer_info {
    const char* conv_key;   /* state_dict prefix of the Conv2d, e.g. "upconv5.1" */
    const char* bn_key;     /* state_dict prefix of its BatchNorm2d, NULL for heads */
    int Cin, Cout, ksize, stride;
    int rot;                /* input-channel rotation used when packing (0 or 3) */
    int is_head;            /* 1: depth_layer (Cout=1, bias, sigmoid) */
} cnm_layer_info;

#define CNM_NET_DEPTH 0
#define CNM_NET_REFINE 1
int cnm_net_num_layers(int net);
int cnm_net_layer(int net, int index, cnm_layer_info* info);   /* D=64 table */

/* w, b: cnm_pack_conv_bn_* (or cnm_pack_head_f32) outputs.  u: optional Winograd-domain filter of
 * cnm_pack_winograd_bn_f32 (3x3) / cnm_pack_winograd_rows_bn_f32 (5x5, 7x7) -- when non-NULL the fp32 executors
 * run that layer (3x3 stride 1; 5x5 / 7x7 stride 1 or 2) through cnm_conv3x3_winograd_c4_f32 /
 * cnm_conv_rows_winograd_c4_f32 (w may then be NULL); ignored by heads, 3x3 stride-2 layers and the fp16 engine.
 * u4: optional 36-point filter -- cnm_pack_winograd4_bn_f32 (F(4x4,3x3)) for 3x3 stride-1 layers,
 * cnm_pack_winograd5x5_bn_f32 (F(2x2,5x5)) for 5x5 stride-1 layers: used instead of u when the layer has enough tiles
 * to fill the chip (>= CNM_WINO4_MIN_WORKGROUPS workgroups of 64 couts x 16 tiles); cnm_pack_winograd4_s2_bn_f32 for
 * 5x5 / 7x7 stride-2 layers: used instead of u where cnm_conv_s2_winograd4_ok() accepts the shape.
 * uu alone (bu = wr = NULL) on a 3x3 stride-2 layer: optional cnm_pack_winograd4_s2_bn_f32(ksize 3) filter -- the layer then runs
 * on the four pixel phases of its input (staged 36-point kernel) where neither the F(2x2) nor the row kernel is chosen.
 * uu, bu, wr: optional, up_conv layers only -- the four composed upsample-then-3x3 phase filters packed as 4*Cout
 * output channels (fp32 engine: cnm_pack_winograd4_bn_f32; fp16 engine: cnm_pack_conv_bn_f16, half data), the folded
 * bias four times, and the ring-pass filter (cnm_pack_upsampled_ring_f32, fp32 for both engines); all three or none. */
typedef struct cnm_layer_weights { const float* w; const float* b; const float* u; const float* u4;
                                   const float* uu; const float* bu; const float* wr;
                                   const float* u4q;   /* [r6, ABI 6] 3x3 stride-1 layers: the quad re-ordering of u4 (cnm_repack_winograd4_quad_f32), or NULL */
                                 } cnm_layer_weights;

/* depthNet.forward (depthNet_model.py:226-263) for P = B*S (ref,src) pairs.
 * weights[i] = packed tensors of layer i of the CNM_NET_DEPTH table (conv1.0 packed with
 * Cin = 3+D).  Outputs: disp1..4 [P,1,H/2^i,W/2^i], iconv1 as c4 [P,16,H,W,4].
 * ws: workspace of cnm_depthnet_workspace_floats(P,H,W,D) floats. */
size_t cnm_depthnet_workspace_floats(int P, int H, int W, int D);
int cnm_depthnet_forward_f32(const cnm_layer_weights* weights, float idepth_scale, int D,
                             const float* ref, const float* src, const float* ref_cam, const float* src_cam,
                             float* disp1, float* disp2, float* disp3, float* disp4, float* iconv1_c4,
                             float* ws, size_t ws_floats, int B, int S, int H, int W, void* stream);

/* DepthRefineNet.forward (depthNet_model.py:331-370).  iconv01/02 are c4 views
 * (base, G_total, g0) with 64 channels.  Outputs disp_refined, prob_map [N,1,H,W];
 * iconv1_depth_c4 [N,16,H,W,4] may be NULL (ReturnVolume=False). */
size_t cnm_refinenet_workspace_floats(int N, int H, int W);
int cnm_refinenet_forward_f32(const cnm_layer_weights* weights, float idepth_scale,
                              const float* idepth01, const float* idepth02, long long idepth_stride,
                              const float* iconv01, int G1_total, int g1,
                              const float* iconv02, int G2_total, int g2,
                              float* disp_refined, float* prob_map, float* iconv1_depth_c4,
                              float* ws, size_t ws_floats, int N, int H, int W, void* stream);

/* Same for a frame with S (even) sources processed by ONE depthnet call (eval.py:635-663, :885-929):
 * idepth_pairs = disp1 [B*S,1,H,W], iconv_pairs_c4 = iconv1 [B*S,16,H,W,4]; workspace as for N = B. */
int cnm_refinenet_forward_multi_f32(const cnm_layer_weights* weights, float idepth_scale,
                                    const float* idepth_pairs, const float* iconv_pairs_c4, int S,
                                    float* disp_refined, float* prob_map, float* iconv1_depth_c4,
                                    float* ws, size_t ws_floats, int B, int H, int W, void* stream);

/* ---------------------------------------------------------------- fp16 path (BASELINE config 5)
 * Same operators on fp16 storage: activations "c8" = [N][G][H][W][8 halfs], channel c = 8g + j, 16 bytes per
 * (pixel, group) exactly like c4, so views are again (base, G_total, g0) with G counted in 8-channel groups.
 * Convolutions run on v_mfma_f32_32x32x16_f16 with fp32 accumulation, fp32 bias/ReLU, fp16 store; the plane
 * sweep computes in fp32 and stores fp16; heads and normals stay fp32.  Tolerance vs the fp32 path is stated in
 * tests/test_gpu_fp16.py.  Weights: [k-step of 64 halfs][8 slots][round64(Cout)][8 halfs]; [r5] the k-steps run filter row by
 * filter row, 8-group block by block, tap by tap within the row -- slot w of step (ky, gb, kx) = channel group 8 gb + w at tap
 * (ky, kx) -- and the G % 8 left-over groups follow with their taps as slots (cnmnet_amd/csrc/conv_mfma.hip: F16Walk). */
size_t cnm_packed_conv_halfs(int Cout, int Cin, int ksize);
int cnm_pack_conv_bn_f16(const float* w_oihw, const float* bn_gamma, const float* bn_beta,
                         const float* bn_mean, const float* bn_var, const float* bias, float eps,
                         int Cout, int Cin, int ksize, int rot, void* w_packed_f16, float* b_packed, void* stream);
int cnm_conv2d_c8_f16(const void* in, int Gin_total, int gin0, int Gin,
                      void* out, int Gout_total, int gout0, int Cout,
                      const void* w_packed_f16, const float* b_packed,
                      int N, int H, int W, int ksize, int stride, int relu, void* stream);
int cnm_conv2d_cat2_c8_f16(const void* in_a, int Ga_total, int ga0, int Ga,
                           const void* in_b, int Gb_total, int gb0, int Gb,
                           void* out, int Gout_total, int gout0, int Cout,
                           const void* w_packed_f16, const float* b_packed,
                           int N, int H, int W, int ksize, int stride, int relu, void* stream);
int cnm_planesweep_cat_c8_f16(const float* ref, const float* src, const float* hmkt, void* x,
                              float* ws, size_t ws_floats, int B, int S, int H, int W, int D,
                              double idepth_min, double idepth_max, void* stream);
int cnm_upsample2x_c8_f16(const void* in, int Gin_total, int gin0, void* out, int Gout_total, int gout0,
                          int N, int G, int H, int W, void* stream);
int cnm_head_sigmoid_c8_f16(const void* in, int Gin_total, int gin0, int C,
                            const float* w_head, const float* bias, float scale,
                            float* disp, void* up_out, int up_Gtotal, int up_g,
                            int N, int H, int W, void* stream);
int cnm_refine_assemble_multi_c8_f16(const float* idepth_pairs, const void* feat_pairs_c8, void* x,
                                     int B, int S, int C, int H, int W, void* stream);
int cnm_nchw_to_c8_f16(const float* nchw, void* c8, int G_total, int g0, int N, int C, int H, int W, void* stream);
int cnm_c8_to_nchw_f16(const void* c8, int G_total, int g0, float* nchw, int N, int C, int H, int W, void* stream);
size_t cnm_depthnet_workspace_floats_f16(int P, int H, int W, int D);
int cnm_depthnet_forward_f16(const cnm_layer_weights* weights, float idepth_scale, int D,
                             const float* ref, const float* src, const float* ref_cam, const float* src_cam,
                             float* disp1, float* disp2, float* disp3, float* disp4, void* iconv1_c8,
                             float* ws, size_t ws_floats, int B, int S, int H, int W, void* stream);
/* [r6] depthNet forwards with the images and cameras as views of the caller's frame tensors (see cnm_planesweep_cat_strided_c4_f32): *_bstride =
 * floats between consecutive frames, 0 = dense. */
int cnm_depthnet_forward_strided_f32(const cnm_layer_weights* weights, float idepth_scale, int D,
                                     const float* ref, long long ref_bstride, const float* src, long long src_bstride,
                                     const float* ref_cam, long long ref_cam_bstride, const float* src_cam, long long src_cam_bstride,
                                     float* disp1, float* disp2, float* disp3, float* disp4, float* iconv1_c4,
                                     float* ws, size_t ws_floats, int B, int S, int H, int W, void* stream);
int cnm_depthnet_forward_strided_f16(const cnm_layer_weights* weights, float idepth_scale, int D,
                                     const float* ref, long long ref_bstride, const float* src, long long src_bstride,
                                     const float* ref_cam, long long ref_cam_bstride, const float* src_cam, long long src_cam_bstride,
                                     float* disp1, float* disp2, float* disp3, float* disp4, void* iconv1_c8,
                                     float* ws, size_t ws_floats, int B, int S, int H, int W, void* stream);
int cnm_refinenet_forward_multi_f16(const cnm_layer_weights* weights, float idepth_scale,
                                    const float* idepth_pairs, const void* iconv_pairs_c8, int S,
                                    float* disp_refined, float* prob_map, void* iconv1_depth_c8,
                                    float* ws, size_t ws_floats, int B, int H, int W, void* stream);

/* ---------------------------------------------------------------- training path
 * The reference trains with stock autograd (train.py:164-310).  These are the backward operators of
 * the conv stack on c4 activations; the Python side wires them into torch.autograd.Functions
 * (cnmnet_amd/autograd.py).  The plane sweep has no backward.
 *
 * Data gradient: dX = conv_transpose(dY, W) as the same MFMA implicit GEMM with flipped taps and
 * swapped channel roles; stride 2 gathers the zero-upsampled dY without materialising it.
 * dy [N,ceil(Cout/4),Ho,Wo,4] -> dx [N,ceil(Cin/4),H,W,4] in the forward input's (rotated) channel order. */
size_t cnm_packed_dgrad_floats(int Cout, int Cin, int ksize);
int cnm_pack_conv_dgrad_f32(const float* w_oihw, int Cout, int Cin, int ksize, int rot, float* w_packed, void* stream);
int cnm_conv2d_dgrad_c4_f32(const float* dy, int Gy_total, int gy0, int Cout,
                            float* dx, int Gx_total, int gx0, int Cin,
                            const float* w_packed_dgrad, int N, int H, int W, int ksize, int stride, void* stream);

/* Weight gradient: dW[co,ci,ky,kx] = sum_{n,y,x} dY[n,co,y,x] X[n,ci,y*s+ky-p,x*s+kx-p], an MFMA GEMM with
 * the pixels as the reduction dimension, split into partial sums (ws) and reduced in fp64.
 * x [N,.,H,W,4] (forward input view), dy [N,.,Ho,Wo,4] -> dw_oihw [Cout,Cin,k,k] (rotation undone). */
size_t cnm_conv2d_wgrad_workspace_floats(int Cout, int Cin, int ksize, int N, int Ho, int Wo);
/* [r6] How the pixel reduction of every weight-gradient GEMM below is shared out.  1 (default): stream-K -- ONE persistent launch of three
 *   workgroups per CU that share the flattened (tile, 16-pixel step) space in equal contiguous ranges; a tile's sum is cut only where a range
 *   boundary falls into it and the (at most one per workgroup) partial tiles are handed over inside the launch, in a fixed order (results
 *   are bit-reproducible on a device; a hand-off that times out is reported through cnm_engine_status, as for the convolution kernels).
 *   0: the earlier split form -- a grid of ~2048 workgroups, every tile's `splits` partial copies written to ws and summed in fp64 by a
 *   second kernel (134 MB written and read back per layer).  The workspace sizes cover both.  Returns the previous value. */
int cnm_tune_wgrad_streamk(int n);
/* ... and which launches take that form: those whose tiles are shared by at most n ranges on average (default 8; 0 = every launch).  The
 *   range that finishes a tile adds the other ranges' partial tiles one round trip after the other, so a launch of a few dozen tiles cut
 *   into 768 ranges would end on 10-20 serial round trips; such launches keep the split form.  Returns the previous value. */
int cnm_tune_wgrad_streamk_share(int n);
/* wgrad_linear [r6]: 1 (default) = the Winograd-domain GEMMs (1 x 1 taps over one row of tiles per frequency point) load their operands at a
 *   per-thread constant offset plus a scalar step offset -- no vector ALU in the loop -- and every other launch whose output width is a multiple of 16
 *   walks in row steps (scalar image / row / column, per-thread constants); 0 = the general coordinate walk everywhere (A/B).  Bit-identical results.
 *   Returns the previous value. */
int cnm_tune_wgrad_linear(int n);
/* The same gradient for a 3x3 stride-1 pad-1 convolution in the Winograd domain of the forward's F(4x4,3x3):
 * dW = G^T [ sum_tiles (A dY A^T) (.) (B^T X B) ] G -- two transform kernels, ONE launch of 36 GEMMs over the tiles (a quarter of
 * the direct gradient's multiplies), a finishing kernel (fp64 split sums, G^T . G, OIHW scatter).  Same arguments as
 * cnm_conv2d_wgrad_c4_f32 without ksize / stride; ws of cnm_conv3x3_wgrad_winograd_workspace_floats floats (transformed X and
 * dY + partial tiles). */
size_t cnm_conv3x3_wgrad_winograd_workspace_floats(int Cout, int Cin, int N, int H, int W);
/* ... and of the stride-2 5x5 / 7x7 layers on the four pixel phases of x (3x3 / 4x4 phase filters, F(4x4,3x3) / F(3x3,4x4), as
 * cnm_conv_s2_winograd4_sync_c4_f32 runs the forward): dY transformed once, X per phase as 4*Cin channels.  H, W (even) = INPUT size. */
/* ... and of the 7x7 stride-1 layer (conv1.0) row-wise, in the domain of the forward's F(4,7): ten frequency points, each a weight
 * gradient with 7 x 1 taps over [N][H][W/4] images (17.5 multiplies per pixel instead of 49). */
size_t cnm_conv7x7_wgrad_winograd_workspace_floats(int Cout, int Cin, int N, int H, int W);
int cnm_conv7x7_wgrad_winograd_c4_f32(const float* x, int Gx_total, int gx0, int Cin,
                                      const float* dy, int Gy_total, int gy0, int Cout,
                                      float* dw_oihw, float* ws, size_t ws_floats,
                                      int N, int H, int W, int rot, void* stream);
/* ... and of the 5x5 stride-1 layer (conv2.0) row-wise on F(4,5): eight gradients with 5 x 1 taps, 10 multiplies per pixel instead of 25. */
size_t cnm_conv5x5_wgrad_winograd_workspace_floats(int Cout, int Cin, int N, int H, int W);
int cnm_conv5x5_wgrad_winograd_c4_f32(const float* x, int Gx_total, int gx0, int Cin,
                                      const float* dy, int Gy_total, int gy0, int Cout,
                                      float* dw_oihw, float* ws, size_t ws_floats,
                                      int N, int H, int W, int rot, void* stream);
size_t cnm_conv_s2_wgrad_winograd_workspace_floats(int Cout, int Cin, int ksize, int N, int H, int W);
int cnm_conv_s2_wgrad_winograd_c4_f32(const float* x, int Gx_total, int gx0, int Cin,
                                      const float* dy, int Gy_total, int gy0, int Cout,
                                      float* dw_oihw, float* ws, size_t ws_floats,
                                      int N, int H, int W, int ksize, int rot, void* stream);
int cnm_conv3x3_wgrad_winograd_c4_f32(const float* x, int Gx_total, int gx0, int Cin,
                                      const float* dy, int Gy_total, int gy0, int Cout,
                                      float* dw_oihw, float* ws, size_t ws_floats,
                                      int N, int H, int W, int rot, void* stream);
int cnm_conv2d_wgrad_c4_f32(const float* x, int Gx_total, int gx0, int Cin,
                            const float* dy, int Gy_total, int gy0, int Cout,
                            float* dw_oihw, float* ws, size_t ws_floats,
                            int N, int H, int W, int ksize, int stride, int rot, void* stream);

/* nn.BatchNorm2d in train mode (+ optional ReLU) on a contiguous c4 tensor [N,ceil(C/4),H,W,4]: batch
 * statistics (biased variance to normalise, unbiased for the running update, momentum/eps as torch),
 * running stats updated in place (may be NULL).  stats_ws: 8*ceil(C/4) doubles. */
int cnm_bn_train_forward_c4_f32(const float* x, const float* gamma, const float* beta,
                                float* running_mean, float* running_var, float momentum, float eps, int relu,
                                float* y, float* save_mean, float* save_invstd, double* stats_ws,
                                int N, int C, int H, int W, void* stream);
int cnm_bn_train_backward_c4_f32(const float* x, const float* y, const float* dy, const float* gamma,
                                 const float* save_mean, const float* save_invstd, int relu,
                                 float* dx, float* dgamma, float* dbeta, double* sums_ws,
                                 int N, int C, int H, int W, void* stream);
/* The same two with a workspace of 8*ceil(C/4) doubles that is ZERO when the call starts and left zero by it (allocate and
 * clear it once, for the widest layer; one per stream), and -- forward -- nn.BatchNorm2d.num_batches_tracked (int64 scalar on
 * the device, may be NULL) incremented by the call: no clearing launch and no counter launch per layer. */
int cnm_bn_train_forward_z_c4_f32(const float* x, const float* gamma, const float* beta,
                                  float* running_mean, float* running_var, float momentum, float eps, int relu,
                                  float* y, float* save_mean, float* save_invstd, double* zero_ws, long long* num_batches_tracked,
                                  int N, int C, int H, int W, void* stream);
int cnm_bn_train_backward_z_c4_f32(const float* x, const float* y, const float* dy, const float* gamma,
                                   const float* save_mean, const float* save_invstd, int relu,
                                   float* dx, float* dgamma, float* dbeta, double* zero_ws,
                                   int N, int C, int H, int W, void* stream);

/* ... with `groups` statistics groups: sample n is normalised with the batch statistics of the samples n' = n (mod groups).  A batch that
 * interleaves the `groups` sources of every frame (pair p = b * groups + s) then computes exactly what `groups` separate calls, one per
 * source, compute -- the reference calls depthNet once per source (train.py:164-167) -- with the running statistics and
 * num_batches_tracked updated `groups` times in source order.  save_mean / save_invstd: groups * C floats; zero_ws: 8 * ceil(C/4) *
 * groups doubles, zero on entry, left zero. */
int cnm_bn_train_forward_zg_c4_f32(const float* x, const float* gamma, const float* beta,
                                   float* running_mean, float* running_var, float momentum, float eps, int relu,
                                   float* y, float* save_mean, float* save_invstd, double* zero_ws, long long* num_batches_tracked,
                                   int N, int C, int H, int W, int groups, void* stream);
int cnm_bn_train_backward_zg_c4_f32(const float* x, const float* y, const float* dy, const float* gamma,
                                    const float* save_mean, const float* save_invstd, int relu,
                                    float* dx, float* dgamma, float* dbeta, double* zero_ws,
                                    int N, int C, int H, int W, int groups, void* stream);

/* The grouped backward without the saved output y: the ReLU mask is recomputed from x, gamma, beta and the saved statistics in
 * the forward's own operation order (bit-identical results); two of the seven tensor passes of a BatchNorm backward less, and
 * the autograd tape keeps x only. */
int cnm_bn_train_backward_zgb_c4_f32(const float* x, const float* dy, const float* gamma, const float* beta,
                                     const float* save_mean, const float* save_invstd, int relu,
                                     float* dx, float* dgamma, float* dbeta, double* zero_ws,
                                     int N, int C, int H, int W, int groups, void* stream);
/* [r6] The same BatchNorm in TWO launches per direction: the reductions leave one partial sum per workgroup in fixed slots of
 * partials_ws (cnm_bn_train_partials_doubles(N, C, H, W, groups) doubles, contents irrelevant on entry and exit, one per stream: no
 * atomics, nothing to clear, a fixed summation order), and the elementwise pass sums the <= 64 partials of its own four channels in its
 * prologue -- every workgroup with the same shuffle tree, so the one that records save_mean / save_invstd / the running statistics /
 * dgamma / dbeta holds bit-identical values.  Semantics as cnm_bn_train_forward_zg_c4_f32 / cnm_bn_train_backward_zg(b)_c4_f32 (groups,
 * num_batches_tracked, y == NULL with beta: the ReLU mask recomputed from x); results equal theirs up to the summation order of the
 * fp64 sums.  76 launches less per training step (train.py:164-310). */
size_t cnm_bn_train_partials_doubles(int N, int C, int H, int W, int groups);
int cnm_bn_train_forward_p_c4_f32(const float* x, const float* gamma, const float* beta,
                                  float* running_mean, float* running_var, float momentum, float eps, int relu,
                                  float* y, float* save_mean, float* save_invstd, double* partials_ws, long long* num_batches_tracked,
                                  int N, int C, int H, int W, int groups, void* stream);
int cnm_bn_train_backward_p_c4_f32(const float* x, const float* y, const float* dy, const float* gamma, const float* beta,
                                   const float* save_mean, const float* save_invstd, int relu,
                                   float* dx, float* dgamma, float* dbeta, double* partials_ws,
                                   int N, int C, int H, int W, int groups, void* stream);

/* Masked mean L1 of the training losses -- IdepthLoss / IdepthwithProbLoss (losses.py:30-73) without the `pred[mask]` gather:
 *   out2[0] = sum_m weight |pred - gt| / count(m),  out2[1] = count(m),  m = gt > 0 && finite(gt) && finite(pred) && pred > 0
 * (losses.py:39-40, :61; weight may be NULL = 1; an empty mask gives NaN like the reference's mean of an empty selection).  n elements,
 * any shape.  One launch, fp64 partial sums added in a fixed order (bit-reproducible); zero_ws: cnm_masked_l1_workspace_doubles()
 * doubles, zero on entry, left zero (one per stream).  Backward: grad_out = d loss / d out2[0] (device scalar), out2 as the forward
 * wrote it; dpred / dweight [n] (either may be NULL): w sign(pred - gt) grad / count and |pred - gt| grad / count on the mask, 0 off it. */
size_t cnm_masked_l1_workspace_doubles(void);
int cnm_masked_l1_f32(const float* pred, const float* gt, const float* weight, long long n, double* zero_ws, float* out2, void* stream);
int cnm_masked_l1_backward_f32(const float* pred, const float* gt, const float* weight, const float* grad_out, const float* out2,
                               long long n, float* dpred, float* dweight, void* stream);
/* [r6] Surface-normal loss terms (reference losses.py:76-122, train.py:226-263) per sample: s[b] = sum over kept pixels of 1 - cos(pred, gt),
 * c[b] = kept pixels; kept = valid && finite(sum_c gt) && finite(sum_c pred); cos as torch's cosine_similarity(dim = 1, eps = 1e-8).
 * pred, gt [B,3,H,W] fp32, valid [B,H,W] bytes (torch bool), HW = H * W; ws: cnm_normal_cos_workspace_doubles(B) doubles, contents irrelevant.
 * Two launches forward (fp64 block partials in fixed slots, summed in block order: bit-reproducible), one backward (grad_s [B] -> dpred). */
size_t cnm_normal_cos_workspace_doubles(int B);
int cnm_normal_cos_terms_f32(const float* pred, const float* gt, const unsigned char* valid, int B, int HW, double* ws, float* s_out, float* c_out, void* stream);
int cnm_normal_cos_terms_backward_f32(const float* pred, const float* gt, const unsigned char* valid, const float* grad_s, int B, int HW, float* dpred, void* stream);

/* Backward of cnm_head_sigmoid_c4_f32 -- depth_layer = Conv2d(C, 1, 3, padding = 1) + Sigmoid, times `scale`
 * (depthNet_model.py:82-84, :246): with ds = grad_disp * disp * (1 - disp / scale),
 *   dx [N, C/4, H, W, 4] (contiguous) = transposed 3x3 of ds with w,  dw_oihw [1,C,3,3] = sum_p ds(p) x(p + tap),  dbias [1] = sum ds.
 * x is a channel-group view (Gx_total, gx0) as everywhere; w_oihw is the module's own weight [1,C,3,3]; grad_disp / disp [N,1,H,W].
 * dx or dw_oihw (with dbias, which may be NULL) may be NULL.  Streaming kernels (x read once, dx written once); the weight
 * gradient adds fp64 block partials in a fixed order.  ws: cnm_head_backward_workspace_doubles(C) doubles (no entry state). */
size_t cnm_head_backward_workspace_doubles(int C);
int cnm_head_backward_c4_f32(const float* x, int Gx_total, int gx0, int C, const float* w_oihw, const float* grad_disp, const float* disp,
                             float scale, float* dx, float* dw_oihw, float* dbias, double* ws, int N, int H, int W, void* stream);

/* Adjoint of cnm_upsample2x_c4_f32: dy [N,G,2H,2W,4] -> dx [N,G,H,W,4] (contiguous). */
int cnm_upsample2x_backward_c4_f32(const float* dy, float* dx, int N, int G, int H, int W, void* stream);

/* ---------------------------------------------------------------- depth -> normal (K6)
 * Replaces Depth2normal.forward without the plane branch (depth_util.py:149-203):
 * depth [B,H,W], K_inv [B,3,3] -> normal [B,3,H,W], points [B,3,H,W]. k odd, 1..15.
 * input_is_idepth != 0: `depth` holds inverse depth and z = 1/idepth is taken on load
 * (the reference's call sites do this in torch first: eval.py:452, train.py:185-186). */
int cnm_depth2normal_f32(const float* depth, const float* K_inv, float* normal, float* points,
                         int B, int H, int W, int ksize, int input_is_idepth, void* stream);

/* Plane-instance regularisation of a normal map, in place: the plane branch of Depth2normal.forward
 * (depth_util.py:205-238) and get_normal_by_planes (:243-278).  normal [B,3,H,W]; instance_segs [B,P,H,W] bytes
 * (non-zero = inside); planes_num [B] ints on the DEVICE, max_planes_num = their maximum (host).  For every image
 * and instance i < planes_num[b], in order: the instance's pixels are replaced by their mean normal, and
 * loss_terms[b*P+i] = mean over ALL pixels of 1 - cos(mean, inside ? normal : 0) (the reference's loss is the
 * sum of these terms; loss_terms may be NULL; entries of unused instances are left untouched). */
int cnm_plane_normals_f32(float* normal, const unsigned char* instance_segs, const int* planes_num, int max_planes_num,
                          float* loss_terms, int B, int P, int H, int W, void* stream);

/* Backward of cnm_depth2normal_f32 w.r.t. its depth (or inverse-depth) input, as needed by the normal losses of
 * train.py:204-263: grad_normal [B,3,H,W], grad_points [B,3,H,W] or NULL -> grad_depth [B,H,W].
 * ws: 9*B*H*W floats.  The validity mask (0 < z < 10) and the det < 1e-5 branch are treated as constants. */
int cnm_depth2normal_backward_f32(const float* depth, const float* K_inv, const float* grad_normal,
                                  const float* grad_points, float* grad_depth, float* ws,
                                  int B, int H, int W, int ksize, int input_is_idepth, void* stream);

/* K^-1 of the intrinsics stored in a camera tensor (cam[:,1,:3,:3].inverse(), train.py:201-202,
 * eval.py:271): cam [B,2,4,4] with cam_stride floats between images -> K_inv [B,3,3]. */
int cnm_intrinsics_inverse_f32(const float* cam, long long cam_stride, float* K_inv, int B, void* stream);

/* ---------------------------------------------------------------- inverse warp (K7)
 * Replaces inverse_warp / pixel2cam / cam2pixel (depthnet/inverse_warp.py:27-118),
 * padding_mode='zeros'.  feat [B,C,H,W], depth [B,H,W], pose [B,3,4], K,K_inv [B,3,3]. */
int cnm_inverse_warp_f32(const float* feat, const float* depth, const float* pose,
                         const float* K, const float* K_inv, float* out,
                         int B, int C, int H, int W, void* stream);
/* The same with the padding_mode the reference hands through to grid_sample (inverse_warp.py:81,116): 0 'zeros' (the
 * function above: out-of-view coordinates forced to 2, :71-75), 1 'border', 2 'reflection' (no out-of-view masking in the
 * reference for these two; align_corners=False).  Forward only -- the training path uses 'zeros'. */
int cnm_inverse_warp_pad_f32(const float* feat, const float* depth, const float* pose,
                             const float* K, const float* K_inv, float* out,
                             int B, int C, int H, int W, int padding_mode, void* stream);

/* Backward of cnm_inverse_warp_f32 w.r.t. the target depth (train.py:284-293 differentiates the sampling
 * position; the sampled map itself carries no gradient there): grad_out [B,C,H,W] -> grad_depth [B,H,W]. */
int cnm_inverse_warp_backward_depth_f32(const float* feat, const float* depth, const float* pose,
                                        const float* K, const float* K_inv, const float* grad_out,
                                        float* grad_depth, int B, int C, int H, int W, void* stream);

/* ---------------------------------------------------------------- host twins ("_cpu")
 * every operator of the path with a twin taking HOST pointers; BASELINE configs[0] ("DepthNet eval, 1 ref +
 * 1 src, 256x192, 32 planes, batch=1 on CPU: plumbing, no GPU").  Plain C++ in the same library (csrc/host_twins.cpp and
 * the EngHost policy of csrc/nets.hip): same arguments without the stream, same c4 layout, same return codes; direct
 * convolutions, one thread per core (CNM_CPU_THREADS overrides).  They are the product's CPU path -- cnmnet_amd.depthnet
 * routes CPU tensors here in eval mode -- and never the measured one; oracle/ is not involved.
 * Filters: cnm_pack_conv_bn_cpu folds eval-mode BatchNorm exactly as cnm_pack_conv_bn_f32 does and lays the filter out as
 * w_packed [Cout][k*k][4*ceil(Cin/4)] (channel position (ci + Cin - rot) % Cin), b_packed [Cout]; heads: cnm_pack_head_cpu.
 * The whole-network twins take cnm_layer_weights whose .w / .b point at those (all other slots unused). */
int cnm_homography_terms_cpu(const float* ref_cam, const float* src_cam, float* hmkt, int B, int S);             /* depth_util.py:13-56 */
int cnm_planesweep_volume_nchw_cpu(const float* ref, const float* src, const float* hmkt, float* volume,
                                   int B, int S, int H, int W, int D, double idepth_min, double idepth_max);     /* depthNet_model.py:185-224 */
int cnm_planesweep_cat_c4_cpu(const float* ref, const float* src, const float* hmkt, float* x,
                              int B, int S, int H, int W, int D, double idepth_min, double idepth_max);          /* + :233 */
size_t cnm_packed_conv_floats_cpu(int Cout, int Cin, int ksize);
int cnm_pack_conv_bn_cpu(const float* w_oihw, const float* bn_gamma, const float* bn_beta, const float* bn_mean, const float* bn_var,
                         const float* bias, float eps, int Cout, int Cin, int ksize, int rot, float* w_packed, float* b_packed);
int cnm_pack_head_cpu(const float* w_oihw, int C, float* w_head);
int cnm_conv2d_cat2_c4_cpu(const float* in_a, int Ga_total, int ga0, int Ga, const float* in_b, int Gb_total, int gb0, int Gb,
                           float* out, int Gout_total, int gout0, int Cout, const float* w_packed, const float* b_packed,
                           int N, int H, int W, int ksize, int stride, int relu);                                 /* depthNet_model.py:19-79 */
int cnm_upsample2x_c4_cpu(const float* in, int Gin_total, int gin0, float* out, int Gout_total, int gout0,
                          int N, int G, int H, int W);                                                            /* :94,105 */
int cnm_head_sigmoid_c4_cpu(const float* in, int Gin_total, int gin0, int C, const float* w_head, const float* bias, float scale,
                            float* disp, float* up_out, int up_Gtotal, int up_g, int N, int H, int W);            /* :82-84,246-261 */
int cnm_nchw_to_c4_cpu(const float* nchw, float* c4, int G_total, int g0, int N, int C, int H, int W);
int cnm_c4_to_nchw_cpu(const float* c4, int G_total, int g0, float* nchw, int N, int C, int H, int W);
int cnm_intrinsics_inverse_cpu(const float* cam, long long cam_stride, float* K_inv, int B);
int cnm_depth2normal_cpu(const float* depth, const float* K_inv, float* normal, float* points,
                         int B, int H, int W, int ksize, int input_is_idepth);                                    /* depth_util.py:149-203 */
int cnm_inverse_warp_cpu(const float* feat, const float* depth, const float* pose, const float* K, const float* K_inv,
                         float* out, int B, int C, int H, int W);                                                 /* inverse_warp.py:27-118 */
size_t cnm_depthnet_workspace_floats_cpu(int P, int H, int W, int D);
int cnm_depthnet_forward_cpu(const cnm_layer_weights* weights, float idepth_scale, int D,
                             const float* ref, const float* src, const float* ref_cam, const float* src_cam,
                             float* disp1, float* disp2, float* disp3, float* disp4, float* iconv1_c4,
                             float* ws, size_t ws_floats, int B, int S, int H, int W);                            /* depthNet_model.py:226-263 */
size_t cnm_refinenet_workspace_floats_cpu(int N, int H, int W);
int cnm_refinenet_forward_cpu(const cnm_layer_weights* weights, float idepth_scale,
                              const float* idepth01, const float* idepth02, long long idepth_stride,
                              const float* iconv01, int G1_total, int g1, const float* iconv02, int G2_total, int g2,
                              float* disp_refined, float* prob_map, float* iconv1_depth_c4,
                              float* ws, size_t ws_floats, int N, int H, int W);                                  /* :331-370 */
int cnm_refinenet_forward_multi_cpu(const cnm_layer_weights* weights, float idepth_scale,
                                    const float* idepth_pairs, const float* iconv_pairs_c4, int S,
                                    float* disp_refined, float* prob_map, float* iconv1_depth_c4,
                                    float* ws, size_t ws_floats, int B, int H, int W);                            /* eval.py:635-663,885-929 */

#ifdef __cplusplus
}
#endif
#endif /* CNM_ENGINE_H */
