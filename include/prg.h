/* prg.h — C-ABI of libprg_hip.so: the MI355X (gfx950) implementation of PointRegGPT's generative data path.
 *
 * The reference (Chen-Suyi/PointRegGPT) has no FFI: this path sits behind plain Python callables.  Each
 * entry point below replaces one of those callables; the reference interface it stands in for is cited as
 *   sd = denoising_diffusion_pytorch/successive_ddnm_diffusion.py      dc = depth_correction_pytorch/depth_correction.py
 * INTEGRATION.md shows the ctypes stub a maintainer of the reference would add to bind them.
 *
 * Conventions
 *   - extern "C", C types only.  Every function returns 0 on success or a negative PRG_E_* code and never
 *     throws or aborts across the boundary; prg_last_error() returns a thread-local message.
 *   - The CALLER owns all tensor memory.  Unless a parameter is documented "host", pointers are DEVICE
 *     pointers (hipMalloc / torch.empty(device='cuda').data_ptr()).  Images are dense row-major float32
 *     (B,1,H,W) exactly as the reference passes them; depth unit: the caller's (the kernels are unit-free).
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream).  Calls are asynchronous with
 *     respect to it and do not synchronise, except *_create / *_destroy / *_reserve.
 *   - The library allocates device memory only inside opaque handles (weights, workspaces, graphs).
 *     Handles are not thread-safe; use one per (process, device).
 */
#ifndef PRG_H
#define PRG_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PRG_ABI_VERSION 1

enum {
  PRG_OK = 0,
  PRG_E_INVALID = -1,   /* bad argument (null pointer, size mismatch, unsupported shape) */
  PRG_E_HIP = -2,       /* a HIP runtime call failed; message has hipGetErrorString */
  PRG_E_NOMEM = -3,     /* device allocation failed */
  PRG_E_STATE = -4      /* handle used in the wrong state */
};

/* storage + MFMA input type of a U-Net handle.  PRG_MXFP8 (BASELINE configs[4]): activations stored in bf16, every 3x3
 * convolution with 64-channel-multiple widths runs on v_mfma_scale_f32_32x32x64_f8f6f4 with OCP e4m3 operands and one
 * E8M0 scale per 32 channels (weights quantised at prg_unet_create, activations while they are staged); the rest = bf16. */
/* PRG_F16X3 (round 4): float32 storage and float32 / float64 normalisation arithmetic exactly as PRG_F32, but every
 * convolution contracts on the f16 matrix pipe with both operands split into two f16 halves (a = hi + lo, three MFMAs per
 * product tile, 22-bit operands; csrc/conv_split.hip).  NOT labelled "parity" (that is PRG_F32 alone): its distance from the
 * reference is MEASURED per chain — point-XYZ L-infinity at the end of round 5: 4.5e-6 m (1000-step ancestral @64x64), 3.8e-5 m
 * (250-step DDIM @128x128), 5.3e-5 m (250-step DDIM @256x256, the shipped setting), 7.6e-6 m (the benchmarked 1000-step chain
 * @128x128), all inside the 1e-4 m north star since round 5 (round 4: 1.7e-4 m at 256x256 — the lo halves of unstandardised
 * weights were subnormal f16; the packer now scales each output channel by an exact power of two).  The 256x256 figure is a
 * draw: equal-precision re-orderings of the arithmetic move it within 3.7e-5 .. 8.8e-5 m (the reference's own 1-vs-8-thread
 * spread there is 4.7e-5 m).  tests/test_gpu_f16x3.py asserts these and bench.py's drift_vs_reference re-measures them in every
 * run; 3.9-4.0x PRG_F32's throughput.
 * Operand range: an activation or scaled weight beyond f16's 65504 becomes inf (visible, not silent).                      */
enum { PRG_F32 = 0, PRG_BF16 = 1, PRG_MXFP8 = 2, PRG_F16X3 = 3 };

int prg_abi_version(void);
const char* prg_last_error(void);
/* Name and compute-unit count of the current HIP device (diagnostics; fails loudly when there is none). */
int prg_device_info(char* name, size_t name_len, int* compute_units);

/* ------------------------------------------------------------------------------------------------------
 * Geometry (memory-bound kernels)
 * ---------------------------------------------------------------------------------------------------- */

/* depth2pc_tensor (sd:176-209): depth (B,1,H,W) + K (B,3,3) -> pc (B,H*W,3), valid (B,H*W) as bytes.
 * valid = clip_lo < depth < clip_hi (pass clip_lo > clip_hi to disable clipping, i.e. clip=None);
 * invalid points are filled with `invalid_value` (the reference default is NaN).                          */
int prg_depth2pc(const float* depth, const float* K, float* pc, uint8_t* valid, int B, int H, int W,
                 float clip_lo, float clip_hi, float invalid_value, void* stream);

/* pc2depth_tensor (sd:212-265): z-buffer.  pc (B,N,3), valid (B,N) bytes or NULL (= all valid), K (B,3,3)
 * -> depth (B,1,H,W) nearest z per pixel (0 where nothing lands), mask (B,1,H,W) bytes.
 * Pixel = round-half-even(x*fx/z + cx, y*fy/z + cy); a point counts iff in frame, valid and z > 0.         */
int prg_pc2depth(const float* pc, const uint8_t* valid, const float* K, float* depth, uint8_t* mask,
                 int B, int N, int H, int W, void* stream);

/* Generator.generate's per-scene form (sd:2531-2547): ragged clouds, CSR offsets (B+1, int64, device),
 * each moved by its pose (B,4,4) as p R^T + t (NULL = identity) and z-buffered with its K.
 * `depth_scale` multiplies the stored depth (the reference multiplies by 0.1 right after, sd:2552).          */
int prg_project_points_zbuffer(const float* points, const int64_t* offsets, const float* pose, const float* K,
                               float* depth, uint8_t* mask, int B, int H, int W, float depth_scale,
                               void* stream);

/* reproject_tensor (sd:268-286) fused: unproject depth*depth_unit, move by pose, z-buffer into the same
 * camera, store z*out_scale.  One kernel, no intermediate point cloud in HBM.                              */
int prg_reproject_zbuffer(const float* depth, const float* K, const float* pose, float* depth_out,
                          uint8_t* mask_out, int B, int H, int W, float depth_unit, float clip_lo,
                          float clip_hi, float out_scale, void* stream);

/* numpy point_cloud + inverse pose (sd:122-143, sd:2627-2628) in float64 like the reference's numpy path:
 * depth (B,1,H,W) float32 * depth_unit -> xyz (B,H*W,3) float64 in the common frame, R^T (p - t), pose NULL =
 * camera frame; rows of invalid pixels are NaN and valid (B,H*W) bytes says which to keep (row-major order
 * is the reference's order after compaction).                                                              */
int prg_unproject_f64(const float* depth, const float* K, const float* pose, double* xyz, uint8_t* valid,
                      int B, int H, int W, float depth_unit, float clip_lo, float clip_hi, void* stream);

/* DepthAugment (dc:577-604): depth (B,1,H,W) -> (B,3,H,W) [depth, 3x3 min over non-zero, difference].   */
int prg_depth_augment(const float* depth, float* out, int B, int H, int W, void* stream);

/* Generator.generate's mask application (sd:2564-2570): keep = prob > thr; depth[~keep] = 0 (in place when
 * depth_out == depth); hit &= keep; img_cond (B,2,H,W) = cat[depth, hit] * 2 - 1 (NULL to skip).
 * `hit` may be NULL (treated as all true, the post-sampling use at sd:2579-2581).                          */
int prg_apply_mask(const float* prob, const float* depth, const uint8_t* hit, float thr, float* depth_out,
                   uint8_t* hit_out, float* img_cond, int B, int H, int W, void* stream);

/* occlusion_filter of Tester.sample (sd:446-463): depth (B,1,H,W) [metres], mask (B,1,H,W) bytes -> out: every pixel
 * more than `threshold` (0.0375) behind the nearest VALID depth of its 3x3 window takes that depth.  out != depth.   */
int prg_occlusion_filter(const float* depth, const uint8_t* mask, float* out, int B, int H, int W, float threshold,
                         void* stream);

/* compute_overlap_ratio of generate_gt.py:68-102 for a batch of cloud pairs, after the caller's voxel down-sampling:
 * pts (total,3) float64 DEVICE, offsets (2*n_pairs+1) int64 DEVICE — pair p is clouds [off[2p],off[2p+1]) and
 * [off[2p+1],off[2p+2]) — counts (n_pairs,2) int32 DEVICE: points of the first / second cloud that have a point of
 * the other strictly within `radius` (float64 squared distances, exact all-pairs test).  max_cloud = largest cloud.  */
int prg_overlap_counts(const double* pts, const int64_t* offsets, int n_pairs, int64_t max_cloud, double radius,
                       int32_t* counts, void* stream);

/* ------------------------------------------------------------------------------------------------------
 * U-Nets (MFMA kernels)
 * ---------------------------------------------------------------------------------------------------- */

typedef struct prg_unet prg_unet;

typedef struct prg_unet_config {
  int32_t dim;               /* base width (64) */
  int32_t n_levels;          /* len(dim_mults) (4) */
  int32_t dim_mults[8];      /* (1,2,4,8) */
  int32_t in_channels;       /* Unet 1 ; MaskUnet 3 (DepthAugment is applied inside prg_maskunet_forward) */
  int32_t conditional;       /* 1: time + camera-parameter conditioning (Unet, sd:802) ; 0: MaskUnet (dc:807) */
  int32_t param_cond_dim;    /* 4 */
  int32_t groups;            /* GroupNorm groups (8) */
  int32_t sigmoid_out;       /* 1: final Sigmoid (MaskUnet) */
} prg_unet_config;

/* Number of float32 parameters the config implies (= sum of the reference module's state_dict sizes).    */
int64_t prg_unet_param_count(const prg_unet_config* cfg);

/* weights: HOST pointer to n_floats float32 = every state_dict tensor of the reference module, flattened and
 * concatenated in state_dict order (sd:802-918 / dc:807-869; pointreggpt_amd.weights.param_spec lists it).
 * The library standardises the Block conv weights (sd:601-616, eps 1e-5), repacks everything for its
 * kernels in `dtype` and uploads it.                                                                        */
int prg_unet_create(const prg_unet_config* cfg, const float* weights, int64_t n_floats, int dtype,
                    prg_unet** out);
int prg_unet_destroy(prg_unet* h);
/* SinusoidalPosEmb frequencies (sd:645-657): freqs (HOST, dim/2 float32) = exp(arange(dim/2) * -ln(1e4)/(dim/2-1)).  The
 * reference evaluates this float32 exp with torch on its own device and its outputs depend on that at the 4e-5 level over
 * a 50-step chain (1 ulp of a frequency x t <= 999), so the table is the caller's: pass what torch computes on the host
 * (pointreggpt_amd.unet does).  Default: this host's libm.  Call before creating samplers on the handle.            */
int prg_unet_set_time_freqs(prg_unet* h, const float* freqs, int n);
/* Pre-size the activation workspace for (B, S) so later forwards never allocate.                          */
int prg_unet_reserve(prg_unet* h, int B, int S);

/* Unet.forward (sd:920-964): x (B,1,S,S), time (B,) int64 DEVICE, param_cond (B,4) -> out (B,1,S,S).      */
int prg_unet_forward(prg_unet* h, const float* x, const int64_t* time, const float* param_cond, float* out,
                     int B, int S, void* stream);
/* MaskUnet.forward (dc:871-906): depth (B,1,S,S) -> keep-probability (B,1,S,S).                            */
int prg_maskunet_forward(prg_unet* h, const float* depth, float* prob, int B, int S, void* stream);

/* Debug taps for kernel unit tests: after a forward, copy an internal activation (converted to float32 NCHW)
 * into `out` (device).  Names: "init_conv","down0_block0","down0_attn","down0_out","mid_attn","up0_out",
 * "final_res","augment".  *C,*H,*W receive its shape.  Enabled by prg_unet_set_taps(h, 1).                  */
int prg_unet_set_taps(prg_unet* h, int enable);
int prg_unet_get_tap(prg_unet* h, const char* name, float* out, int64_t out_capacity_floats, int* C, int* H,
                     int* W, void* stream);

/* Kernel unit-test hook: one 3x3 / stride 1 / pad 1 convolution through the library's own dispatch in `dtype` (PRG_BF16
 * or PRG_MXFP8).  x (B,Cin,H,W) float32 DEVICE, w (Cout,Cin,3,3) float32 HOST (used as is: no standardisation), bias
 * (Cout) float32 HOST or NULL, out (B,Cout,H,W) float32 DEVICE (the bf16 result widened).  Synchronises.           */
int prg_debug_conv3x3(const float* x, const float* w, const float* bias, float* out, int B, int Cin, int Cout, int H, int W,
                      int dtype, void* stream);
/* Kernel unit-test hook (round 4): the two convolutions of a ResnetBlock's Block pair (sd:681-697, 731-733) in bf16 mode through
 * the library's own dispatch:  h = conv3x3(x, w1) + b1, its GroupNorm statistics taken in the epilogue;
 * y = conv3x3(SiLU(GroupNorm_groups(h) * gamma + beta), w2) + b2 with the norm in conv2's fused prologue.  h16 = 0: h is stored as
 * bf16; h16 = 1: as f16, prologue in packed f16, f16 MFMA operands in conv2 (PRG_E_INVALID when the shape's kernels do not
 * implement that).  x (B,Cin,H,W) float32 DEVICE; w1 (C,Cin,3,3), w2 (C,C,3,3), b1, b2, gamma, beta (C) float32 HOST, used as
 * they are; out (B,C,H,W) float32 DEVICE.  Cin, C multiples of 64.  Synchronises.                                              */
int prg_debug_block_pair(const float* x, const float* w1, const float* b1, const float* gamma, const float* beta, const float* w2,
                         const float* b2, float* out, int B, int Cin, int C, int H, int W, int groups, int h16, void* stream);
/* The same for Upsample = nn.Upsample(scale_factor 2, nearest) + Conv2d(Cin, Cout, 3, pad 1) (sd:592-594) in bf16:
 * out (B,Cout,2H,2W).  Shapes the 256-pixel kernel covers run as four 2 x 2-tap sub-pixel convolutions of the source image.   */
int prg_debug_upsample_conv3x3(const float* x, const float* w, const float* bias, float* out, int B, int Cin, int Cout, int H,
                               int W, void* stream);
/* The same for Downsample's Conv2d(Cin, Cout, 4, stride 2, pad 1) (sd:596-597) in bf16: w (Cout,Cin,4,4),
 * out (B,Cout,H/2,W/2).                                                                                             */
int prg_debug_conv4x4s2(const float* x, const float* w, const float* bias, float* out, int B, int Cin, int Cout, int H, int W,
                        void* stream);
/* General form for the float32-storage modes (PRG_F32 / PRG_F16X3): K x K kernel (1, 3 or 4), stride 1 or 2, pad = 0 for
 * K = 1 and 1 otherwise; w (Cout,Cin,K,K), out (B,Cout,Ho,Wo) float32.  Also accepts PRG_BF16 / PRG_MXFP8 for K = 3 / 4.  */
int prg_debug_conv(const float* x, const float* w, const float* bias, float* out, int B, int Cin, int Cout, int H, int W,
                   int dtype, int K, int stride, void* stream);
/* nn.Upsample(x2, nearest) + Conv2d(3, pad 1) (sd:592-594) in any storage mode (PRG_F16X3: the sub-pixel form of the split
 * wave-specialised kernel when Cout % 128 == 0; round 5).  out: (B, Cout, 2H, 2W) float32. */
int prg_debug_upsample_conv(const float* x, const float* w, const float* bias, float* out, int B, int Cin, int Cout, int H, int W,
                            int dtype, void* stream);

/* ------------------------------------------------------------------------------------------------------
 * Sampler: GaussianDiffusion.sample / p_sample_loop / ddim_sample (sd:1283-1409), DDNM replacement included
 * ---------------------------------------------------------------------------------------------------- */

typedef struct prg_sampler prg_sampler;

/* One denoising transition.  With u = Unet(x, t, param_cond):
 *     x0p = (clip_pred & 1) ? clamp(u,-1,1) : u                              (ddim_sample, sd:1199-1201)
 *     eps = (sqrt_recip * x - x0p) / sqrt_recipm1                            (sd:1158-1162; used iff c_eps != 0)
 *     x0  = known ? cond_depth : x0p                                         (DDNM replacement, sd:1210-1218)
 *     x0  = (clip_pred & 2) ? clamp(x0,-1,1) : x0                            (p_mean_variance, sd:1250-1251: the ancestral
 *           sampler clamps AFTER the replacement; ddim_sample does not, so known pixels > 1 enter the state as they are)
 *     x'  = c_x0 * x0 + c_x * x + c_eps * eps + sigma * noise                (sd:1173-1180,1280 / sd:1369-1373)
 * Ancestral step t: c_x0 = posterior_mean_coef1[t], c_x = coef2[t], c_eps = 0, sigma = exp(0.5 logvar[t])
 * (0 at t = 0).  DDIM pair (t, t'): c_x0 = sqrt(ac[t']), c_x = 0, c_eps = c, sigma = sigma; last pair:
 * c_x0 = 1, everything else 0.  The host (pointreggpt_amd.diffusion) fills this table from the float64
 * schedule exactly as the reference computes it.                                                            */
typedef struct prg_step {
  int32_t t;            /* timestep fed to the U-Net */
  int32_t clip_pred;    /* bit 0: clamp the network output before deriving eps (DDIM rows = 1);
                           bit 1: clamp x0 after the DDNM replacement (ancestral rows = 2);
                           bit 2 (= 4, alone): refine row of has_refine_step (sd:1307-1314, 1374-1388):
                                  x' = known ? clamp(u,-1,1) : x, no replacement, coefficients ignored */
  float c_x0, c_x, c_eps, sigma;
  float sqrt_recip, sqrt_recipm1;
} prg_step;

/* steps: HOST array of n_steps transitions, executed in order.  The handle owns the state image, the
 * per-step conditioning table and one captured hipGraph of a full transition (U-Net + update) that is
 * replayed n_steps times; the step index lives in device memory so no host sync occurs inside a run.      */
int prg_sampler_create(prg_unet* unet, const prg_step* steps, int n_steps, int B, int S, prg_sampler** out);
int prg_sampler_destroy(prg_sampler* h);
/* 1 (default): replay the captured hipGraph; 0: launch every kernel eagerly (debug / profiling).          */
int prg_sampler_set_graph(prg_sampler* h, int enable);

/* param_cond (B,4); img_cond (B,2,S,S) in [-1,1] or NULL (unconditional: no DDNM replacement);
 * noise: NULL -> on-device Philox4x32-10 keyed per scene by seeds[b] (HOST array of B uint64; results do
 * not depend on batch composition or rank) ; else DEVICE float32 (noise_slabs, B, S, S) in the reference's
 * draw order (sd:1293,1279 / sd:1339,1369): slab 0 is the start image, slab k+1 feeds transition k and is
 * read only where sigma != 0 — both samplers draw nothing on their last transition, so noise_slabs = n_steps
 * suffices; the call fails with PRG_E_INVALID if a transition with sigma != 0 would read past noise_slabs.
 * out (B,1,S,S) = (x_final + 1) * 0.5  (sd:1316).                                                          */
int prg_sampler_run(prg_sampler* h, const float* param_cond, const float* img_cond, const float* noise,
                    int64_t noise_slabs, const uint64_t* seeds, float* out, void* stream);

/* Kernel unit-test / bandwidth hook (round 6): the transition update above ALONE (sampler_step_kernel: what p_sample / ddim_sample
 * do after model_predictions, sd:1257-1281 / sd:1369-1373) on caller tensors, `reps` launches back to back on `stream`:
 * x (B,HW) DEVICE, updated in place by every launch; u (B,HW) DEVICE = the network output; img_cond (B,2,HW) DEVICE or NULL;
 * seeds (B) DEVICE uint64 Philox keys (launch i draws noise index i + 1); step: HOST, the same row for every launch.
 * *avg_us (HOST, may be NULL) = HIP-event microseconds per launch.  Synchronises.                                            */
int prg_debug_sampler_step(float* x, const float* u, const float* img_cond, const uint64_t* seeds, const prg_step* step, int B,
                           int HW, int reps, float* avg_us, void* stream);

/* Wall-clock free timing hook for bench.py: average duration in milliseconds of the dominant kernel class
 * (implicit-GEMM convolution launches) measured with HIP events on the run's own stream during the last
 * prg_sampler_run when profiling was enabled with prg_sampler_set_profile(h, 1) (forces eager launches).
 * conv_ms = total time inside conv launches, conv_launches = their count, conv_flops = their 2*MAC count. */
int prg_sampler_set_profile(prg_sampler* h, int enable);
int prg_sampler_get_profile(prg_sampler* h, double* conv_ms, int64_t* conv_launches, double* conv_flops,
                            double* total_ms);
/* Algorithmic bytes of the same launches (each input and output element once, plus the weights): what the PMC-measured
 * HBM traffic of bench.py's `roofline.traffic` is compared with. */
int prg_sampler_get_profile_bytes(prg_sampler* h, double* conv_bytes);
/* 2 * MAC count the same launches EXECUTED: equal to conv_flops except for Upsample convs that ran as four 2 x 2-tap sub-pixel
 * convolutions (4 / 9 of the algorithmic count, which stays the reference operator's). */
int prg_sampler_get_profile_executed(prg_sampler* h, double* conv_flops_executed);
/* Per-SHAPE totals of the same launches (round 5; bench.py `roofline.per_kernel`): one row per distinct convolution shape of the
 * profiled run — launches, milliseconds inside them (HIP events), algorithmic and executed 2 * MAC counts.  rows may be null with
 * max_rows = 0 to query the row count; at most max_rows rows are written, *n_rows receives the number available.
 * No reference counterpart (sd: has no profiler on this path): measurement hook like prg_sampler_get_profile. */
typedef struct prg_profile_shape {
  int32_t cin, cout, k, stride, ups, hout, wout;   /* Conv2d(cin, cout, k, stride) on (hout, wout) outputs; ups: after nn.Upsample(x2) */
  int32_t two_source, prologue;                     /* virtual concat of two tensors (skip connection); fused GroupNorm+SiLU on the input */
  int32_t mx;                                       /* 1: the launches ran on MX-fp8 operands (scale-MFMA); occupies the former padding */
  int64_t launches;
  double ms, flops, flops_executed;
} prg_profile_shape;
int prg_sampler_get_profile_shapes(prg_sampler* h, prg_profile_shape* rows, int32_t max_rows, int32_t* n_rows);
/* The same for the per-transition update kernel (x0 / DDNM replace / posterior / noise: HBM-bound, 20 B per pixel). */
int prg_sampler_get_profile_step(prg_sampler* h, double* step_ms, int64_t* step_launches);

/* ------------------------------------------------------------------------------------------------------
 * Host post-processing of the generated views (HOST pointers; plain C++ threads, no device work)
 *
 * Replaces what Generator.generate delegates to open3d / torchvision / cv2 after every batch (sd:2484-2500,
 * 2586-2685): compaction, rigid moves, PointCloud.crop(AxisAlignedBoundingBox), voxel_down_sample, io.write_point_cloud,
 * utils.save_image, cv2.imwrite, np.savetxt.  The pool runs them on worker threads while the GPU samples the next
 * batch.  Semantics = pointreggpt_amd/postprocess.py (Open3D 0.17 as recalled: parity-unpinned).
 * ---------------------------------------------------------------------------------------------------- */

/* pts (n,3) float64 -> out (<= n,3), *n_out; keep lo <= p <= hi (inclusive).                                     */
int prg_host_crop_aabb(const double* pts, int64_t n, const double* lo, const double* hi, double* out, int64_t* n_out);
/* Voxel-grid mean: voxel = floor((p - (min - voxel/2)) / voxel); out (<= n,3) in ascending voxel order.         */
int prg_host_voxel_down_sample(const double* pts, int64_t n, double voxel, double* out, int64_t* n_out);
/* binary_little_endian PLY with `double x y z` vertices (what the example dataloaders read).                    */
int prg_host_write_ply(const char* path, const double* pts, int64_t n);

typedef struct prg_pool prg_pool;
int prg_pool_create(int n_threads, prg_pool** out);
int prg_pool_destroy(prg_pool* p);              /* finishes queued jobs first */
/* Block until every submitted job has finished; returns the first job error (message via prg_last_error).       */
int prg_pool_wait(prg_pool* p, int64_t* jobs_done);
/* One cloud file (inputs are copied; the call returns immediately): xyz (n,3) float64 rows with valid[i] != 0 (NULL =
 * all) -> T_pre (4x4 row-major or NULL) -> crop to [lo,hi] (if crop) -> voxel mean (if voxel > 0) -> T_post -> PLY.
 * sample-000000: (valid, NULL, crop, 0.025, NULL) (sd:2484-2500); sample-000001: (valid, pose0, crop, 0.025,
 * pose0^-1) (sd:2641-2658).                                                                                    */
int prg_pool_submit_cloud(prg_pool* p, const char* path, const double* xyz, int64_t n, const uint8_t* valid,
                          const double* T_pre, int crop, const double* lo, const double* hi, double voxel,
                          const double* T_post);
/* img (H,W) float32 in [0,1]: kind 0 = utils.save_image (8-bit RGB, x*255+0.5 clamped; sd:2588-2612),
 * kind 1 = cv2.imwrite of uint16(img * 1e4) (sd:2618-2620).                                                     */
int prg_pool_submit_image(prg_pool* p, const char* path, const float* img, int H, int W, int kind);
/* np.savetxt(path, values (rows, cols)) with the default "%.18e" format (sd:2462-2467, 2555-2561).              */
int prg_pool_submit_text(prg_pool* p, const char* path, const double* values, int rows, int cols);

#ifdef __cplusplus
}
#endif
#endif /* PRG_H */
