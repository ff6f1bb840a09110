/* prg_cpu.h — C-ABI of libprg_cpu.so: plain C++ / OpenMP twins of the hot path with HOST pointers (SURVEY.md section 8b:
 * "the prg_cpu_* twins").
 *
 * What it is for: BASELINE configs[0] (`generate_dataset.py -start=0 -stop=1 on CPU, 64x64, 50-step DDIM`: the plumbing run
 * that needs no GPU, selected EXPLICITLY with `--device cpu`) and an independent, native second opinion next to the torch
 * oracle.  What it is not: a fallback.  libprg_hip.so never calls it, pointreggpt_amd never routes to it on its own, and the
 * GPU entry points keep failing loudly when there is no HIP device.
 *
 * Same conventions as prg.h (extern "C", status codes, the caller owns every tensor), except that every pointer is a
 * HOST pointer, calls are synchronous, and there is no stream argument.  Arithmetic: float32 with float64 accumulation
 * in the contractions and the normalisation statistics (like the library's fp32 parity mode), IEEE expf / erff / divide;
 * the geometry functions evaluate the reference's float expressions in the reference's order (bit-exact against the
 * fixtures, like their HIP twins).  sd / dc = the reference's successive_ddnm_diffusion.py / depth_correction.py.
 */
#ifndef PRG_CPU_H
#define PRG_CPU_H

#include "prg.h"

#ifdef __cplusplus
extern "C" {
#endif

const char* prg_cpu_last_error(void);

/* ---- geometry: twins of prg_depth2pc / prg_pc2depth / prg_project_points_zbuffer / prg_reproject_zbuffer /
 *      prg_unproject_f64 / prg_depth_augment / prg_apply_mask (same arguments without the stream) ---- */
int prg_cpu_depth2pc(const float* depth, const float* K, float* pc, uint8_t* valid, int B, int H, int W, float clip_lo,
                     float clip_hi, float invalid_value);                                              /* sd:176-209 */
int prg_cpu_pc2depth(const float* pc, const uint8_t* valid, const float* K, float* depth, uint8_t* mask, int B, int N,
                     int H, int W);                                                                    /* sd:212-265 */
int prg_cpu_project_points_zbuffer(const float* points, const int64_t* offsets, const float* pose, const float* K,
                                   float* depth, uint8_t* mask, int B, int H, int W, float depth_scale);   /* sd:2531-2552 */
int prg_cpu_reproject_zbuffer(const float* depth, const float* K, const float* pose, float* depth_out, uint8_t* mask_out,
                              int B, int H, int W, float depth_unit, float clip_lo, float clip_hi, float out_scale);  /* sd:268-286 */
int prg_cpu_unproject_f64(const float* depth, const float* K, const float* pose, double* xyz, uint8_t* valid, int B, int H,
                          int W, float depth_unit, float clip_lo, float clip_hi);                      /* sd:122-143, 2627-2628 */
int prg_cpu_depth_augment(const float* depth, float* out, int B, int H, int W);                         /* dc:577-604 */
int prg_cpu_apply_mask(const float* prob, const float* depth, const uint8_t* hit, float thr, float* depth_out,
                       uint8_t* hit_out, float* img_cond, int B, int H, int W);                        /* sd:2564-2570 */

/* ---- U-Nets: twins of prg_unet_create / _destroy / _set_time_freqs / prg_unet_forward / prg_maskunet_forward ---- */
typedef struct prg_cpu_unet prg_cpu_unet;
int prg_cpu_unet_create(const prg_unet_config* cfg, const float* weights, int64_t n_floats, prg_cpu_unet** out);
int prg_cpu_unet_destroy(prg_cpu_unet* h);
int prg_cpu_unet_set_time_freqs(prg_cpu_unet* h, const float* freqs, int n);
int prg_cpu_unet_forward(prg_cpu_unet* h, const float* x, const int64_t* time, const float* param_cond, float* out, int B,
                         int S);                                                                       /* sd:920-964 */
int prg_cpu_maskunet_forward(prg_cpu_unet* h, const float* depth, float* prob, int B, int S);           /* dc:871-906 */

/* ---- sampler: twin of prg_sampler_create + prg_sampler_run in one call (no graph to keep) ----
 * steps: the prg_step table (n_steps rows); noise: stored draws (noise_slabs, B, S*S) in the reference's order, or NULL ->
 * Philox4x32-10 keyed by seeds[B] (the device generator's counter layout; libm's logf / sincosf, so not bit-identical to
 * the device's draws).  img_cond (B,2,S,S) or NULL.  out (B,1,S,S) in [0,1] (or beyond: ddim_sample does not clamp the
 * replaced pixels).  sd:1283-1409.                                                                                      */
int prg_cpu_sampler_run(prg_cpu_unet* h, const prg_step* steps, int n_steps, const float* param_cond, const float* img_cond,
                        const float* noise, int64_t noise_slabs, const uint64_t* seeds, float* out, int B, int S);

#ifdef __cplusplus
}
#endif
#endif
