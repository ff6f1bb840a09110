// sampler.h — the per-transition elementwise update of the DDNM / DDIM samplers and the on-device noise.
#pragma once
#include "common.h"

namespace prg {

struct SamplerStepArgs {
  float* x;                 // state (B, HW), updated in place
  const float* u;           // U-Net output (B, HW)
  const float* cond;        // (B, 2, HW) in [-1,1] or null
  const float* noise;       // stored draws (n_steps + 1, B, HW) or null -> Philox
  const prg_step* steps;    // device copy of the transition table
  int* step_idx;            // device step counter: read by every workgroup, advanced by the last one to finish
  int* ticket;              // arrival counter of the workgroups of one launch (self-resetting)
  const uint64_t* seeds;    // device (B,) Philox keys (used when noise == null)
  float* final_out;         // (B, HW): written on the last transition as (x + 1) / 2
  int B, HW, n_steps;
};

int launch_sampler_step(const SamplerStepArgs& a, hipStream_t s);
// x = start image: noise slab 0, or Philox draw 0
int launch_sampler_init(float* x, const float* noise, const uint64_t* seeds, int B, int HW, hipStream_t s);

}  // namespace prg
