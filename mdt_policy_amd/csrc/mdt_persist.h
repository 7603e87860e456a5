// mdt_persist.h -- the persistent decoder kernel (mdt_persist.hip) as seen by the model-level host code.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mdt_internal.h"

struct mdt_model;
// can the step loop of a `batch`-sample mdt_sample_ddim call on this handle run as ONE persistent launch?
bool mdt_persist_supported(mdt_model* m, int64_t batch);
// enqueue it (preconditions: context encoded + folded, modulation table of the n_steps sigmas, m->steps, first embedding)
mdt_status mdt_persist_sample(mdt_model* m, int64_t batch, int n_steps, const float* x_T, float* out, hipStream_t s);
void mdt_persist_free(mdt_model* m);
