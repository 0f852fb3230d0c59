// model.h -- Whisper encoder/decoder state (weights, activations, KV caches) and the
// launch wrappers of the HIP kernels that implement boundary #2.
#pragma once
#include "wm_internal.h"

int wm_model_create(wm_ctx *ctx, const wm_dims *dims);
void wm_model_destroy(wm_ctx *ctx);
