// model_api.cpp -- boundary #2 entry points (placeholder until the model lands).
#include "model.h"
int wm_model_create(wm_ctx *, const wm_dims *) { wm_set_error("model not built yet"); return WM_ERR_STATE; }
void wm_model_destroy(wm_ctx *) {}
#define STUB { wm_set_error("model not built yet"); return WM_ERR_STATE; }
extern "C" {
int wm_set_tensor(wm_ctx *, const char *, const float *, size_t) STUB
int wm_get_tensor(wm_ctx *, const char *, float *, size_t) STUB
int wm_load_weights(wm_ctx *, const char *) STUB
int wm_init_synthetic(wm_ctx *, uint64_t) STUB
int wm_finalize(wm_ctx *) STUB
int wm_get_dims(const wm_ctx *, wm_dims *) STUB
int wm_encode(wm_ctx *, const float *, int, float *, wm_mem) STUB
int wm_decode_logits(wm_ctx *, const int32_t *, int, int, const float *, float *, wm_mem) STUB
int wm_detect_language(wm_ctx *, const float *, int, int32_t, int32_t, int32_t, int32_t *, wm_mem) STUB
int wm_transcribe_greedy(wm_ctx *, const void *, wm_dtype, int, const int32_t *, int, int, int32_t, int32_t *, int32_t *, wm_mem) STUB
}
