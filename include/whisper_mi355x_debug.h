/*
 * whisper_mi355x_debug.h -- test hooks exported by libwhisper_mi355x.so.
 *
 * NOT part of the drop-in surface: these let tests/ drive individual HIP kernels (and a
 * few host-side helpers) through the same C ABI conventions, so each kernel can be
 * compared with the oracle in isolation.  All pointers are HOST pointers; the hooks
 * stage through HBM themselves.
 */
#ifndef WHISPER_MI355X_DEBUG_H
#define WHISPER_MI355X_DEBUG_H

#include "whisper_mi355x.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Host-only: slaney mel filterbank as librosa.filters.mel(sr=16000, n_fft=400, n_mels)
 * builds it (the recipe behind export_m80.py:4's mel_filters.npz); out: [n_mels][201]. */
int wmdbg_mel_filterbank(int n_mels, float *out);
/* Host-only: the embedded copy of the reference's m80.npy (80*201 f32). */
int wmdbg_mel80(float *out);

#ifdef __cplusplus
}
#endif
#endif
