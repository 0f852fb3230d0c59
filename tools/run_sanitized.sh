#!/bin/bash
# The SANITIZER LEG (README "Sanitizers"): host code of the whole library under AddressSanitizer + UBSan.
#   tools/run_sanitized.sh          CPU box: every test that reaches the library without a GPU (ABI loading, WAV reader,
#                                   de-tokenizer, token payloads, partition) + the hypothesis fuzz tests with more examples
#   tools/run_sanitized.sh gpu      GPU box: additionally a UBSan-only build (ASan cannot live beside the HIP runtime here) under
#                                   the canary / malformed-weight-file / WAV-to-tokens / host-harness / decode-policy GPU tests
# Exit status: pytest's.  A sanitizer report aborts the python process (halt_on_error), i.e. fails the run.
set -e
cd "$(dirname "$0")/.."
[ "$1" = gpu ] || python openai-whisper-coreml_amd/build.py --asan
RT=$(python - <<'PY'
import importlib.util
s = importlib.util.spec_from_file_location("b", "openai-whisper-coreml_amd/build.py")
m = importlib.util.module_from_spec(s); s.loader.exec_module(m); print(m.asan_runtime())
PY
)
export WM_LIB_PATH=$PWD/openai-whisper-coreml_amd/libwhisper_mi355x_asan.so WM_DBG_LIB_PATH=$PWD/openai-whisper-coreml_amd/libwhisper_mi355x_asan.so
# leaks: python itself and the HIP runtime keep memory until exit; everything else is fatal
export ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:abort_on_error=1:protect_shadow_gap=0 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1
export WM_FUZZ_EXAMPLES=${WM_FUZZ_EXAMPLES:-600}
SKIP="not exports_only_the_public_header and not environment_variables"   # properties of the PRODUCT .so file itself
if [ "$1" != gpu ]; then   # (on a GPU box `import torch` / hipInit abort under the ASan runtime: the ASan leg is the CPU box's)
  LD_PRELOAD=$RT python -m pytest tests/test_fuzz_cpu.py tests/test_abi.py tests/test_detok_cpu.py tests/test_sharding_cpu.py -q -x -m "not gpu" -k "$SKIP"
fi
if [ "$1" = gpu ]; then
  # GPU box: hipInit aborts under the AddressSanitizer runtime (shadow memory vs the HSA address-space reservation; no ASan
  # build of ROCm in this image), so the GPU leg is UBSan (signed overflow, bad shifts, misaligned / null access, float ->
  # int overflow, array bounds on every host-side size computation) + the guard-band canaries around every caller-owned buffer
  python openai-whisper-coreml_amd/build.py --ubsan
  export WM_LIB_PATH=$PWD/openai-whisper-coreml_amd/libwhisper_mi355x_ubsan.so WM_DBG_LIB_PATH=$PWD/openai-whisper-coreml_amd/libwhisper_mi355x_ubsan.so
  python -m pytest tests/test_canary_gpu.py tests/test_frontend_gpu.py tests/test_fuzz_cpu.py tests/test_model_gpu.py -q -x -m "gpu or not gpu" \
      -k "canar or malformed or guarded or generate_spectrogram or frontend or fuzz or wav or error_paths or harness or swift_surface or detect_language or timestamp or early_stop_equals or suppress"
fi
