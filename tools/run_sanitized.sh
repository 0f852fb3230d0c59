#!/bin/bash
# The SANITIZER LEG (README "Sanitizers"): host code of the whole library under AddressSanitizer + UBSan.
#   tools/run_sanitized.sh          CPU box: every test that reaches the library without a GPU (ABI loading, WAV reader,
#                                   de-tokenizer, token payloads, partition) + the hypothesis fuzz tests with more examples
#   tools/run_sanitized.sh gpu      GPU box: additionally the canary / malformed-weight-file / WAV-to-tokens / host-harness
#                                   GPU tests (every caller-owned output buffer, every host-pointer entry)
# Exit status: pytest's.  A sanitizer report aborts the python process (halt_on_error), i.e. fails the run.
set -e
cd "$(dirname "$0")/.."
python openai-whisper-coreml_amd/build.py --asan
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
LD_PRELOAD=$RT python -m pytest tests/test_fuzz_cpu.py tests/test_abi.py tests/test_detok_cpu.py tests/test_sharding_cpu.py -q -x -m "not gpu" -k "$SKIP"
if [ "$1" = gpu ]; then
  # (handle_abort: a stack for an abort() raised below us, e.g. by the HIP runtime)
  ASAN_OPTIONS=$ASAN_OPTIONS:handle_abort=1 LD_PRELOAD=$RT python -m pytest tests/test_canary_gpu.py tests/test_frontend_gpu.py -q -x -m gpu
fi
