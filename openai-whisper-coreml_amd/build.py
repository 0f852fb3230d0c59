"""hipcc build recipe for libwhisper_mi355x.so (gfx950 only, built IN-TREE).

    python openai-whisper-coreml_amd/build.py [--force]

Each translation unit is compiled to an object next to its source (build/ is git-ignored,
the final .so too) and linked with hipcc.  Cross-compiles without a GPU.
"""
import concurrent.futures
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "libwhisper_mi355x.so")
DBG_LIB = os.path.join(HERE, "libwhisper_mi355x_dbg.so")   # product objects + csrc/debug_hooks.cpp (tests / tools only)
DEBUG_ONLY = ("debug_hooks.cpp", "f32_path.hip")   # the wmdbg_* hooks and the all-fp32 debug model path
HOST_BIN = os.path.join(HERE, "host", "lid_main")

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-fvisibility=hidden",   # only the WM_API functions of include/*.h are exported
         "-I", os.path.join(ROOT, "include")] + os.environ.get("WM_EXTRA_HIPCC_FLAGS", "").split()


# Per-file flags.  dec_kernels.hip: up to 16 SGPRs (the flag's value: an upper bound) of a decode kernel's leading SCALAR
# arguments are filled by the dispatcher -- the GEMV signatures lead with exactly 14 dwords of hot scalars (4 pointers + 6
# ints), so 14 is what gets preloaded there -- and the first weight / K-V loads of the latency-bound decode chain do not wait
# for a kernarg fetch (kernel signatures are ordered for it: see dec_gemv_kernel / dec_rows_attn_kernel).  The option is a
# hidden LLVM one: tests/test_isa_cpu.py pins its effect on this compiler and stays tolerant on others; WM_NO_KERNARG_PRELOAD=1
# builds without it (same results, ~0.3 us more per decode launch).
FILE_FLAGS = {"dec_kernels.hip": ["-mllvm", "-amdgpu-kernarg-preload-count=16"]}
if os.environ.get("WM_NO_KERNARG_PRELOAD"):   # A/B builds (tools/): same kernels, arguments fetched by s_load
    FILE_FLAGS = {}


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp")))


def _deps_mtime():
    t = 0.0
    for d in (CSRC, os.path.join(ROOT, "include")):
        for f in os.listdir(d):
            if f.endswith((".h", ".inc", ".hpp")):
                t = max(t, os.path.getmtime(os.path.join(d, f)))
    return t


def _compile(src, force, hdr_t):
    obj = os.path.join(OBJ, src + ".o")
    s = os.path.join(CSRC, src)
    if (not force and os.path.exists(obj) and os.path.getmtime(obj) >= os.path.getmtime(s)
            and os.path.getmtime(obj) >= hdr_t):
        return obj, False
    cmd = [HIPCC] + FLAGS + FILE_FLAGS.get(src, []) + ["-x", "hip", "-c", s, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return obj, True


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    hdr_t = _deps_mtime()
    srcs = sources()
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        res = list(ex.map(lambda s: _compile(s, force, hdr_t), srcs))
    objs = [o for o, _ in res]
    prod = [o for o, s_ in zip(objs, srcs) if s_ not in DEBUG_ONLY]
    changed = any(c for _, c in res) or force
    # linker version scripts: the product exports the C ABI of include/whisper_mi355x.h and nothing else (no libstdc++
    # template instantiations, no __hip_cuid_*); the debug library additionally exports the wmdbg_* hooks
    maps = {}
    for name, pats in (("product", ["generate_spectrogram", "wm_*"]), ("debug", ["generate_spectrogram", "wm_*", "wmdbg_*"])):
        maps[name] = os.path.join(OBJ, "exports_%s.map" % name)
        text = "{\n  global:\n%s  local: *;\n};\n" % "".join("    %s;\n" % p_ for p_ in pats)
        if not os.path.exists(maps[name]) or open(maps[name]).read() != text:
            with open(maps[name], "w") as f:
                f.write(text)
    for lib, members, vmap in ((LIB, prod, maps["product"]), (DBG_LIB, objs, maps["debug"])):
        if changed or not os.path.exists(lib):
            cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + members + [
                "-Wl,-rpath,/opt/rocm/lib", "-Wl,--version-script=" + vmap, "-lpthread", "-ldl"]   # RCCL: dlopen'ed by wm_multi_create, never linked
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
            if verbose:
                print("built", lib)
        elif verbose:
            print("up to date:", lib)
    return LIB


ASAN_LIB = os.path.join(HERE, "libwhisper_mi355x_asan.so")


def asan_runtime():
    """The shared AddressSanitizer runtime of the ROCm clang (to be LD_PRELOADed into the python that loads ASAN_LIB)."""
    r = subprocess.run([HIPCC, "--print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True)
    p = r.stdout.strip()
    if not os.path.isabs(p):   # hipcc wrapper: ask the clang it drives
        import glob
        c = sorted(glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so"))
        p = c[-1] if c else p
    return p


UBSAN_LIB = os.path.join(HERE, "libwhisper_mi355x_ubsan.so")


def build_sanitized(force=False, verbose=True, kind="asan"):
    """The SANITIZER LEG (README: `python openai-whisper-coreml_amd/build.py --asan`, then tools/run_sanitized.sh): the same
    sources, HOST code instrumented with AddressSanitizer + UndefinedBehaviorSanitizer (device code untouched:
    -fno-gpu-sanitize), into libwhisper_mi355x_asan.so -- product objects + the debug hooks, so that every C-ABI entry that
    parses untrusted bytes (WAV, vocab.json, weight files, token payloads) and every host-side argument check runs under
    the sanitizers, on the CPU box (front-end-free entries) and on the GPU box (everything)."""
    # kind "ubsan": UndefinedBehaviorSanitizer only -- the flavour that can run NEXT TO THE HIP RUNTIME.  (Measured on the GPU
    # box: under the AddressSanitizer runtime hipInit aborts -- the HSA runtime's address-space reservation collides with
    # ASan's shadow memory and this image ships no ASan build of ROCm -- so the GPU leg is UBSan + the guard-band canaries of
    # tests/test_canary_gpu.py, the CPU leg ASan + UBSan.)
    san = "address,undefined" if kind == "asan" else "undefined,bounds,float-cast-overflow"
    obj_dir = os.path.join(HERE, "build_" + kind)
    os.makedirs(obj_dir, exist_ok=True)
    flags = [f for f in FLAGS if f != "-O3"] + ["-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=" + san,
                                                "-fno-sanitize-recover=all", "-fno-gpu-sanitize", "-shared-libsan"]
    hdr_t = _deps_mtime()

    def comp(src):
        obj = os.path.join(obj_dir, src + ".o")
        sp = os.path.join(CSRC, src)
        if (not force and os.path.exists(obj) and os.path.getmtime(obj) >= os.path.getmtime(sp) and os.path.getmtime(obj) >= hdr_t):
            return obj
        r = subprocess.run([HIPCC] + flags + FILE_FLAGS.get(src, []) + ["-x", "hip", "-c", sp, "-o", obj], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc (sanitized) failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
        return obj
    with concurrent.futures.ThreadPoolExecutor(max_workers=8) as ex:
        objs = list(ex.map(comp, sources()))
    vmap = os.path.join(OBJ, "exports_debug.map")
    if not os.path.exists(vmap):
        build(verbose=False)
    out = ASAN_LIB if kind == "asan" else UBSAN_LIB
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs + [
        "-fsanitize=" + san, "-shared-libsan", "-Wl,-rpath,/opt/rocm/lib", "-Wl,-rpath," + os.path.dirname(asan_runtime()),
        "-Wl,--version-script=" + vmap, "-lpthread", "-ldl"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link (sanitized) failed:\n%s\n%s" % (r.stdout, r.stderr))
    if verbose:
        print("built", out, "(LD_PRELOAD=%s)" % asan_runtime() if kind == "asan" else "(no preload needed)")
    return out


def build_host(force=False, verbose=True, name="lid_main"):
    """C++ host harnesses (stand-ins for the Swift caller): lid_main = the reference's language-ID
    flow on one GPU, multi_main = all GPUs of the node from one dlopen-only process."""
    src = os.path.join(HERE, "host", name + ".cpp")
    exe = os.path.join(HERE, "host", name)
    if not os.path.exists(src):
        return None
    hdr = os.path.join(ROOT, "include", "whisper_mi355x.h")
    if (not force and os.path.exists(exe) and os.path.getmtime(exe) >= os.path.getmtime(src)
            and os.path.getmtime(exe) >= os.path.getmtime(hdr)):
        return exe
    cmd = ["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "include"), src, "-o", exe, "-ldl"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("host build failed:\n%s\n%s" % (r.stdout, r.stderr))
    if verbose:
        print("built", exe)
    return exe


if __name__ == "__main__":
    if "--asan" in sys.argv or "--ubsan" in sys.argv:
        build_sanitized(force="--force" in sys.argv, kind="asan" if "--asan" in sys.argv else "ubsan")
        sys.exit(0)
    build(force="--force" in sys.argv)
    build_host(force="--force" in sys.argv)
    build_host(force="--force" in sys.argv, name="multi_main")
