"""ISA lint of the decode GEMV (CPU box: hipcc cross-compiles gfx950, llvm-objdump disassembles).

Round 5 found that the round-4 kernel walked FOUR dependent scalar (kernarg) round trips before its first weight load was
issued, and then -- twice in one afternoon -- that an innocent-looking edit (one cold struct field read in front of the
loads) puts such a trip back without any test noticing: the results are identical, only every one of the ~190 GEMV
launches of a decoder position is 0.3 us slower.  This test pins the property in the machine code: in every
dec_gemv_kernel instantiation, the straight-line code that issues the first weight loads contains no
scalar-memory WAIT (the leading arguments arrive preloaded in SGPRs; csrc/dec_kernels.hip "KERNEL ARGUMENTS").
DESIGN.md section 4; no reference counterpart (the reference's decoder is a CoreML graph)."""
import os
import re
import subprocess

import pytest

from conftest import ROOT

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
LLVM = "/opt/rocm/lib/llvm/bin"


@pytest.fixture(scope="module")
def gemv_kernels(tmp_path_factory):
    if not (os.path.exists(HIPCC) and os.path.exists(os.path.join(LLVM, "llvm-objdump"))):
        pytest.skip("no ROCm toolchain")
    d = tmp_path_factory.mktemp("isa")
    co, elf = str(d / "dec.co"), str(d / "dec.elf")
    src = os.path.join(ROOT, "openai-whisper-coreml_amd", "csrc", "dec_kernels.hip")
    import importlib.util
    spec = importlib.util.spec_from_file_location("_b", os.path.join(ROOT, "openai-whisper-coreml_amd", "build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    flags = [f for f in b.FLAGS if f != "-fPIC"] + b.FILE_FLAGS.get("dec_kernels.hip", [])
    subprocess.run([HIPCC] + flags + ["--cuda-device-only", "-c", "-x", "hip", src, "-o", co], check=True, capture_output=True)
    subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + co,
                    "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + elf], check=True, capture_output=True)
    out = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", elf], check=True, capture_output=True,
                         text=True).stdout
    kernels, cur = {}, None
    for ln in out.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:$", ln)
        if m:
            cur = m.group(1) if "dec_gemv_kernel" in m.group(1) else None
            if cur:
                kernels[cur] = []
            continue
        if cur and ln.startswith("\t"):
            ins = ln.split("//")[0].strip()
            if ins and not ins.startswith("s_nop"):
                kernels[cur].append(ins)
    assert len(kernels) >= 20, "no dec_gemv_kernel instantiations found in the disassembly"
    return kernels


KNOWN_COMPILER = "7.2."   # the HIP version whose prologue shape is pinned below (hipcc --version: "HIP version: 7.2.x")


def _hip_version():
    try:
        out = subprocess.run([HIPCC, "--version"], capture_output=True, text=True).stdout
        m = re.search(r"HIP version:\s*(\S+)", out)
        return m.group(1) if m else ""
    except OSError:
        return ""


def test_kernarg_preload_is_on_for_the_decode_kernels(gemv_kernels):
    """-amdgpu-kernarg-preload-count=16 (build.py FILE_FLAGS; a HIDDEN LLVM option): 16 is the upper bound of SGPRs the
    dispatcher may fill; the GEMV signatures lead with 14 dwords of hot scalars, and the kernels start with the
    backward-compatibility prologue that loads exactly those -- s_load x2 + x8 + x4, one wait.  The exact sequence is a
    property of ONE compiler: on another ROCm the check is skipped and the tolerant test below -- no scalar wait
    in front of the first weight load -- is what still holds the line; WM_NO_KERNARG_PRELOAD=1 is the documented fallback."""
    ver = _hip_version()
    if not ver.startswith(KNOWN_COMPILER):
        pytest.skip("prologue shape pinned for HIP %sx only (this is %r); the tolerant ISA test still runs" % (KNOWN_COMPILER, ver))
    for name, ins in gemv_kernels.items():
        assert ins[0].startswith("s_load_dwordx2") and ins[1].startswith("s_load_dwordx8") and ins[2].startswith("s_load_dwordx4"), (name, ins[:4])
        assert ins[3].startswith("s_waitcnt lgkmcnt(0)"), (name, ins[:4])


def test_no_scalar_round_trip_in_front_of_the_first_weight_load(gemv_kernels):
    bad = []
    for name, ins in gemv_kernels.items():
        # the weight stream: consecutive k-steps of a tile are 1 KiB apart (the L2 warm-up loop has no offsets)
        first = next((i for i, s in enumerate(ins) if s.startswith("global_load_dwordx4") and "offset:1024" in s), None)
        assert first is not None, name
        start = max((i for i in range(first) if ins[i].startswith(("s_cbranch", "s_branch"))), default=0)
        block = ins[start + 1:first]
        # (a scalar load ISSUED early is harmless; what costs the round trip is WAITING for one before the weight loads go out)
        offenders = [s for s in block if s.startswith("s_waitcnt lgkmcnt")]
        if offenders:
            bad.append((name, offenders[:3]))
    assert not bad, "a kernel-argument fetch sits in front of the first weight load again: %r" % bad[:3]
