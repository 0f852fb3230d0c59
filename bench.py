#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric on MI355X: audio-sec/s (+ decoder tok/s), Whisper-large-v2
geometry, 30 s chunks, batch 8 per GPU (BASELINE.json configs[3]), chunk-parallel over N GPUs.

One "step" = one pass of the hot path over one batch of synthetic 16 kHz audio that is
already resident in HBM (int16 PCM): log-mel front end -> encoder -> cross-attention K/V
-> KV-cached greedy decode of 224 tokens (n_text_ctx // 2, EOT suppressed so the work is
fixed) -> tokens on the host.  Steps are independent batches; the engine batches them continuously:
--fuse F (default 16) consecutive batches are decoded as ONE group of up to F*8 = 128 chunks (the decoder weights are
streamed once per group and position), and --inflight S (default 3) groups are kept in flight per GPU on S
weight-sharing contexts (one group's encoder overlaps the others' decode).  The latency of a single batch of 8
is reported next to the pipelined throughput.  Everything runs through the C ABI of libwhisper_mi355x.so
(no torch compute; torch is only used for torch.distributed / RCCL at N > 1).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--model large-v2] [--batch 8]

N > 1: one rank per GPU -- started by torch.distributed.run (the driver), or by bench.py itself when `--gpus N` is given
without a launcher (it re-executes under torch.distributed.run); a launcher whose WORLD_SIZE differs from --gpus is an
error.  Each rank owns its own chunks (weak scaling, no data-path collective) and the token streams are all-gathered over
RCCL inside the timed region.  Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
MFMA_BF16_PEAK_TF = 2500.0  # MI355X_MICROARCH.md: ~2.5 PF dense bf16


def synth_pcm16(chunk_idx):
    """BASELINE.md synthetic audio: rng(1234+idx), clip(0.1 N(0,1)), int16 = round(x * 32767)."""
    rng = np.random.default_rng(1234 + chunk_idx)
    x = np.clip(0.1 * rng.standard_normal(480000), -1.0, 1.0).astype(np.float32)
    return np.round(x * 32767).astype(np.int16)


def structured_pcm16():
    n = np.arange(480000, dtype=np.float64)
    x = 0.3 * np.cos(2 * np.pi * 1000 * n / 16000) + 0.2 * np.cos(2 * np.pi * 2400 * n / 16000) \
        + 0.1 * np.cos(2 * np.pi * 5600 * n / 16000)
    return np.round(x * 32767).astype(np.int16)


def algorithmic_work(family, dims, B):
    """Algorithmic bytes / flops of ONE launch of a kernel family (SURVEY.md 8d formulas)."""
    d, L, H, V = dims["n_text_state"], dims["n_text_layer"], dims["n_text_head"], dims["n_vocab"]
    M = B * 1500
    gemv = {"dec_gemv_ln_qkv": 3 * d * d, "dec_gemv_attn_out": d * d, "dec_gemv_ln_q": d * d,
            "dec_gemv_ln_fc1": 4 * d * d, "dec_gemv_fc2": 4 * d * d, "dec_gemv_ln_logits": V * d}
    if family in gemv:
        return "hbm", 2.0 * gemv[family]                      # bf16 weights streamed once
    if family == "dec_attn_cross":
        return "hbm", B * 2.0 * 1500 * d * 2                  # cached K and V of one layer
    if family == "dec_attn_cross_fq":
        return "hbm", B * 2.0 * 1500 * d * 2 + 2.0 * d * d    # ... + the query projection's weights (fused launch, 5 .. 12 chunks)
    if family == "dec_attn_self":
        return "hbm", B * 2.0 * 112 * d * 2                   # mean context ~ 224/2 positions
    flops = {"gemm_qkv_enc": 2.0 * M * 3 * d * d, "gemm_gelu_bf16": 2.0 * M * 4 * d * d,
             "gemm_resid_f32": 2.0 * M * d * d * 2.5,         # mean of out-proj (d*d) and fc2 (4d*d)
             "gemm_xkv": 2.0 * M * 2 * d * d, "gemm_conv2_f32": 2.0 * M * 3 * d * d,
             "enc_attention": 4.0 * B * 1500 * 1500 * d}
    if family in flops:
        return "mfma", flops[family]
    return None, 0.0


def family_source_sha256(family):
    """sha256 of the kernel source file that implements a kernel family: profiles/pmc_traffic.json stores it with every
    PMC figure, so that a figure taken on another version of the kernel is reported as null instead of going stale."""
    import hashlib
    f = ("dec_kernels.hip" if family.startswith(("dec_", "argmax")) else "gemm.hip" if family.startswith("gemm_")
         else "frontend.hip" if family.startswith("logmel") else "enc_kernels.hip")
    try:
        return f, hashlib.sha256(open(os.path.join(ROOT, "openai-whisper-coreml_amd", "csrc", f), "rb").read()).hexdigest()
    except OSError:
        return f, None


def _stage_work(dims, nb, n_prompt, max_new, mean_grp_steps):
    """Algorithmic work of ONE step (= nb chunks) per stage (SURVEY.md 8d formulas): front-end bytes, encoder + cross-K/V
    flops, decode bytes (the decoder weights are streamed once per decode group of mean_grp_steps steps and position)."""
    d, L, V = dims["n_text_state"], dims["n_text_layer"], dims["n_vocab"]
    da, La, nm = dims["n_audio_state"], dims["n_audio_layer"], dims["n_mels"]
    dec_steps = n_prompt + max_new - 1
    fe_bytes = nb * (480000 * 2 + nm * 3000 * 4)
    enc_flops = nb * (2.0 * 3000 * da * nm * 3 + 2.0 * 1500 * da * da * 3
                      + La * (8.0 * 1500 * da * da + 4.0 * 1500 * 1500 * da + 16.0 * 1500 * da * da))
    xkv_flops = nb * L * 4.0 * 1500 * d * d
    t_mean = (n_prompt + max_new) / 2.0
    dec_bytes = dec_steps * (2.0 * (L * 14 * d * d + V * d) / mean_grp_steps + nb * L * 2 * 1500 * d * 2 + nb * L * 2 * t_mean * d * 2)
    return fe_bytes, enc_flops + xkv_flops, dec_bytes, dec_steps


def stage_rooflines(dims, nb, n_prompt, max_new, mean_grp_steps, stage_s):
    """Per-stage algorithmic work (per GPU per step) against the roofline that bounds the stage: front end and decode =
    HBM bytes, encoder + cross-K/V = bf16 MFMA flops.  stage_s: seconds per step of the three stages."""
    fe_bytes, flops, dec_bytes, dec_steps = _stage_work(dims, nb, n_prompt, max_new, mean_grp_steps)
    out = {}
    if stage_s[0] > 0:
        a = fe_bytes / stage_s[0] / 1e9
        out["frontend"] = {"bound": "hbm", "achieved": a, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": a / HBM_PEAK_GBS}
    if stage_s[1] > 0:
        a = flops / stage_s[1] / 1e12
        out["encoder_xkv"] = {"bound": "mfma", "achieved": a, "peak": MFMA_BF16_PEAK_TF, "unit": "TFLOP/s",
                              "frac": a / MFMA_BF16_PEAK_TF}
    if stage_s[2] > 0:
        a = dec_bytes / stage_s[2] / 1e9
        out["decode"] = {"bound": "hbm", "achieved": a, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": a / HBM_PEAK_GBS,
                         "bytes_per_position": dec_bytes / dec_steps}
    return out


def step_roofline(dims, nb, n_prompt, max_new, mean_grp_steps, step_s):
    """Whole step against the sum of its stages' rooflines: HBM time of the front end and of the decode bytes at
    8 TB/s plus MFMA time of the encoder + cross-K/V flops at 2.5 PFLOP/s, over the measured wall time per step."""
    fe_bytes, flops, dec_bytes, _ = _stage_work(dims, nb, n_prompt, max_new, mean_grp_steps)
    bound_s = (fe_bytes + dec_bytes) / (HBM_PEAK_GBS * 1e9) + flops / (MFMA_BF16_PEAK_TF * 1e12)
    return {"bound_ms": bound_s * 1e3, "measured_ms": step_s * 1e3, "frac": bound_s / step_s,
            "hbm_bytes": fe_bytes + dec_bytes, "mfma_flops": flops}


def other_configs(B, main_model, pcm16, max_new=224):
    """The other BASELINE.json configurations on this GPU, a few seconds each: tiny.en single chunk
    (configs[1], latency), base batch 32 as one decode group (configs[2]), large-v3 15 chunks as one group (the per-GPU
    shard of configs[4]: 1 h = 120 chunks over 8 GPUs; `--model large-v3 --total-chunks 120 --gpus 8` is that job itself).
    Same weights recipe, same fixed-length greedy decode; every entry: audio-s/s of the call, its stage split (HIP events
    inside the library) and the stage rooflines."""
    out = {}
    # (key, model, chunks, repetitions, lanes): lanes 1 = ONE decode group on one lane; lanes 0 = the product's own policy
    # (round 5, measured: one group below 32 chunks, two groups up to 143, three from 144; rounds 1-4 cut 15 chunks into 8 + 7
    # on two lanes, which is what lanes = 3 -- an explicit lane count -- still does: reported beside it)
    cases = [("tiny.en_b1_latency", "tiny.en", 1, 3, 1), ("base_b32_one_group", "base", 32, 2, 1),
             ("base_b32_product_policy", "base", 32, 2, 0),
             ("large-v3_15_chunks_one_group", "large-v3", 15, 2, 1), ("large-v3_15_chunks_product_policy", "large-v3", 15, 2, 0),
             ("large-v3_15_chunks_three_lanes", "large-v3", 15, 2, 3)]
    ctx_cache = {}
    for key, model, nb, reps, lanes in cases:
        try:
            dims = B.MODEL_DIMS[model]
            c = ctx_cache.get(model)
            if c is None:
                for o in ctx_cache.values():
                    o.close()
                ctx_cache.clear()
                c = B.Context(dims)
                # the lively recipe of THIS width (weights.lively_gain: 6 below d = 1024, 4 from there): with the
                # d = 1280 gain, base decoded 32 distinct recordings to 4 distinct rows and the cross-checks below were blind
                c.init_synthetic(20240928, matrix_gain=__import__("importlib").import_module(
                    "openai_whisper_coreml_amd.weights").lively_gain(dims))
                c.finalize()
                ctx_cache[model] = c
            c.set_lanes(lanes)
            prompt = [50258, 50259, 50359, 50363] if dims["n_vocab"] >= 51865 else [50257, 50362]
            dp = c.to_device(pcm16[np.arange(nb) % len(pcm16)])
            best, stage = None, None
            for i in range(reps + 1):                      # first call: graph capture, allocations
                t0 = time.perf_counter()
                toks, _ = c.transcribe_greedy(dp, prompt, max_new, eot=-1, mem=B.WM_MEM_DEVICE, pcm_dtype=B.WM_I16, B=nb)
                dt = time.perf_counter() - t0
                if i > 0 and (best is None or dt < best):
                    best, stage = dt, np.array(c.last_stage_ms(), dtype=np.float64) / 1e3
            out[key] = {"model": model, "chunks": nb, "value": 30.0 * nb / best, "unit": "audio-sec/s",
                        "ms_per_step": best * 1e3,
                        "step_roofline": step_roofline(dims, nb, len(prompt), max_new, 1.0, best),
                        "distinct_token_rows": len({r.tobytes() for r in toks}),
                        # the chunks are nb DIFFERENT recordings: (nearly) nb distinct rows, or the token checks are blind (a random-init
                        # model may drive a few recordings into the same token cycle: >= 90 % distinct is the bar)
                        "rows_mostly_distinct": bool(len({r.tobytes() for r in toks}) >= 0.9 * min(nb, len(pcm16))),
                        "timing": "min of %d calls after one warm-up call, %s" % (
                            reps, "one decode group on one lane" if lanes == 1 else
                            "the product's own group / lane policy (wm_transcribe_greedy default)" if lanes == 0 else
                            "wm_set_lanes(%d): groups of ~8 chunks on weight-sharing lanes (the rounds-1-4 rule)" % lanes)}
            if lanes == 1:   # per-stage figures only mean something for ONE group (lanes overlap their stages)
                out[key].update({
                    "decode_stage_tok_per_s": nb * max_new / max(stage[2], 1e-9),
                    "decoder_ms_per_position": stage[2] * 1e3 / (len(prompt) + max_new - 1),
                    "stage_ms": {"frontend": stage[0] * 1e3, "encoder_xkv": stage[1] * 1e3, "decode": stage[2] * 1e3},
                    "stage_roofline": stage_rooflines(dims, nb, len(prompt), max_new, 1.0, stage)})
            else:
                ref_t = out.get("_tokens_" + model)
                out[key]["tokens_equal_one_group_run"] = None if ref_t is None else bool(np.array_equal(toks, ref_t))
            if lanes == 1:
                out["_tokens_" + model] = toks
            c.dev_free(dp)
        except Exception as e:   # never take the headline down
            out[key] = {"model": model, "chunks": nb, "value": None, "error": repr(e)}
    for k_ in [k_ for k_ in out if k_.startswith("_tokens_")]:
        out.pop(k_)
    for o in ctx_cache.values():
        o.close()
    for name, fn in (("small_lid_reference_flow", reference_flow_small), ("frontend_reference_abi", frontend_alone)):
        try:
            out[name] = fn(B, pcm16)
        except Exception as e:
            out[name] = {"value": None, "error": repr(e)}
    return out


def reference_flow_small(B, pcm16):
    """The ONLY flow the reference itself runs (ContentView.swift:56-63, Whisper.swift:23-40): Whisper-small, ONE 30 s chunk
    as 480 000 host doubles -> generate_spectrogram at the reference's f64 ABI (host buffers: the PCIe copies are in the
    time) -> f64 -> f32 -> encoder -> ONE decoder step on <|startoftranscript|> -> arg-max over the 99 language ids.
    Wall ms of binding.Whisper.encode + .decode (the Swift surface's names), min of 5 after one warm-up call; beside it the
    CPU oracle's time for the same flow on this box's host cores (C front end on 1 thread, torch fp32 model)."""
    import ctypes as C
    w = B.Whisper("small", synthetic_seed=20240928)
    x = pcm16[1].astype(np.float64) / 32768.0
    best, parts = None, None
    lang = None
    for i in range(6):
        t0 = time.perf_counter()
        feats = w.encode(x)
        t1 = time.perf_counter()
        lang = w.decode(feats)
        t2 = time.perf_counter()
        if i > 0 and (best is None or t2 - t0 < best):
            best, parts = t2 - t0, (t1 - t0, t2 - t1)
    dims = B.MODEL_DIMS["small"]
    res = {"model": "small", "chunks": 1, "unit": "ms", "value": best * 1e3, "higher_is_better": False,
           "encode_ms": parts[0] * 1e3, "decode_lid_ms": parts[1] * 1e3, "language": lang, "audio_s_per_s": 30.0 / best,
           "what": "host f64[480000] -> generate_spectrogram (f64 ABI, host pointers: PCIe in the time) -> f32 -> wm_encode "
                   "(host in / host out: 4.6 MB of features back over PCIe, as the CoreML call returns them) -> "
                   "wm_detect_language (features host -> device again, T = 1 step, arg-max over 99 ids); min of 5"}
    # The same flow DEVICE-RESIDENT: the 480 000 samples are in HBM already, every intermediate stays
    # there (wm_logmel f32 -> wm_encode -> wm_detect_language, all WM_MEM_DEVICE), 4 bytes come back.  Prices the two 4.6 MB
    # PCIe crossings of the feature map (and the f64 spectrogram's) that exist above only because the ABI mirrors CoreML's
    # host tensors.
    try:
        c = w.ctx
        d_x = c.to_device(x)
        d_mel = c.dev_malloc(80 * 3000 * 4)
        d_xa = c.dev_malloc(1500 * dims["n_audio_state"] * 4)
        d_lang = c.dev_malloc(4)
        best_d = None
        for i in range(6):
            t0 = time.perf_counter()
            assert c.lib.wm_logmel(c.handle, d_x, B.WM_F64, 1, 80, d_mel, B.WM_F32, B.WM_MEM_DEVICE) == 0
            assert c.lib.wm_encode(c.handle, d_mel, 1, d_xa, B.WM_MEM_DEVICE) == 0
            assert c.lib.wm_detect_language(c.handle, d_xa, 1, 50258, 50259, 50357, d_lang, B.WM_MEM_DEVICE) == 0
            idx_d = int(c.download(d_lang, (1,), np.int32)[0])
            dt_ = time.perf_counter() - t0
            if i > 0 and (best_d is None or dt_ < best_d):
                best_d = dt_
        res["device_resident_ms"] = best_d * 1e3
        res["device_resident_language_matches"] = bool(B.Whisper.LANGUAGES[idx_d] == lang)
        for p_ in (d_x, d_mel, d_xa, d_lang):
            c.dev_free(p_)
    except Exception as e:
        res["device_resident_ms"] = None
        res["device_resident_error"] = repr(e)
    try:   # the CPU oracle on the same flow (checker used as a timed baseline, never as the product)
        import torch
        from oracle import whisper_ref as R
        so = os.path.join(ROOT, "oracle", "liboracle_logmel.so")
        lib = C.CDLL(so)
        sd = {n: torch.from_numpy(w.ctx.get_tensor(n, s_)) for n, s_, _ in
              __import__("importlib").import_module("openai_whisper_coreml_amd.weights").tensor_specs(dims)}
        buf = np.zeros(480400)
        out = np.zeros(240000)

        def cpu_flow():
            buf[:] = 0.0
            buf[200:480200] = x
            lib.generate_spectrogram(buf.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
            xa = R.encode(sd, dims, out.astype(np.float32).reshape(1, 80, 3000))
            return R.detect_language(sd, dims, xa)
        t_cpu, (idx, _conf) = _best_of(cpu_flow, 2)
        res["cpu_oracle_ms"] = t_cpu * 1e3
        res["cpu_oracle_language_matches"] = bool(B.Whisper.LANGUAGES[int(np.asarray(idx).ravel()[0])] == lang)
        res["cpu_threads"] = torch.get_num_threads()
    except Exception as e:
        res["cpu_oracle_ms"] = None
        res["cpu_oracle_error"] = repr(e)
    w.ctx.close()
    return res


def frontend_alone(B, pcm16):
    """The front end ALONE at the reference ABI (stft.swift:8-19 -> `generate_spectrogram`, host f64 in, host f64 out,
    one chunk per call as the Rust crate is called): audio-s/s with ONE caller and with 8 concurrent caller threads."""
    x = pcm16[1].astype(np.float64) / 32768.0
    B.generateSpectrogram(x)                      # warm-up: context of the ABI symbol, tables
    t1, _ = _best_of(lambda: B.generateSpectrogram(x), 5)
    n_thr, per = 8, 6

    def worker(bar):
        bar.wait()
        for _ in range(per):
            B.generateSpectrogram(x)
    t8 = None
    for rep in range(3):   # first pass: the pool of front-end contexts behind the symbol is created (one per concurrent caller)
        bar = threading.Barrier(n_thr + 1)
        ths = [threading.Thread(target=worker, args=(bar,)) for _ in range(n_thr)]
        for t in ths:
            t.start()
        bar.wait()
        t0 = time.perf_counter()
        for t in ths:
            t.join()
        dt = time.perf_counter() - t0
        if rep > 0:
            t8 = dt if t8 is None else min(t8, dt)
    return {"unit": "audio-sec/s", "value": 30.0 / t1, "one_caller_ms_per_chunk": t1 * 1e3,
            "eight_callers_audio_s_per_s": 30.0 * n_thr * per / t8,
            "what": "generate_spectrogram(double*, double*) through ctypes: 3.84 MB in + 1.92 MB out over PCIe per chunk "
                    "(f64), kernel time ~0.1 ms: the call is bound by the host copies and the Python wrapper's buffers"}


def build_summary(line):
    """Compact digest of a bench line, emitted as its LAST key: the driver keeps a 2 000-character tail of
    the line, and the 25 KB of tables in front used to push every figure a reader needs out of it.  A pure function of
    the line (CPU-tested on a recorded one); <= 1 500 characters as JSON."""
    def frac(d, *path):
        for k in path:
            d = d.get(k) if isinstance(d, dict) else None
        return round(d, 4) if isinstance(d, (int, float)) else None

    def r1(v):
        return round(v, 1) if isinstance(v, (int, float)) else None
    out = {"value": r1(line.get("value")), "n_gpus": line.get("n_gpus"), "ms_per_step": r1(line.get("ms_per_step")),
           "value_batch8": r1(line.get("value_batch8")), "single_batch_latency_ms": r1(line.get("single_batch_latency_ms")),
           "early_stop_value": r1((line.get("early_stop") or {}).get("value")),
           "tok_per_s": r1(line.get("tok_per_s_end_to_end")),
           "stage_frac": {k: frac(line, "stage_roofline", k, "frac") for k in ("frontend", "encoder_xkv", "decode")},
           "step_frac": frac(line, "step_roofline", "frac"),
           "roofline": {"kernel": (line.get("roofline") or {}).get("kernel"), "frac": frac(line, "roofline", "frac"),
                        "avg_us": frac(line, "roofline", "avg_us"), "in_situ_frac": frac(line, "roofline", "in_situ", "frac"),
                        "traffic_over_alg": None},
           "token_checks": line.get("token_checks"), "distinct_token_rows": [line.get("distinct_token_rows"), line.get("token_rows")],
           "cpu_baseline": None,
           "rccl_ranks": line.get("rccl_ranks")}
    cb = line.get("cpu_baseline") or {}
    if isinstance(cb.get("value"), (int, float)):
        out["cpu_baseline"] = {"audio_s_per_s": round(cb["value"], 3), "cores": cb.get("cores")}
    roof = line.get("roofline") or {}
    if roof.get("traffic") and roof.get("alg_bytes_per_launch"):
        out["roofline"]["traffic_over_alg"] = round(roof["traffic"] / roof["alg_bytes_per_launch"], 4)
    oc = line.get("other_configs") or {}
    # one number per other configuration: audio-s/s (ms for the reference's own flow), + the step's roofline fraction
    out["other"] = {}
    for k, v in oc.items():
        if not isinstance(v, dict):
            continue
        e = [r1(v.get("value")), frac(v, "step_roofline", "frac")] if "step_roofline" in v else r1(v.get("value"))
        if "stage_roofline" in v:   # one-group entries: + the encoder stage's fraction of the MFMA peak
            e.append(frac(v, "stage_roofline", "encoder_xkv", "frac"))
        out["other"][k] = e
    flags = [v["rows_mostly_distinct"] for v in oc.values() if isinstance(v, dict) and "rows_mostly_distinct" in v]
    eq = [v["tokens_equal_one_group_run"] for v in oc.values() if isinstance(v, dict) and v.get("tokens_equal_one_group_run") is not None]
    out["other_token_checks"] = {"rows_mostly_distinct": all(flags) if flags else None, "equal_one_group_run": all(eq) if eq else None}
    if isinstance(oc.get("small_lid_reference_flow"), dict) and oc["small_lid_reference_flow"].get("device_resident_ms") is not None:
        out["other"]["small_lid_device_resident_ms"] = round(oc["small_lid_reference_flow"]["device_resident_ms"], 3)
    return out


def effective_cores():
    """Host cores this process may actually use (affinity mask and cgroup CPU quota)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = max(1, min(n, int(float(q) / float(per) + 0.5)))
    except Exception:
        pass
    return n


def _best_of(fn, reps):
    """min wall time of `reps` calls (first call may page code / weights in: it is one of the repetitions, never alone)."""
    best, out = None, None
    for _ in range(reps):
        t0 = time.perf_counter()
        out = fn()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    return best, out


def cpu_baseline(ctx, dims, pcm16_chunk, prompt):
    """CPU oracle timed on this box's host cores (a reported baseline, not the target):
    C restatement of the Rust front end (1 thread, as the crate is single-threaded) +
    PyTorch fp32 restatement of the model, same synthetic weights pulled back from HBM.
    BOUNDED sample (~15-30 s of CPU work), every figure the MINIMUM of 3 repetitions (a single timing of this leg swung
    3x between rounds): 1 chunk; front end in full; conv stem + 1 and + 3 encoder layers (per-layer = the difference);
    cross-K/V and 16 decode steps (batch 1) of the first 4 decoder layers; per-layer times are extrapolated
    to the full depth and 224 tokens and labelled so.  tiny.en (BASELINE.json configs[1]) is additionally run IN FULL
    (4 + 4 layers, 224 tokens, nothing extrapolated)."""
    import subprocess
    import importlib
    import torch
    from oracle import whisper_ref as R
    W = importlib.import_module("openai_whisper_coreml_amd.weights")
    so = os.path.join(ROOT, "oracle", "liboracle_logmel.so")
    if not os.path.exists(so):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True, capture_output=True)
    lib = ctypes.CDLL(so)
    cores = effective_cores()
    threads = min(cores, 64)
    torch.set_num_threads(threads)
    REPS = 3
    x = (pcm16_chunk.astype(np.float32) / 32768.0)[None, :]
    out = np.zeros((1, 80, 3000))
    t_fe, _ = _best_of(lambda: lib.oracle_logmel_batch_f32(x.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(1),
                                                           out.ctypes.data_as(ctypes.c_void_p)), REPS)
    # ... and chunk-parallel over all usable cores (OpenMP; SURVEY.md 8d asks for both)
    n_par = max(2, min(threads, 32))
    xs = np.ascontiguousarray(np.repeat(x, n_par, axis=0))
    outs = np.zeros((n_par, 80, 3000))
    t_fe_par = None
    if hasattr(lib, "oracle_logmel_batch_f32_omp"):
        os.environ.setdefault("OMP_NUM_THREADS", str(threads))
        t_par, _ = _best_of(lambda: lib.oracle_logmel_batch_f32_omp(xs.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(n_par),
                                                                    outs.ctypes.data_as(ctypes.c_void_p)), 2)
        t_fe_par = t_par / n_par
    nl = 4
    sub = dict(dims, n_audio_layer=nl, n_text_layer=nl)
    keep = {n: s for n, s, _ in W.tensor_specs(sub)}
    sd = {n: torch.from_numpy(ctx.get_tensor(n, s)) for n, s in keep.items()}
    mel = out.astype(np.float32)
    if dims["n_mels"] != 80:
        mel = np.zeros((1, dims["n_mels"], 3000), np.float32)
    one, three = dict(sub, n_audio_layer=1), dict(sub, n_audio_layer=3)
    t1, _ = _best_of(lambda: R.encode(sd, one, mel), REPS)
    t3, xa = _best_of(lambda: R.encode(sd, three, mel), REPS)
    per_enc = max((t3 - t1) / 2.0, 1e-9)
    t_enc = max(t1 - per_enc, 0.0) + per_enc * dims["n_audio_layer"]

    def xkv():
        for i in range(nl):
            p = f"decoder.blocks.{i}.cross_attn"
            torch.nn.functional.linear(xa, sd[p + ".key.weight"])
            torch.nn.functional.linear(xa, sd[p + ".value.weight"], sd[p + ".value.bias"])
    t_x, _ = _best_of(xkv, REPS)
    t_xkv = t_x / nl * dims["n_text_layer"]
    n_steps = 16
    t_dec, _ = _best_of(lambda: R.greedy(sd, sub, xa, prompt, n_steps, eot=-1), REPS)
    # a step = L layers + the logits GEMV; the 4-layer run over-counts the (shared) logits part,
    # so this extrapolation is slightly pessimistic for the CPU
    t_step = max(t_dec - t_x, 1e-9) / n_steps / nl * dims["n_text_layer"]
    total = t_fe + t_enc + t_xkv + 224 * t_step
    # tiny.en and base in full (SURVEY.md 8d / BASELINE.md): timing does not depend on the weight values -> N(0, 0.02^2) drawn here
    def small_model_in_full(name, reps):
        try:
            import openai_whisper_coreml_amd as pkg
            td = pkg.binding.MODEL_DIMS[name]
            g = torch.Generator().manual_seed(1)
            tsd = {}
            for n, shp, kind in W.tensor_specs(td):
                if kind == W.K_LN_W:
                    tsd[n] = torch.ones(shp)
                elif kind == W.K_LN_B:
                    tsd[n] = torch.zeros(shp)
                else:
                    tsd[n] = torch.randn(shp, generator=g) * 0.02
            tmel = out.astype(np.float32)
            tprompt = [50257, 50362] if td["n_vocab"] < 51865 else list(prompt)

            def full():
                txa = R.encode(tsd, td, tmel)
                R.greedy(tsd, td, txa, tprompt, 224, eot=-1)
            t_m, _ = _best_of(full, reps)
            return {"audio_s_per_s": 30.0 / (t_fe + t_m), "wall_s": t_fe + t_m,
                    "what": "%s, 1 chunk, front end + encoder + 224-token KV-cached greedy, in full (min of %d)" % (name, reps)}
        except Exception as e:   # never take the large-v2 figure down
            return {"audio_s_per_s": None, "what": "failed: %r" % (e,)}
    tiny = small_model_in_full("tiny.en", 2)
    base = small_model_in_full("base", 1)
    cpu_model = ""
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                cpu_model = ln.split(":", 1)[1].strip()
                break
    except Exception:
        pass
    return {"value": 30.0 / total, "unit": "audio-sec/s", "cores": threads if cores >= threads else cores,
            "kind": "port",
            "sample": "1 chunk on %d host threads (%d usable cores), every timing the min of %d repetitions: C front-end "
                      "restatement (1 thread) %.3f s; torch-fp32 conv stem + 1 and + 3 encoder layers timed, per layer "
                      "%.3f s x %d layers => encoder %.2f s; cross-K/V %.2f s; 16 decode steps of 4 decoder layers at "
                      "batch 1 timed => %.3f s/step, 224 steps extrapolated.  The Rust crate and the CoreML models "
                      "themselves cannot run here." % (threads, cores, REPS, t_fe, per_enc, dims["n_audio_layer"], t_enc,
                                                       t_xkv, t_step),
            "decoder_tok_per_s": 1.0 / t_step,
            "frontend_audio_s_per_s_1_thread": 30.0 / t_fe,
            "frontend_audio_s_per_s_all_cores": (30.0 / t_fe_par) if t_fe_par else None,
            "tiny_en_full": tiny, "base_full": base, "cpu_model": cpu_model, "nproc": os.cpu_count()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None,
                    help="world size: one rank per GPU.  Default: the launcher's WORLD_SIZE when there is one, else 1; an "
                         "explicit value that disagrees with the launcher is an error; without a launcher N > 1 starts its own ranks")
    ap.add_argument("--pin-numa", action="store_true",
                    help="pin this rank's host threads (the lane threads launch microsecond-scale graphs) to the cores of its "
                         "GPU's NUMA node (sharding.numa_cpus_for_rank); off by default")
    ap.add_argument("--steps", type=int, default=72)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--model", default="large-v2")
    ap.add_argument("--batch", type=int, default=8, help="30 s chunks per GPU per step")
    ap.add_argument("--new-tokens", type=int, default=224)
    ap.add_argument("--inflight", type=int, default=3,
                    help="independent batches kept in flight per GPU (each on its own HIP stream / context clone)")
    ap.add_argument("--fuse", type=int, default=16,
                    help="consecutive steps (batches) decoded together as ONE group of fuse*batch chunks (<= 128): the "
                         "decoder weights are streamed once per group and position")
    ap.add_argument("--weight-gain", type=float, default=None,
                    help="matrix gain of the random-init weights (wm_init_synthetic_gain).  Default: the `lively` gain of the "
                         "model's width (weights.lively_gain: 4 at d >= 1024, 6 below) -- tokens depend on the audio "
                         "and on the decode history, so that the token cross-checks below can fail; 1 = plain N(0, 0.02^2) (a "
                         "nearly input-independent model).  Timing does not depend on it.")
    ap.add_argument("--total-chunks", type=int, default=0,
                    help="STRONG scaling: the job is this many 30 s chunks in total, block-partitioned over the ranks "
                         "(sharding.partition: 120 chunks = 1 h -> 15 per GPU at 8 GPUs, BASELINE.json configs[4]); a step "
                         "is then one pass over the whole job and --batch is ignored.  Must be divisible by --gpus.")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the other BASELINE.json configurations")
    ap.add_argument("--tuning", default="",
                    help="EXPERIMENTS ONLY: key=value,... launch-shape knobs (wmdbg_set_tuning).  Loads the DEBUG library "
                         "instead of the product -- the product has no such knobs -- and marks the line \"experimental\".")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-early-stop", action="store_true", help="skip the additional early-stop workload")
    ap.add_argument("--no-single-batch", action="store_true",
                    help="skip the one-batch-in-flight latency measurement (profiling runs: keeps every decode launch of the "
                         "process at the timed group size)")
    args = ap.parse_args()

    # --gpus N decides the world size (it used to be parsed and never used).  Under a launcher
    # (torch.distributed.run sets WORLD_SIZE) an EXPLICIT --gpus must agree with it; when --gpus is not given the launcher's
    # world size is adopted (`torchrun --nproc-per-node 8 bench.py` works); without a launcher, N > 1 starts the N ranks itself.
    env_world = os.environ.get("WORLD_SIZE")
    if args.gpus is None:
        args.gpus = int(env_world) if env_world is not None else 1
    if env_world is not None and int(env_world) != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%s ranks: refusing to report a line whose "
                         "n_gpus would not be what was asked for" % (args.gpus, env_world))
    if env_world is None and args.gpus > 1:
        # --standalone: torch.distributed.run picks its own free rendezvous port (no bind-then-close race here)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
               "--nproc-per-node", str(args.gpus), os.path.abspath(__file__)] + sys.argv[1:]
        sys.stdout.flush()
        os.execv(sys.executable, cmd)   # the ranks' stdout is this process's: rank 0's JSON line comes out as before

    # One hardware queue per HIP stream (the runtime's default of 4 makes the fifth stream of the process share a queue
    # with an earlier one -- under torch.distributed the null stream and RCCL's streams come first, and two decode lanes
    # on one queue would run one after the other).  Must be in the environment before the HIP runtime initialises.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    torch = None
    collective_ranks = None
    use_dist = world > 1 or os.environ.get("WM_BENCH_FORCE_DIST") == "1"
    # WM_BENCH_DIST_BACKEND=gloo: the same N-rank code path with the collectives on the host -- lets a ONE-GPU box run two
    # ranks that share device 0 (RCCL refuses two ranks on one device); the driver's runs use the default, nccl == RCCL.
    dist_backend = os.environ.get("WM_BENCH_DIST_BACKEND", "nccl")
    dist_dev = "cuda" if dist_backend == "nccl" else None
    if use_dist:
        import torch
        import torch.distributed as dist
        if dist_backend == "nccl":
            if local_rank >= torch.cuda.device_count():
                raise SystemExit("bench.py: rank %d has no GPU (%d visible): --gpus %d needs %d devices on this node"
                                 % (rank, torch.cuda.device_count(), args.gpus, args.gpus))
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))  # nccl == RCCL on ROCm
        else:
            dist.init_process_group(backend=dist_backend)
        if dist.get_world_size() != args.gpus:
            raise SystemExit("bench.py: the process group has %d ranks, --gpus asked for %d" % (dist.get_world_size(), args.gpus))
        world = dist.get_world_size()   # from here on: what the backend actually initialised
        # the ranks the COLLECTIVE really spans: a 4-byte all-reduce of ones over the data-path backend (RCCL under nccl),
        # also at N = 1 under WM_BENCH_FORCE_DIST -- `rccl_ranks` of the line is this sum, not an echo of WORLD_SIZE
        one = torch.ones(1, dtype=torch.int32)
        one = one.cuda() if dist_backend == "nccl" else one
        dist.all_reduce(one)
        collective_ranks = int(one.item())
        if collective_ranks != world:
            raise SystemExit("bench.py: an all-reduce over the process group counted %d ranks, the group reports %d" % (collective_ranks, world))

    import importlib
    import openai_whisper_coreml_amd as pkg
    B = pkg.binding
    sharding = importlib.import_module("openai_whisper_coreml_amd.sharding")
    dims = B.MODEL_DIMS[args.model]
    if args.weight_gain is None:
        args.weight_gain = importlib.import_module("openai_whisper_coreml_amd.weights").lively_gain(dims)
    tuning = dict(kv.split("=") for kv in args.tuning.split(",") if kv)
    if tuning:
        dbg = B.load_debug_library()
        dbg.wmdbg_set_tuning.argtypes = [ctypes.c_char_p, ctypes.c_int]
        for k, v in tuning.items():
            if dbg.wmdbg_set_tuning(k.encode(), int(v)) != 0:
                raise SystemExit("unknown tuning key %r" % k)
    # WM_BENCH_LOCAL_DEVICE: device ordinal override (two ranks sharing one GPU in the one-GPU-box test)
    ctx = B.Context(dims, device=int(os.environ.get("WM_BENCH_LOCAL_DEVICE", local_rank)), debug=bool(tuning))
    ctx.init_synthetic(20240928, matrix_gain=args.weight_gain)
    ctx.finalize()

    nb = args.batch
    if args.total_chunks > 0:
        if args.total_chunks % world:
            raise SystemExit("--total-chunks %d is not divisible by the %d ranks" % (args.total_chunks, world))
        lo, hi = sharding.partition(args.total_chunks, world, rank)   # this rank's contiguous block of the recording
        nb = hi - lo
    F = max(1, min(args.fuse, 128 // nb if nb <= 128 else 1))
    S = max(1, args.inflight)
    F = min(F, max(1, -(-args.steps // S)))   # few timed steps: smaller groups rather than idle lanes
    # EVERY chunk of the run is a different recording (seeded noise; chunk 0 of rank 0 the structured KAT-3 chunk): decode
    # group g reads its own slice of the resident PCM, so a crossed row, cache or lane shows up in the token checks below
    n_res = nb * max(F, args.steps, args.warmup if args.warmup > 0 else 0)
    pcm = np.stack([structured_pcm16() if (i == 0 and rank == 0) else synth_pcm16(rank * n_res + i) for i in range(n_res)])
    d_pcm = ctx.to_device(pcm)
    chunk_bytes = 480000 * 2

    def pcm_at(first_chunk):
        return ctypes.c_void_p(d_pcm.value + first_chunk * chunk_bytes)
    sot, eot = 50258, 50257
    prompt = [sot, sot + 1, sot + 101, sot + 105] if dims["n_vocab"] >= 51865 else [50257, 50362]
    max_new = args.new_tokens

    # S independent decode groups in flight per GPU: the decode chain of ONE group is latency-bound, so
    # consecutive steps (= independent batches) are software-pipelined over S contexts that share the weights and
    # own a HIP stream + KV caches each.  The groups are formed by sharding.plan_groups -- a pure function of
    # (steps, fuse, inflight), identical on every rank -- and the token streams are exchanged ONCE per run with a
    # fixed-stride all-gather (sharding.run_grouped), so the collective never depends on which lane finished first.
    pinned = None
    if args.pin_numa:   # before the lane threads exist: they inherit the mask
        try:
            hip = ctypes.CDLL("libamdhip64.so")
            n_dev = ctypes.c_int(0)
            hip.hipGetDeviceCount(ctypes.byref(n_dev))
            bdfs = []
            for i in range(n_dev.value):
                b = ctypes.create_string_buffer(64)
                hip.hipDeviceGetPCIBusId(b, 64, i)
                bdfs.append(b.value.decode())
            pinned = sharding.pin_to_numa(bdfs, int(os.environ.get("WM_BENCH_LOCAL_DEVICE", local_rank)))
        except Exception as e:   # placement is an optimisation: never take the run down
            print("bench.py: --pin-numa failed: %r" % (e,), file=sys.stderr)
    ctxs = [ctx] + [ctx.clone() for _ in range(S - 1)]
    for c in ctxs:
        c.set_lanes(1)   # this harness supplies the concurrency itself (S host threads x one decode group each)
    gathered = None
    seen = {}   # pass tag -> {group index: tokens}: compared after the timed region
    pass_tag = ["warmup"]
    es_tokens = [0]

    t_origin = [time.perf_counter()]

    def group_budgets(g, n_chunks):
        """Synthetic early-stop workload: chunk i of decode group g stops after uniform(40..200) tokens (seeded)."""
        return np.random.default_rng(4000 + g).integers(40, 201, size=n_chunks)

    def run_steps(n_steps, collective=True, early_stop=False):
        """n_steps batches of nb chunks as decode groups, S in flight; returns summed stage ms of all groups.
        collective=False: no token all-gather (rank 0's instrumented passes, which the other ranks do not run)."""
        nonlocal gathered
        stage_sum = np.zeros(3)
        lock = threading.Lock()
        plan = sharding.plan_groups(n_steps, F, S)
        first = np.concatenate([[0], np.cumsum(plan)])[:-1] * nb   # first resident chunk of every group

        def run_group(w, g, k):
            c = ctxs[w]
            ta = time.perf_counter()
            toks, lens = c.transcribe_greedy(pcm_at(int(first[g]) % n_res), prompt, max_new, eot=-1, mem=B.WM_MEM_DEVICE,
                                             pcm_dtype=B.WM_I16, B=nb * k,
                                             budgets=group_budgets(g, nb * k) if early_stop else None)
            if early_stop:
                with lock:
                    es_tokens[0] += int(lens.sum())
                return toks, lens
            if os.environ.get("WM_BENCH_TRACE"):   # per-group host timeline (debugging lane overlap)
                print("trace lane %d group %d (%d chunks): call %.1f..%.1f ms, stages %s" % (
                    w, g, nb * k, (ta - t_origin[0]) * 1e3, (time.perf_counter() - t_origin[0]) * 1e3,
                    np.round(c.last_stage_ms(), 1)), file=sys.stderr)
            with lock:
                stage_sum[:] += c.last_stage_ms()
                seen.setdefault(pass_tag[0], {})[g] = toks
            return toks, lens

        _, g = sharding.run_grouped(plan, S, run_group, nb, max_new, dist=dist if (use_dist and collective) else None,
                                    world_size=world, device=dist_dev if use_dist else None)
        if g is not None:
            gathered = g
        return stage_sum

    def sync_all():
        if use_dist:
            dist.barrier()
            if dist_dev:
                torch.cuda.synchronize()
        for c in ctxs:
            c.sync()

    # warm-up: at least W steps, and the timed plan itself, so that every lane has captured the decode graph of the group
    # size it will run.  (Groups are whole batches: balancing them to one chunk -- 54/53/53 instead of 56/56/48 -- was
    # measured and is 0.5 % slower: 53 x 20 pairs put the cross-attention on 212 workgroups instead of 224 / 240.)
    # The line reports both: "warmup" = W as asked, "warmup_steps_run" = what actually ran before the timed region.
    warm_steps_run = max(args.warmup, args.steps) if args.warmup > 0 else 0
    run_steps(warm_steps_run)
    sync_all()
    pass_tag[0] = "timed"
    t0 = time.perf_counter()
    t_origin[0] = t0
    stage = run_steps(args.steps)
    sync_all()
    dt = time.perf_counter() - t0
    if use_dist:
        tt = torch.tensor([dt], dtype=torch.float64)
        tt = tt.cuda() if dist_dev else tt
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    # Token cross-checks (rank-local; every one of them can fail: the groups decode DIFFERENT chunks and, with the default
    # --weight-gain 4, the rows are pairwise distinct and history-dependent).  (a) here: the warm-up pass ran the same plan,
    # so group g of the warm-up and of the timed pass must agree bit for bit (graph replays, three lanes racing in a
    # different order); (b) and (c) below, after the timed region: the first batch decoded ALONE (a group of nb rows on one
    # lane) and group 0 decoded alone with eager launches must reproduce the rows they had inside the timed run.
    pass_tag[0] = "after"
    timed_tokens = seen.get("timed", {})
    token_checks = {}
    if args.steps > 0 and warm_steps_run == args.steps and args.warmup > 0:
        wu = seen.get("warmup", {})
        token_checks["warmup_pass_equals_timed_pass"] = bool(
            len(wu) == len(timed_tokens) and all(np.array_equal(wu[g], timed_tokens[g]) for g in timed_tokens))
    all_rows = np.concatenate([timed_tokens[g] for g in sorted(timed_tokens)], axis=0) if timed_tokens else np.zeros((0, max_new), np.int32)
    distinct_rows = len({r.tobytes() for r in all_rows})
    per_row = [len(set(r.tolist())) for r in all_rows]

    # Early stop: the same plan, but every chunk stops after a synthetic budget of uniform(40..200)
    # tokens instead of the forced 224 -- finished rows leave the attention walk, finished groups are not decoded further.
    # Reported beside the headline (which stays the fixed-length decode); every rank runs it (the collective is symmetric).
    early = None
    if args.steps > 0 and not args.no_early_stop:
        run_steps(args.steps, early_stop=True)      # graphs of the stop variant captured outside the timed pass
        sync_all()
        es_tokens[0] = 0
        t_e0 = time.perf_counter()
        run_steps(args.steps, early_stop=True)
        sync_all()
        dt_e = time.perf_counter() - t_e0
        if use_dist:
            te = torch.tensor([dt_e], dtype=torch.float64)
            te = te.cuda() if dist_dev else te
            dist.all_reduce(te, op=dist.ReduceOp.MAX)
            dt_e = float(te.item())
        early = {"value": 30.0 * nb * world * args.steps / dt_e, "unit": "audio-sec/s", "ms_per_step": dt_e / args.steps * 1e3,
                 "workload": "same batches and decode groups; chunk i of group g stops after uniform(40..200) tokens "
                             "(numpy default_rng(4000 + g)), no stop token",
                 "mean_tokens_per_chunk": es_tokens[0] / max(1, nb * args.steps),
                 "tok_per_s_end_to_end": es_tokens[0] * world / dt_e,
                 "speedup_vs_fixed_%d" % max_new: dt / dt_e}

    # for transparency: the same workload with ONE batch in flight (latency of a single batch of nb chunks)
    single_ms = None
    if (S > 1 or F > 1) and not args.no_single_batch:
        sync_all()
        t1 = time.perf_counter()
        for _ in range(2):
            one_t, _ = ctx.transcribe_greedy(d_pcm, prompt, max_new, eot=-1, mem=B.WM_MEM_DEVICE, pcm_dtype=B.WM_I16, B=nb)
        ctx.sync()
        single_ms = (time.perf_counter() - t1) / 2 * 1e3
        if 0 in timed_tokens:   # (b) the first batch alone == its rows inside group 0 of the timed run
            token_checks["first_batch_alone_equals_its_rows_in_group0"] = bool(np.array_equal(one_t, timed_tokens[0][:nb]))

    # Roofline of the dominant kernel, at the decode-group size the timed region actually ran: a second pass of ONE group
    # of that size (one lane, eager launches) with every launch bracketed by HIP events on the launch stream.
    # `profiles/` holds the rocprofv3 --kernel-trace summary of the same single-lane command (profiles/README.md).
    roof = None
    prof = {}
    grp_steps = max(sharding.plan_groups(args.steps, F, S)) if args.steps > 0 else 1
    mean_grp_steps = args.steps / max(1, len(sharding.plan_groups(args.steps, F, S)))   # weights stream once per group
    grp_chunks = nb * grp_steps
    if rank == 0:
        ctx.profile_reset()
        ctx.profile_enable(True)
        n_prof = 1 if grp_chunks > 16 else min(args.steps, 3)
        t_p0 = time.perf_counter()
        for _ in range(n_prof):
            prof_t, _ = ctx.transcribe_greedy(d_pcm, prompt, max_new, eot=-1, mem=B.WM_MEM_DEVICE, pcm_dtype=B.WM_I16, B=grp_chunks)
        ctx.sync()
        if 0 in timed_tokens and timed_tokens[0].shape[0] == grp_chunks:   # (c) eager, one lane == graphs, S lanes
            token_checks["group0_eager_single_lane_equals_timed_run"] = bool(np.array_equal(prof_t, timed_tokens[0]))
        prof_wall_ms = (time.perf_counter() - t_p0) * 1e3
        prof_steps = n_prof * grp_steps          # batches ("steps") the profiled pass covered: n_prof groups of grp_steps
        prof = ctx.profile()
        ctx.profile_enable(False)
        ev_over = ctx.profile_overhead_us()      # cost of the two hipEventRecord calls themselves
        traffic_db = {}
        try:
            traffic_db = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        except Exception:
            pass
        if prof:
            dom = max(prof, key=lambda k: prof[k]["ms"])
            kind, work = algorithmic_work(dom, dims, grp_chunks)
            raw_us = prof[dom]["ms"] / prof[dom]["n"] * 1e3
            avg_s = max(raw_us - ev_over, 1e-3) * 1e-6
            # HBM bytes per launch from the separate rocprofv3 --pmc FETCH_SIZE pass (profiles/), valid only for the
            # geometry it was taken on: keyed "<family>@<model>:B<group chunks>"
            t = traffic_db.get("%s@%s:B%d" % (dom, args.model, grp_chunks))
            src_file, src_sha = family_source_sha256(dom)
            fresh = bool(t) and src_sha is not None and t.get("kernel_source_sha256") == src_sha
            traffic = t["hbm_read_bytes_per_launch"] if fresh else None
            traffic_note = ("PMC pass of this build (%s)" % t.get("source", "profiles/") if fresh else
                            "null: no PMC pass for this geometry" if not t else
                            "null: the PMC pass on file was taken on another version of csrc/%s (sha256 differs)" % src_file)
            common = {"kernel": dom, "group_chunks": grp_chunks, "traffic": traffic, "traffic_note": traffic_note,
                      "avg_us": avg_s * 1e6,
                      "avg_us_events_raw": raw_us, "event_overhead_us": ev_over, "launches": prof[dom]["n"]}
            # IN SITU: the same family's mean launch duration in the TIMED configuration -- the same plan on all S lanes at
            # once (eager launches, every launch of every lane bracketed by events on its own stream).  `frac` above is the
            # kernel alone on the chip; this is the kernel next to the other groups' kernels, as the headline was timed.
            if S > 1 and args.steps > 0 and not os.environ.get("WM_BENCH_NO_INSITU"):
                for c in ctxs:
                    c.profile_reset()
                    c.profile_enable(True)
                t_i0 = time.perf_counter()
                run_steps(args.steps, collective=False)
                for c in ctxs:
                    c.sync()
                insitu_wall_ms = (time.perf_counter() - t_i0) * 1e3
                tot_ms, tot_n = 0.0, 0
                fam_all = {}
                for c in ctxs:
                    pf = c.profile()
                    c.profile_enable(False)
                    if dom in pf:
                        tot_ms += pf[dom]["ms"]
                        tot_n += pf[dom]["n"]
                    for k_, v_ in pf.items():
                        a_ = fam_all.setdefault(k_, [0.0, 0])
                        a_[0] += v_["ms"]
                        a_[1] += v_["n"]
                if tot_n:
                    us = max(tot_ms / tot_n * 1e3 - ev_over, 1e-3)
                    # mean algorithmic work per launch over the plan's groups (they differ by at most one batch)
                    plan_now = sharding.plan_groups(args.steps, F, S)
                    w_mean = sum(algorithmic_work(dom, dims, nb * k)[1] * k for k in plan_now) / max(1, sum(plan_now))
                    peak = HBM_PEAK_GBS * 1e9 if kind == "hbm" else MFMA_BF16_PEAK_TF * 1e12
                    common["in_situ"] = {"lanes": S, "avg_us": us, "launches": tot_n, "frac": w_mean / (us * 1e-6) / peak,
                                         "wall_ms": insitu_wall_ms,
                                         # every decode family's mean launch duration (event bias subtracted) in this pass
                                         "families_avg_us": {k_: max(v_[0] / v_[1] * 1e3 - ev_over, 0.0)
                                                             for k_, v_ in sorted(fam_all.items()) if k_.startswith(("dec_", "argmax")) and v_[1]},
                                         "note": "same plan, %d groups in flight, eager event-bracketed launches on every lane; a "
                                                 "launch's duration here includes the time its (short-lived) workgroups wait for "
                                                 "CUs that other groups' kernels hold, so this frac understates the stream's own "
                                                 "rate -- the decode stage's aggregate rate is stage_roofline.decode" % S}
            if kind == "hbm":
                ach = work / avg_s / 1e9
                roof = dict(common, bound="hbm", achieved=ach, peak=HBM_PEAK_GBS, unit="GB/s", frac=ach / HBM_PEAK_GBS,
                            alg_bytes_per_launch=work)
            elif kind == "mfma":
                ach = work / avg_s / 1e12
                roof = dict(common, bound="mfma", achieved=ach, peak=MFMA_BF16_PEAK_TF, unit="TFLOP/s",
                            frac=ach / MFMA_BF16_PEAK_TF, alg_flops_per_launch=work)

    others = None
    if rank == 0 and world == 1 and not args.no_other_configs and args.model == "large-v2" and args.steps > 0:
        others = other_configs(B, args.model, pcm, max_new)

    cpu = None
    if rank == 0 and not args.no_cpu_baseline:   # at N > 1 too: CPU-only, after the timed region; the other ranks wait at the closing barrier
        try:
            cpu = cpu_baseline(ctx, dims, pcm[1], prompt)
        except Exception as e:  # the baseline leg must never take the GPU number down with it
            cpu = {"value": None, "unit": "audio-sec/s", "cores": os.cpu_count(), "kind": "port",
                   "sample": "failed: %r" % (e,)}

    if rank == 0:
        total_audio = 30.0 * nb * world * args.steps
        dec_steps = len(prompt) + max_new - 1
        stage_s = stage / 1e3 / max(args.steps, 1)
        # the single-lane profiled pass covered prof_steps batches (n_prof groups of grp_steps): per-step figures divide by that
        fams = {k: {"ms_per_step": v["ms"] / prof_steps, "launches_per_step": v["n"] / prof_steps,
                    "avg_us": v["ms"] / v["n"] * 1e3} for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])}
        fam_sum = sum(f["ms_per_step"] for f in fams.values())
        line = {
            "metric": "audio-sec/s (RTF) + decoder tok/s, Whisper-large-v2 30s chunks, 1->8 GPU",
            "value": total_audio / dt,
            "unit": "audio-sec/s",
            "n_gpus": world,   # = the ranks of the initialised process group (== --gpus, checked above)
            # ranks counted by a 4-byte all-reduce over RCCL (0: the collectives ran over gloo; None: single process, no collective)
            "rccl_ranks": (collective_ranks if dist_backend == "nccl" else 0) if use_dist else None,
            "collective_ranks": collective_ranks,
            "dist_backend": (dist_backend if use_dist else None),
            "host_cores_pinned": (len(pinned) if pinned else None),   # --pin-numa: cores of this rank's GPU's NUMA node (rank 0's count)
            "steps": args.steps, "warmup": args.warmup, "warmup_steps_run": warm_steps_run,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong" if args.total_chunks > 0 else "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "whisper-%s geometry, random-init weights, batches of %d x 30 s int16 chunks resident "
                                   "in HBM, greedy %d new tokens (EOT suppressed), prompt %d tokens; a step = one batch; "
                                   "up to %d consecutive batches are decoded as one group (largest: %d chunks), %d groups in "
                                   "flight per GPU (one HIP stream + KV cache each, weights shared)"
                                   % (args.model, nb, max_new, len(prompt), F, grp_chunks, S),
                       "chunks_per_gpu": nb, "new_tokens": max_new, "decode_group_chunks": grp_chunks,
                       "decode_groups": sharding.plan_groups(args.steps, F, S),
                       "inflight_batches_per_gpu": S * F, "parallelism": "chunk-dp%d" % world,
                       "total_chunks": args.total_chunks if args.total_chunks > 0 else None},
            "rtf": dt / total_audio,
            # THE decoder figure: generated tokens of the whole job / wall time of the whole job
            "tok_per_s_end_to_end": (nb * world * max_new * args.steps) / dt,
            # a per-lane-stage construction (tokens of S groups / mean decode-stage time of a group): the rate while all S
            # lanes are in their decode stage, NOT a whole-job figure
            "decode_stage_tok_per_s_all_lanes": (S * nb * world * max_new) / max(stage_s[2], 1e-9),
            "inflight_batches_per_gpu": S * F,
            # every cross-check of token_checks holds (each one can fail: distinct chunks per group, distinct rows)
            # null = no cross-check could run with these flags (not the same thing as "inconsistent")
            "tokens_consistent_across_groups": (all(token_checks.values()) if token_checks else None),
            "token_checks": token_checks,
            "token_checks_skipped": sorted({"warmup_pass_equals_timed_pass", "first_batch_alone_equals_its_rows_in_group0",
                                            "group0_eager_single_lane_equals_timed_run"} - set(token_checks)),
            "distinct_token_rows": distinct_rows, "token_rows": int(all_rows.shape[0]),
            "distinct_tokens_per_row": {"min": int(min(per_row)) if per_row else 0,
                                        "mean": float(np.mean(per_row)) if per_row else 0.0},
            "weight_gain": args.weight_gain,
            "experimental": {"debug_library": True, "tuning": tuning} if tuning else None,
            # the literal BASELINE.json configs[3] figure: ONE batch of %d chunks in flight, nothing else on the GPU
            "value_batch8": (30.0 * nb / (single_ms * 1e-3)) if single_ms else None,
            "single_batch_latency_ms": single_ms,
            "early_stop": early,
            "decoder_ms_per_step": stage_s[2] * 1e3 / dec_steps,
            "stage_ms": {"frontend": stage_s[0] * 1e3, "encoder_xkv": stage_s[1] * 1e3, "decode": stage_s[2] * 1e3},
            "roofline": roof,
            # S pipelines overlap: per-step share of wall time
            "stage_roofline": stage_rooflines(dims, nb, len(prompt), max_new, mean_grp_steps, stage_s / S),
            "step_roofline": step_roofline(dims, nb, len(prompt), max_new, mean_grp_steps, dt / args.steps),
            "other_configs": others,
            "roofline_note": "dominant kernel family of ONE decode group of the size the timed region ran (single lane, eager launches): mean launch duration from per-launch HIP events on the launch stream minus the event-bracketing bias calibrated on a kernel of known device-clock duration; the rocprofv3 --kernel-trace summary of the same single-lane command is in profiles/; traffic = 2 x FETCH_SIZE from a separate rocprofv3 --pmc pass (profiles/pmc_traffic.json), null when no pass exists for this geometry",
            "cpu_baseline": cpu,
            # per-family table of the single-lane, eager, event-bracketed pass (one group of grp_chunks chunks = grp_steps
            # steps): sum(ms_per_step) <= kernel_families_pass.wall_ms_per_step, the pass's own wall time per step
            "kernel_families_pass": {"steps": prof_steps, "group_chunks": grp_chunks, "lanes": 1,
                                     "wall_ms_per_step": prof_wall_ms / max(prof_steps, 1),
                                     "sum_family_ms_per_step": fam_sum},
            "kernel_families": fams,
        }
        line["summary"] = build_summary(line)   # LAST key: inside the 2 000-character tail the driver keeps
        print(json.dumps(line))
    if use_dist:
        dist.barrier()   # rank 0 ran the instrumented pass / CPU baseline: leave together
        dist.destroy_process_group()
    for c in ctxs[1:]:
        c.close()
    ctx.close()


if __name__ == "__main__":
    main()
