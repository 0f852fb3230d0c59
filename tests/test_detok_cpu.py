"""CPU tests of the ids -> text step: wm_vocab_load / wm_detokenize against a Python restatement
of GPT-2's byte-level BPE decoding (openai-whisper's tokenizer [3p]) on a synthetic vocab.json -- neither the reference
nor this image ships a vocabulary."""
import json
import os

import numpy as np
import pytest


def bytes_to_unicode():
    """GPT-2 encoder.py bytes_to_unicode(), restated."""
    bs = list(range(ord("!"), ord("~") + 1)) + list(range(0xA1, 0xAC + 1)) + list(range(0xAE, 0xFF + 1))
    cs = bs[:]
    n = 0
    for b in range(256):
        if b not in bs:
            bs.append(b)
            cs.append(256 + n)
            n += 1
    return dict(zip(bs, [chr(c) for c in cs]))


def make_vocab(tmp_path, ensure_ascii):
    b2u = bytes_to_unicode()
    pieces = [" the", " cat", "日本", "語", " caf", "é", "\n", "\"quoted\"", "back\\slash", " \t", "😀", "", " naïve", "ë"]
    pieces += [bytes([b]) .decode("latin-1") for b in range(256)]            # every single byte as its own piece
    vocab = {}
    raw = []
    for i, p in enumerate(pieces):
        bts = p.encode("utf-8") if i < 14 else bytes([i - 14])
        key = "".join(b2u[b] for b in bts)
        if key in vocab:
            continue
        vocab[key] = len(raw)
        raw.append(bts)
    path = os.path.join(tmp_path, "vocab_%d.json" % ensure_ascii)
    with open(path, "w", encoding="utf-8") as f:
        json.dump(vocab, f, ensure_ascii=bool(ensure_ascii), indent=1 if ensure_ascii else None)
    return path, raw


@pytest.mark.parametrize("ensure_ascii", [0, 1])
def test_detokenize_matches_the_python_restatement(tmp_path, ensure_ascii):
    import openai_whisper_coreml_amd as pkg
    path, raw = make_vocab(tmp_path, ensure_ascii)      # \uXXXX escapes (incl. surrogate pairs) and raw UTF-8 both parse
    v = pkg.binding.Vocab(path)
    assert len(v) == len(raw)
    rng = np.random.default_rng(3)
    for _ in range(20):
        ids = rng.integers(0, 14, size=12)
        want = b"".join(raw[i] for i in ids).decode("utf-8", "replace")
        assert v.decode(ids) == want
    ids = [0, 1, 50257, 2, 3, 99999, -1]                 # ids without a piece: special tokens / timestamps
    assert v.decode(ids, skip_special=True) == " the cat日本語"
    assert v.decode(ids, skip_special=False) == " the cat<|50257|>日本語<|99999|><|-1|>"
    # a multi-byte character split across tokens is reassembled at byte level
    e_acute = "é".encode("utf-8")
    a, b = [raw.index(bytes([x])) for x in e_acute]
    assert v.decode([4, a, b]) == " café"
    assert v.decode([]) == ""
    v.close()


def test_vocab_errors_are_reported(tmp_path):
    import ctypes
    import openai_whisper_coreml_amd as pkg
    with pytest.raises(pkg.binding.WhisperError, match="cannot open"):
        pkg.binding.Vocab(os.path.join(tmp_path, "missing.json"))
    bad = os.path.join(tmp_path, "bad.json")
    open(bad, "w").write('{"a": 1, "b" 2}')
    with pytest.raises(pkg.binding.WhisperError, match="JSON"):
        pkg.binding.Vocab(bad)
    path, _ = make_vocab(tmp_path, 1)
    v = pkg.binding.Vocab(path)
    lib = v.lib
    ids = np.array([0, 1], np.int32)
    buf = ctypes.create_string_buffer(4)
    need = ctypes.c_size_t()
    st = lib.wm_detokenize(v.handle, ids.ctypes.data_as(ctypes.c_void_p), 2, 1, buf, 4, ctypes.byref(need))
    assert st != 0 and need.value == len(" the cat") + 1 and buf.value == b" th"     # truncated + NUL, size reported
    v.close()
