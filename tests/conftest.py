import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_lib():
    """The CPU oracle's C restatement (oracle/logmel_ref.c), built on demand with gcc."""
    so = os.path.join(ROOT, "oracle", "liboracle_logmel.so")
    src = os.path.join(ROOT, "oracle", "logmel_ref.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True, capture_output=True)
    lib = ctypes.CDLL(so)
    lib.oracle_mel80.restype = ctypes.POINTER(ctypes.c_float)
    return lib


@pytest.fixture(scope="session")
def m80():
    return np.load(os.path.join(GOLDEN, "m80.npy")).reshape(80, 201)


@pytest.fixture(scope="session")
def pkg():
    """The product package; the .so is built in-tree by __graft_entry__.build()."""
    import openai_whisper_coreml_amd as p
    if not os.path.exists(p.binding.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    p.load_library()
    return p


def oracle_logmel(oracle_lib, x, filt=None):
    """Run the C oracle on one chunk (480000 samples, any float dtype) -> [n_mels][3000] f64
    plus the mutated padded buffer (stft.swift:10-11 zero pads + lib.rs:34-40 reflect)."""
    buf = np.zeros(480400, dtype=np.float64)
    buf[200:480200] = np.asarray(x, dtype=np.float64)
    if filt is None:
        out = np.zeros(80 * 3000, dtype=np.float64)
        oracle_lib.generate_spectrogram(buf.ctypes.data_as(ctypes.c_void_p),
                                        out.ctypes.data_as(ctypes.c_void_p))
        return out.reshape(80, 3000), buf
    filt = np.ascontiguousarray(filt, dtype=np.float32)
    out = np.zeros(filt.shape[0] * 3000, dtype=np.float64)
    oracle_lib.oracle_generate_spectrogram_filt(buf.ctypes.data_as(ctypes.c_void_p),
                                                out.ctypes.data_as(ctypes.c_void_p),
                                                filt.ctypes.data_as(ctypes.c_void_p),
                                                ctypes.c_int(filt.shape[0]))
    return out.reshape(filt.shape[0], 3000), buf
