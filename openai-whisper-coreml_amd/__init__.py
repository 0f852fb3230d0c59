"""MI355X-native drop-in for the hot path of tanmayb123/OpenAI-Whisper-CoreML.

Only what the path needs lives here:
  csrc/      hand-written HIP (gfx950) kernels + the C ABI (include/whisper_mi355x.h)
  build.py   hipcc recipe producing libwhisper_mi355x.so in this directory
  binding.py ctypes mirror of the reference's Swift surface (generateSpectrogram,
             Whisper.encode / Whisper.decode) on top of the C ABI
  weights.py state-dict naming, synthetic weights, flat weight file, converters
  host/      C++ host harness mirroring ContentView.swift:56-63 -> Whisper.swift:23-40
"""
from . import binding  # noqa: F401
from .binding import Whisper, generateSpectrogram, load_library  # noqa: F401
