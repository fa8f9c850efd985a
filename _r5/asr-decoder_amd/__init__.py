"""asr-decoder_amd: MI355X-native batched WFST token-passing decoder.

A drop-in for ONE path of datemoon/ASR-decoder -- the frame-synchronous token-passing search of
``src/my-decoder`` over the flat HCLG of ``src/newfst`` -- behind the C ABI of
``include/wfst_decoder.h``.  The product is ``csrc/`` (HIP kernels + C ABI) and ``host/`` (the
C++ mirror of the reference's DecoderItf / DecodableInterface / Fst / Lattice); this Python
package only holds the build script, the ctypes binding used by tests and bench.py, and the
synthetic input generator.  Import with ``importlib.import_module("asr-decoder_amd")``.
"""
from . import build as _build  # noqa: F401
from . import shard, synth, wfstdec  # noqa: F401

__all__ = ["shard", "synth", "wfstdec"]
