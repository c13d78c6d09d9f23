"""MI355X-native Parler-TTS generate() engine (drop-in for the reference's Python API surface).

Exports mirror /root/reference/parler_tts/__init__.py:4-13.  The model classes need the in-tree HIP
library (``parler_tts_amd/csrc/libparler_tts_hip.so``, built by ``__graft_entry__.build()``); configs,
the delay-pattern functions and the weight utilities are host-only.
"""
__version__ = "0.1"

from .configuration import (  # noqa: F401
    DACConfig,
    GenerationConfig,
    ParlerTTSConfig,
    ParlerTTSDecoderConfig,
    T5EncoderConfig,
    mini_v01_config,
    scaled_config,
    tiny_config,
)
from .delay_pattern import apply_delay_pattern_mask, build_delay_pattern_mask  # noqa: F401


def __getattr__(name):
    # model classes are imported lazily: they load the HIP library and fail loudly without it
    if name in ("ParlerTTSForConditionalGeneration", "ParlerTTSForCausalLM", "DACModel"):
        from . import modeling

        return getattr(modeling, name)
    if name == "AudioStreamer":
        from .streamer import AudioStreamer

        return AudioStreamer
    raise AttributeError(name)
