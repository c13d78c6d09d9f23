"""`import parler_tts` resolves to the MI355X engine: user code written against the reference package name
(`from parler_tts import ParlerTTSForConditionalGeneration`, README.md:36 of the reference) runs unchanged."""
import parler_tts_amd as _engine

__ptts_alias__ = True          # oracle/ref_shim.py evicts this module before it imports the real reference in the build container
__version__ = _engine.__version__
__all__ = ["ParlerTTSConfig", "ParlerTTSDecoderConfig", "DACConfig", "ParlerTTSForConditionalGeneration", "ParlerTTSForCausalLM",
           "DACModel", "apply_delay_pattern_mask", "build_delay_pattern_mask"]


def __getattr__(name):        # lazy, like parler_tts_amd itself: the model classes load the HIP library on first touch
    return getattr(_engine, name)
