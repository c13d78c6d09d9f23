"""ctypes binding of the in-tree C-ABI library (include/parler_tts_hip.h).

Fails loudly when the library is missing: there is no CPU / PyTorch fallback for the hot path.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libparler_tts_hip.so")

ABI_VERSION = 8
DTYPE_F32, DTYPE_BF16 = 0, 1
STATE_SLOTS = 4        # PTTS_STATE_SLOTS

# every symbol include/parler_tts_hip.h declares (tests check the export table against this list)
EXPORTS = [
    "ptts_last_error", "ptts_abi_version", "ptts_create", "ptts_destroy", "ptts_load_tensor", "ptts_weights_ready",
    "ptts_t5_encode", "ptts_dac_decode", "ptts_dac_encode", "ptts_set_prompt_embeds", "ptts_set_decoder_mask", "ptts_prefill", "ptts_decode_steps", "ptts_forward_step", "ptts_select_step", "ptts_get_state", "ptts_state_snapshot", "ptts_state_poll", "ptts_last_logits",
    "ptts_set_block_table", "ptts_pool_info", "ptts_set_forced_tokens", "ptts_set_forced_sequence", "ptts_profile_read", "ptts_test_gemm",
    "ptts_debug_stop_after", "ptts_debug_buffer", "ptts_dac_kernel_counts",
]

# kernel families of ptts_dac_kernel_counts (include/parler_tts_hip.h)
DAC_KERNEL_KINDS = ["last", "stream96", "stream192", "fused96", "fused192", "s1", "ring96_t7", "ring96_t2", "ring96_t1",
                    "ring192_t7", "ring192_t2", "ring192_t1", "h3", "f32"]


class PttsConfig(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32), ("dtype", C.c_int32),
        ("hidden", C.c_int32), ("layers", C.c_int32), ("heads", C.c_int32), ("ffn", C.c_int32), ("vocab", C.c_int32),
        ("codebooks", C.c_int32), ("max_positions", C.c_int32),
        ("pad_token_id", C.c_int32), ("eos_token_id", C.c_int32), ("bos_token_id", C.c_int32),
        ("prompt_vocab", C.c_int32), ("enc_hidden", C.c_int32),
        ("t5_layers", C.c_int32), ("t5_heads", C.c_int32), ("t5_dkv", C.c_int32), ("t5_dff", C.c_int32),
        ("t5_vocab", C.c_int32), ("t5_gated", C.c_int32), ("t5_act", C.c_int32), ("t5_eps", C.c_float),
        ("max_batch", C.c_int32), ("max_desc_len", C.c_int32), ("max_prompt_len", C.c_int32), ("max_length", C.c_int32),
        ("dac_codebook_size", C.c_int32), ("dac_codebook_dim", C.c_int32), ("dac_latent", C.c_int32),
        ("dac_hidden", C.c_int32), ("dac_n_up", C.c_int32), ("dac_up", C.c_int32 * 8),
        ("dac_enc_hidden", C.c_int32),
        ("dec_variant", C.c_int32),
    ]


class PttsSampling(C.Structure):
    _fields_ = [("do_sample", C.c_int32), ("temperature", C.c_float), ("top_k", C.c_int32),
                ("guidance_scale", C.c_float), ("seed", C.c_uint64), ("top_p", C.c_float),
                ("eos_token_id", C.c_int32), ("pad_token_id", C.c_int32), ("eos_free_until", C.c_int32)]

    def __init__(self, do_sample=0, temperature=1.0, top_k=0, guidance_scale=1.0, seed=0, top_p=1.0, eos_token_id=-1,
                 pad_token_id=-1, eos_free_until=0):
        super().__init__(do_sample, temperature, top_k, guidance_scale, seed, top_p, eos_token_id, pad_token_id, eos_free_until)


class PttsState(C.Structure):
    _fields_ = [("pos", C.c_int32), ("cur_len", C.c_int32), ("prompt_len", C.c_int32), ("n_unfinished", C.c_int32),
                ("done_len", C.c_int32), ("step_count", C.c_int32)]


class PttsProfile(C.Structure):
    _fields_ = [("launches", C.c_int64), ("total_ms", C.c_double), ("algo_bytes", C.c_double), ("steps", C.c_int64),
                ("step_ms", C.c_double), ("fused_launches", C.c_int64)]


class HipLibraryMissing(RuntimeError):
    pass


_lib: Optional[C.CDLL] = None


def load_library() -> C.CDLL:
    # PTTS_HIP_LIB: alternative build of the same ABI (A/B experiments)
    """dlopen the engine; raises HipLibraryMissing with build instructions when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("PTTS_HIP_LIB", LIB_PATH)
    if not os.path.exists(path):
        raise HipLibraryMissing(
            f"{path} not found. The Parler-TTS hot path only exists as HIP code for gfx950; build it with "
            f"`python -c 'import __graft_entry__ as g; g.build()'` (or `make -C parler_tts_amd/csrc`). "
            f"There is no CPU or PyTorch fallback."
        )
    lib = C.CDLL(path)
    vp, i32, i64p = C.c_void_p, C.c_int32, C.POINTER(C.c_int64)
    lib.ptts_last_error.restype = C.c_char_p
    lib.ptts_abi_version.restype = i32
    lib.ptts_create.argtypes = [C.POINTER(PttsConfig), C.POINTER(vp)]
    lib.ptts_destroy.argtypes = [vp]
    lib.ptts_load_tensor.argtypes = [vp, C.c_char_p, vp, i64p, i32, vp]
    lib.ptts_weights_ready.argtypes = [vp]
    lib.ptts_t5_encode.argtypes = [vp, i32, i32, vp, vp, vp, vp, vp]
    lib.ptts_dac_decode.argtypes = [vp, i32, i32, vp, vp, vp]
    lib.ptts_dac_encode.argtypes = [vp, i32, i32, vp, vp, vp]
    lib.ptts_set_prompt_embeds.argtypes = [vp, vp]
    lib.ptts_set_decoder_mask.argtypes = [vp, vp]
    lib.ptts_prefill.argtypes = [vp, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, i32, vp]
    lib.ptts_decode_steps.argtypes = [vp, i32, C.POINTER(PttsSampling), i32, vp]
    lib.ptts_forward_step.argtypes = [vp, vp]
    lib.ptts_select_step.argtypes = [vp, C.POINTER(PttsSampling), vp]
    lib.ptts_get_state.argtypes = [vp, C.POINTER(PttsState), vp]
    lib.ptts_state_snapshot.argtypes = [vp, i32, vp]
    lib.ptts_state_poll.argtypes = [vp, i32, C.POINTER(PttsState)]
    lib.ptts_last_logits.argtypes = [vp, C.POINTER(vp), C.POINTER(vp)]
    lib.ptts_set_block_table.argtypes = [vp, C.POINTER(C.c_int32), i32, i32]
    lib.ptts_pool_info.argtypes = [vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int64)]
    lib.ptts_set_forced_tokens.argtypes = [vp, i64p, i32]
    lib.ptts_set_forced_sequence.argtypes = [vp, vp, vp]
    lib.ptts_profile_read.argtypes = [vp, C.POINTER(PttsProfile), i32]
    lib.ptts_debug_stop_after.argtypes = [vp, i32]
    lib.ptts_debug_buffer.argtypes = [vp, C.c_char_p, C.POINTER(vp), C.POINTER(C.c_int64)]
    lib.ptts_dac_kernel_counts.argtypes = [i64p, i32, i32]
    lib.ptts_test_gemm.argtypes = [i32, i32, i32, i32, vp, vp, i32, vp, vp, C.c_float, i32, vp, vp, vp]
    for name in EXPORTS:
        if name not in ("ptts_last_error",):
            getattr(lib, name).restype = i32
    lib.ptts_last_error.restype = C.c_char_p
    if lib.ptts_abi_version() != ABI_VERSION:
        raise RuntimeError(f"{LIB_PATH}: ABI version {lib.ptts_abi_version()} != binding {ABI_VERSION}; rebuild")
    _lib = lib
    return lib


def dac_kernel_counts(reset: bool = False) -> dict:
    """launches per codec conv kernel family since the last reset (test hook)"""
    lib = load_library()
    buf = (C.c_int64 * len(DAC_KERNEL_KINDS))()
    check(lib.ptts_dac_kernel_counts(buf, len(DAC_KERNEL_KINDS), 1 if reset else 0))
    return {k: int(buf[i]) for i, k in enumerate(DAC_KERNEL_KINDS)}


class PttsError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"parler_tts_hip error {code}: {msg}")
        self.code = code


def check(rc: int) -> None:
    if rc != 0:
        raise PttsError(rc, load_library().ptts_last_error().decode("utf-8", "replace"))
