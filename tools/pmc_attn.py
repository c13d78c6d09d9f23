#!/usr/bin/env python3
"""Torch-free driver for PMC runs (rocprofv3 --pmc crashes inside the profiler when PyTorch is loaded on
this image).  Builds a 2-layer engine at mini_v0.1 widths through the C ABI with numpy + libamdhip64 only,
runs a bf16 batch-32 greedy decode to S = max_length, so that the self-attention kernel's HBM traffic
(FETCH_SIZE) can be compared with its algorithmic bytes 2*B*S*D*2 per launch.

    rocprofv3 --pmc TCC_EA0_RDREQ_sum --output-format csv -d out -- python3 tools/pmc_attn.py [L] [layers] [P]
`900 1` (one layer, S <= 899) is the configuration recorded as completing under --pmc; larger ones (`1400 2`, `2580 24 16` = the
bench workload's decode loop) are recorded as aborting inside librocprofiler-sdk in most passes, with or without
PTTS_NO_GRAPH=1 (profiles/r02_rocprofv3_pmc_abort_symbolized.txt, profiles/r03_pmc_abort_root_cause.txt).  A long context
with FEW launches: a long prompt, e.g. `580 1 2000` (S = 2001 .. 2579 over 579 steps; the prompt positions go through the
batched prefill).   PTTS_PMC_BATCH=n: utterances (default 32).   PTTS_DUMP_MAPS=<file>: /proc/self/maps written AFTER the first
decode step (when the HSA queues of the launch path exist), PTTS_DUMP_ALLOCS=<file>: the engine's own device ranges.
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from parler_tts_amd import _hip  # noqa: E402  (ctypes structs only; no torch import)

hip = C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
hip.hipDeviceSynchronize.argtypes = []


def dev(arr):
    arr = np.ascontiguousarray(arr)
    p = C.c_void_p()
    assert hip.hipMalloc(C.byref(p), max(arr.nbytes, 16)) == 0
    assert hip.hipMemcpy(p, arr.ctypes.data_as(C.c_void_p), arr.nbytes, 1) == 0
    return p


def main():
    L = int(sys.argv[1]) if len(sys.argv) > 1 else 1400
    lib = _hip.load_library()
    NL = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    P = int(sys.argv[3]) if len(sys.argv) > 3 else 0          # prompt positions (bench: 16)
    B, T, D, H, F, V, K, ENC = int(os.environ.get("PTTS_PMC_BATCH", "32")), 40, 1024, 16, 4096, 1088, 9, 768
    c = _hip.PttsConfig()
    c.abi_version, c.dtype = _hip.ABI_VERSION, _hip.DTYPE_BF16
    c.hidden, c.layers, c.heads, c.ffn, c.vocab, c.codebooks, c.max_positions = D, NL, H, F, V, K, 4096
    c.pad_token_id, c.eos_token_id, c.bos_token_id = 1024, 1024, 1025
    c.prompt_vocab, c.enc_hidden, c.t5_layers = 64, ENC, 0
    c.max_batch, c.max_desc_len, c.max_prompt_len, c.max_length = B, T, P, L
    h = C.c_void_p()
    _hip.check(lib.ptts_create(C.byref(c), C.byref(h)))
    rng = np.random.default_rng(0)

    def load(name, shape, std=0.02, ones=False, zero_from=None):
        a = np.ones(shape, np.float32) if ones else (rng.standard_normal(shape, dtype=np.float32) * std)
        if zero_from is not None:
            a[zero_from:] = 0
        sh = (C.c_int64 * len(shape))(*shape)
        _hip.check(lib.ptts_load_tensor(h, name.encode(), dev(a), sh, len(shape), None))

    p = "decoder.model.decoder."
    for k in range(K):
        load(f"{p}embed_tokens.{k}.weight", (V + 1, D))
        load(f"decoder.lm_heads.{k}.weight", (V, D), zero_from=1024)
    load(f"{p}embed_positions.weights", (4096, D), std=0.5)
    load(f"{p}layer_norm.weight", (D,), ones=True)
    load(f"{p}layer_norm.bias", (D,), std=0.0)
    load("embed_prompts.weight", (64, D))
    load("enc_to_dec_proj.weight", (D, ENC))
    load("enc_to_dec_proj.bias", (D,))
    for l in range(NL):
        q = f"{p}layers.{l}."
        for a in ("self_attn", "encoder_attn"):
            for pr in ("q_proj", "k_proj", "v_proj", "out_proj"):
                load(f"{q}{a}.{pr}.weight", (D, D))
        for ln in ("self_attn_layer_norm", "encoder_attn_layer_norm", "final_layer_norm"):
            load(f"{q}{ln}.weight", (D,), ones=True)
            load(f"{q}{ln}.bias", (D,), std=0.0)
        load(f"{q}fc1.weight", (F, D))
        load(f"{q}fc2.weight", (D, F))
    hip.hipDeviceSynchronize()
    _hip.check(lib.ptts_weights_ready(h))
    enc = dev(rng.standard_normal((B, T, ENC), dtype=np.float32))
    amask = dev(np.ones((B, T), np.int32))
    pattern = np.full((B * K, L), -1, np.int64)
    pattern[:, 0] = 1025
    raw = np.full((B * K, L), 1024, np.int64)
    raw[:, 0] = 1025
    d_pat, d_raw = dev(pattern), dev(raw)
    pids = dev(rng.integers(0, 64, (B, max(P, 1))).astype(np.int64)) if P else None
    _hip.check(lib.ptts_prefill(h, B, B, T, P, L, enc, amask, pids, None, d_pat, d_raw, 1, None))
    sp = _hip.PttsSampling(0, 1.0, 0, 1.0, 0)
    _hip.check(lib.ptts_decode_steps(h, 1, C.byref(sp), 0, None))
    hip.hipDeviceSynchronize()
    if os.environ.get("PTTS_DUMP_MAPS"):      # address map of this process, to symbolise a profiler abort afterwards
        with open("/proc/self/maps") as src, open(os.environ["PTTS_DUMP_MAPS"], "w") as dst:
            dst.write(src.read())
    _hip.check(lib.ptts_decode_steps(h, L - 2, C.byref(sp), 0, None))
    hip.hipDeviceSynchronize()
    st = _hip.PttsState()
    _hip.check(lib.ptts_get_state(h, C.byref(st), None))
    print(f"done: cur_len={st.cur_len} pos={st.pos} layers={NL} B={B} D={D}", flush=True)
    if os.environ.get("PTTS_STAMPS_OUT"):     # diagnostic library (make stamps; PTTS_HIP_LIB=...): in-kernel clock stamps of the LAST step
        ptr, nbytes = C.c_void_p(), C.c_int64()
        _hip.check(lib.ptts_debug_buffer(h, b"stamps", C.byref(ptr), C.byref(nbytes)))
        host = np.empty(nbytes.value // 8, np.uint64)
        assert hip.hipMemcpy(host.ctypes.data_as(C.c_void_p), ptr, nbytes.value, 2) == 0
        host = host.reshape(-1, 1024, 8)
        used = int((host[:, :, 0] != 0).any(axis=1).sum())
        np.save(os.environ["PTTS_STAMPS_OUT"], host[:used])
        print(f"stamps: {used} launch slots saved", flush=True)
    lib.ptts_destroy(h)


if __name__ == "__main__":
    main()
