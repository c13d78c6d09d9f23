#!/usr/bin/env python3
"""Probe (round 6): what the token loop loses when its queue may only use a subset of the CUs - the precondition of running the
codec of the finished frames BESIDE the loop on the remaining CUs (`hipExtStreamCreateWithCUMask`).
   python3 tools/cu_mask_probe.py MASK [bench.py arguments ...]
MASK = "all" | "lowN" (the N lowest mask bits) | "strideK" (every bit whose index % 8 != K: one bit in eight off) | hex words
The masked stream is planted as the process's side stream (parler_tts_amd.modeling._side_streams) and bench.py runs unchanged."""
import ctypes as C
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from parler_tts_amd import modeling


def mask_words(spec: str, ncu: int = 256):
    bits = [1] * ncu
    if spec == "all":
        pass
    elif spec.startswith("low"):
        n = int(spec[3:])
        bits = [1 if i < n else 0 for i in range(ncu)]
    elif spec.startswith("stride"):
        k = int(spec[6:])
        bits = [0 if i % 8 == k else 1 for i in range(ncu)]
    elif spec.startswith("only"):                      # the complement of strideK: one CU in eight
        k = int(spec[4:])
        bits = [1 if i % 8 == k else 0 for i in range(ncu)]
    else:
        words = [int(w, 16) for w in spec.split(",")]
        return words
    return [sum(bits[32 * w + b] << b for b in range(32)) for w in range(ncu // 32)]


def main():
    spec = sys.argv[1]
    words = mask_words(spec)
    hip = C.CDLL("libamdhip64.so")
    hip.hipExtStreamCreateWithCUMask.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32)]
    hip.hipExtStreamCreateWithCUMask.restype = C.c_int
    torch.cuda.init()
    dev = torch.device("cuda:0")
    with torch.cuda.device(dev):
        torch.zeros(1, device=dev)
        h = C.c_void_p()
        arr = (C.c_uint32 * len(words))(*words)
        rc = hip.hipExtStreamCreateWithCUMask(C.byref(h), len(words), arr)
    print(f"mask {spec}: words {[hex(w) for w in words]} ({sum(bin(w).count('1') for w in words)} CUs) rc={rc}", file=sys.stderr)
    if rc != 0 or not h.value:
        sys.exit(3)
    if spec != "none" and not (len(sys.argv) > 2 and sys.argv[2] in ("--beside", "--dac")):
        modeling._side_streams[0] = torch.cuda.ExternalStream(h.value, device=dev)
    if len(sys.argv) > 2 and sys.argv[2] == "--dac":   # the codec alone (one 16-clip chunk at the benchmark length) on the masked queue
        import time
        from parler_tts_amd.configuration import mini_v01_config
        from parler_tts_amd.modeling import DACModel
        from parler_tts_amd.weights import synthetic_state_dict
        cfg = mini_v01_config()
        dac = DACModel(cfg, synthetic_state_dict(cfg, seed=1, parts=("audio_encoder",)), device="cuda:0")
        torch.manual_seed(0)
        codes = torch.randint(0, 1024, (1, 16, 9, 2571))
        st = torch.cuda.ExternalStream(h.value, device=dev)
        with torch.cuda.stream(st):
            dac.decode(codes, None)
            st.synchronize()
            t0 = time.perf_counter()
            dac.decode(codes, None)
            st.synchronize()
        print(f"mask {spec}: codec chunk {1e3 * (time.perf_counter() - t0):.1f} ms")
        return
    if len(sys.argv) > 2 and sys.argv[2] == "--beside":   # the token loop on its usual (unmasked) queue, N codec chunks queued on the masked one first
        import time
        from parler_tts_amd.configuration import mini_v01_config
        from parler_tts_amd.modeling import DACModel, ParlerTTSForConditionalGeneration
        from parler_tts_amd.weights import synthetic_state_dict
        sys.path.insert(0, ROOT)
        import bench
        n_codec = int(sys.argv[3]) if len(sys.argv) > 3 else 2
        cfg = mini_v01_config()
        sd = synthetic_state_dict(cfg, seed=1234, never_eos=True)
        model = ParlerTTSForConditionalGeneration(cfg, sd, dtype="bf16", device=dev)
        B, T, P, L = 32, 40, 16, 2580
        desc, prompt = bench.synth_inputs(cfg, B, T, P, seed=1234)
        desc, prompt = desc.to(dev), prompt.to(dev)
        amask = torch.ones(B, T, dtype=torch.long, device=dev)
        dac = model.audio_encoder
        codes = torch.randint(0, 1024, (1, 16, 9, 2571))
        kw = dict(do_sample=False, max_length=L, return_raw_ids=True)
        model.generate(input_ids=desc, attention_mask=amask, prompt_input_ids=prompt, **kw)
        torch.cuda.synchronize()
        st = torch.cuda.ExternalStream(h.value, device=dev)
        with torch.cuda.stream(st):
            dac.decode(codes, None)
        torch.cuda.synchronize()
        from parler_tts_amd import _hip
        ceng = dac._engine()
        codes_d = codes[0].to(dev).contiguous()
        wav_d = torch.empty(16, 2571 * cfg.audio_encoder.hop_length, device=dev, dtype=torch.float32)
        torch.cuda.synchronize()
        for rep in range(2):
            for beside in (0, n_codec):
                t0 = time.perf_counter()
                for _ in range(beside):      # the C entry point itself: enqueue only (DACModel.decode synchronises for its temporaries)
                    _hip.check(ceng.lib.ptts_dac_decode(ceng.handle, 16, 2571, C.c_void_p(codes_d.data_ptr()), C.c_void_p(wav_d.data_ptr()),
                                                        C.c_void_p(h.value)))
                t1 = time.perf_counter()
                # on a non-blocking stream of its own: the masked stream is a BLOCKING one (hipExtStreamCreateWithCUMask takes no flags),
                # so anything that touches the NULL stream would wait for the codec
                with torch.cuda.stream(modeling._own_stream(dev)):
                    model.generate(input_ids=desc, attention_mask=amask, prompt_input_ids=prompt, **kw)
                    torch.cuda.current_stream().synchronize()
                t2 = time.perf_counter()
                st.synchronize()
                t3 = time.perf_counter()
                print(f"mask {spec}: {beside} codec chunks beside: enqueue {1e3 * (t1 - t0):.0f} ms, token loop {1e3 * (t2 - t1):.0f} ms, "
                      f"codec queue drained {1e3 * (t3 - t2):.0f} ms later")
        return
    sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:]
    runpy.run_path(sys.argv[0], run_name="__main__")


if __name__ == "__main__":
    main()
