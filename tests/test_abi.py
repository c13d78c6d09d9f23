"""CPU-side checks of the drop-in boundary: the C-ABI library builds/loads and exports exactly what
include/parler_tts_hip.h declares; the Python API surface mirrors the reference's names."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "parler_tts_amd", "csrc", "libparler_tts_hip.so")
HEADER = os.path.join(ROOT, "include", "parler_tts_hip.h")


@pytest.fixture(scope="module")
def built_lib():
    if not os.path.exists(LIB):
        import __graft_entry__ as g

        g.build()
    return LIB


def header_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ptts_[a-z0-9_]+)\s*\(", src)))


def test_header_and_binding_agree():
    from parler_tts_amd import _hip

    assert header_symbols() == sorted(_hip.EXPORTS)


def test_library_exports_every_declared_symbol(built_lib):
    out = subprocess.run(["nm", "-D", built_lib], check=True, capture_output=True, text=True).stdout
    for s in header_symbols():
        assert re.search(rf" T {s}$", out, flags=re.M), f"{s} not exported"


def test_library_loads_and_reports_abi(built_lib):
    from parler_tts_amd import _hip

    lib = _hip.load_library()
    assert lib.ptts_abi_version() == _hip.ABI_VERSION


def test_create_rejects_bad_config_without_touching_a_gpu(built_lib):
    import ctypes as C

    from parler_tts_amd import _hip

    lib = _hip.load_library()
    cfg = _hip.PttsConfig()
    cfg.abi_version = 999
    h = C.c_void_p()
    assert lib.ptts_create(C.byref(cfg), C.byref(h)) == -1
    assert b"ABI version" in lib.ptts_last_error()
    cfg.abi_version = _hip.ABI_VERSION
    cfg.dtype = 1
    cfg.hidden, cfg.heads = 100, 2
    assert lib.ptts_create(C.byref(cfg), C.byref(h)) == -5      # head_dim must be 64
    assert lib.ptts_create(None, C.byref(h)) == -1


def test_python_api_surface_matches_reference_exports():
    import parler_tts_amd as p

    for name in ("ParlerTTSConfig", "ParlerTTSDecoderConfig", "DACConfig", "build_delay_pattern_mask",
                 "apply_delay_pattern_mask"):
        assert hasattr(p, name)
    from parler_tts_amd import modeling

    for name in ("ParlerTTSForConditionalGeneration", "ParlerTTSForCausalLM", "DACModel"):
        assert hasattr(modeling, name)
    import inspect

    sig = inspect.signature(modeling.ParlerTTSForConditionalGeneration.generate)
    assert list(sig.parameters)[:8] == ["self", "inputs", "generation_config", "logits_processor", "stopping_criteria",
                                        "synced_gpus", "streamer", "kwargs"]


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from parler_tts_amd import _hip

    monkeypatch.setattr(_hip, "_lib", None)
    monkeypatch.setattr(_hip, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_hip.HipLibraryMissing):
        _hip.load_library()


def test_config_roundtrip_and_mini_numbers(tmp_path):
    from parler_tts_amd import ParlerTTSConfig, mini_v01_config
    from parler_tts_amd.weights import all_keys

    cfg = mini_v01_config()
    cfg.save_pretrained(str(tmp_path))
    again = ParlerTTSConfig.from_pretrained(str(tmp_path))
    assert again == cfg and again.sampling_rate == 44100
    d = cfg.decoder
    assert (d.vocab_size, d.hidden_size, d.num_hidden_layers, d.num_attention_heads, d.ffn_dim, d.num_codebooks) == \
        (1088, 1024, 24, 16, 4096, 9)
    n_dec = sum(int(__import__("numpy").prod(s)) for k, s in all_keys(cfg).items() if k.startswith("decoder."))
    assert abs(n_dec / 1e6 - 427.1) < 0.2      # SURVEY.md Appendix A parameter budget
    with pytest.raises(ValueError):
        ParlerTTSConfig(vocab_size=3)


def test_weight_norm_folding():
    import torch

    from parler_tts_amd.weights import fold_weight_norm

    v = torch.randn(6, 4, 3)
    g = torch.rand(6, 1, 1) + 0.5
    ref = torch._weight_norm(v, g, 0)
    out = fold_weight_norm({"a.weight_g": g, "a.weight_v": v, "a.bias": torch.zeros(6),
                            "b.parametrizations.weight.original0": g, "b.parametrizations.weight.original1": v})
    assert torch.allclose(out["a.weight"], ref, atol=1e-6) and torch.allclose(out["b.weight"], ref, atol=1e-6)
    assert "a.bias" in out and "a.weight_g" not in out


def test_no_mfma_is_guarded_by_exec_masking_only():
    """MFMA ignores EXEC: a tile guard that hipcc lowers to s_and_saveexec without a skip branch would let the MFMA
    run on unloaded registers.  tools/check_mfma_guards.py compiles the kernels to ISA and looks for that pattern."""
    import importlib.util
    import os

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("check_mfma_guards", os.path.join(root, "tools", "check_mfma_guards.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    srcs = [os.path.join(root, "parler_tts_amd", "csrc", f) for f in ("gemm.hip", "conv.hip", "fused.hip")]
    assert mod.main(srcs) == 0


def test_no_mfma_result_is_read_by_inline_asm_too_early():
    """hipcc pads the gfx950 "MFMA writes a VGPR -> LDS / VMEM / VALU reads it" hazard only for instructions it models;
    an `asm` statement is opaque to it.  tools/check_mfma_asm_hazard.py walks every control-flow path backwards from
    every inline-asm instruction and fails on an MFMA result read closer than LLVM's own wait-state table allows
    (round 1's K-split hand-over was such a pair; ptts_common.h).  The scanner's self-test proves it still sees one."""
    import importlib.util
    import os
    import subprocess
    import tempfile

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("check_mfma_asm_hazard", os.path.join(root, "tools", "check_mfma_asm_hazard.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    srcs = [os.path.join(root, "parler_tts_amd", "csrc", f) for f in ("gemm.hip", "conv.hip", "attn.hip", "fused.hip")]
    assert mod.main(srcs) == 0
    # positive control: the round-1 pattern in ten lines must be flagged
    probe = r'''
#include <hip/hip_runtime.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
__global__ void k(const bf16x8* a, const bf16x8* b, float* o) {
    __shared__ f32x4 sh[64];
    f32x4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[threadIdx.x], b[threadIdx.x], c, 0, 0, 0);
    unsigned addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)&sh[threadIdx.x];
    asm volatile("ds_write_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" ::"v"(addr), "v"(c) : "memory");
    __syncthreads();
    o[threadIdx.x] = sh[threadIdx.x ^ 1][0];
}
'''
    with tempfile.TemporaryDirectory() as td:
        src = os.path.join(td, "probe.hip")
        open(src, "w").write(probe)
        assert mod.main([src]) == 1


def test_generation_config_refuses_fields_it_does_not_serve():
    """A generation_config.json may carry fields this engine does not act on.  Bookkeeping keys and parameters at their
    neutral defaults are ignored; anything that would change the reference's tokens is refused, not dropped."""
    from parler_tts_amd.configuration import GenerationConfig

    GenerationConfig(transformers_version="4.40.0", _from_model_config=True, repetition_penalty=1.0, no_repeat_ngram_size=0,
                     min_new_tokens=10, output_attentions=False).validate()
    assert GenerationConfig(min_new_tokens=10).min_new_tokens == 10
    for bad in (dict(repetition_penalty=1.2), dict(no_repeat_ngram_size=3), dict(bad_words_ids=[[5]]), dict(typical_p=0.9),
                dict(some_future_knob=1)):
        with pytest.raises(NotImplementedError):
            GenerationConfig(**bad).validate()


def test_dac_kernel_counts_hook_answers_without_a_gpu(built_lib):
    """ptts_dac_kernel_counts (ABI v7) is host-side bookkeeping: it must answer (all zero) in a process that never touched a GPU"""
    from parler_tts_amd import _hip

    counts = _hip.dac_kernel_counts(reset=True)
    assert list(counts) == _hip.DAC_KERNEL_KINDS and len(counts) == 14 and set(counts.values()) == {0}
    import ctypes as C

    assert _hip.load_library().ptts_dac_kernel_counts(None, 0, 0) == -1


def test_integration_md_quotes_the_executed_binding():
    """INTEGRATION.md section B shows the reference-side ctypes binding; the block must be tests/reference_side_binding.py byte
    for byte - the file tests/test_gpu_reference_binding.py executes on the GPU - and must not pin an ABI version literal."""
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    m = re.search(r"<!-- BEGIN tests/reference_side_binding.py -->\n```python\n(.*?)```\n<!-- END tests/reference_side_binding.py -->",
                  doc, flags=re.S)
    assert m, "INTEGRATION.md lost the quoted binding"
    src = open(os.path.join(ROOT, "tests", "reference_side_binding.py")).read()
    assert m.group(1) == src
    assert "ptts_abi_version()" in src and not re.search(r"PttsConfig\(\s*\d", src)
    assert not re.search(r"^\s*(from|import)\s+parler_tts_amd", src, flags=re.M)
    # every entry point the binding calls is declared in the header
    used = set(re.findall(r"_lib\.(ptts_[a-z0-9_]+)", src))
    assert used and used <= set(header_symbols())


def test_reference_side_binding_structs_match_the_header_layout(built_lib, monkeypatch):
    """the binding file declares its own ctypes structures (it must not import parler_tts_amd): same fields, offsets and sizes as
    the repo's binding, which test_header_and_binding_agree / the GPU suite hold against the header"""
    import ctypes as C
    import importlib
    import sys

    from parler_tts_amd import _hip

    monkeypatch.setenv("PTTS_HIP_LIB", built_lib)
    sys.modules.pop("tests.reference_side_binding", None)
    rb = importlib.import_module("tests.reference_side_binding")
    for mine, theirs in ((_hip.PttsConfig, rb.PttsConfig), (_hip.PttsSampling, rb.PttsSampling), (_hip.PttsState, rb.PttsState)):
        assert C.sizeof(mine) == C.sizeof(theirs)
        assert [(n, getattr(mine, n).offset, getattr(mine, n).size) for n, *_ in mine._fields_] == \
               [(n, getattr(theirs, n).offset, getattr(theirs, n).size) for n, *_ in theirs._fields_]
    assert rb._lib.ptts_abi_version() == _hip.ABI_VERSION


def test_readme_lists_every_environment_switch_the_code_reads():
    """ONE table of the PTTS_* switches (product / A-B / test hook).  Every switch the library, the Python layer or
    bench.py reads must be in it, and the table must not advertise switches that no longer exist."""
    import glob

    pat = re.compile(r"PTTS_[A-Z0-9_]+")
    read = set()
    for f in glob.glob(os.path.join(ROOT, "parler_tts_amd", "csrc", "*")):
        if f.endswith((".hip", ".inc", ".h")):
            read |= set(re.findall(r'getenv\("(PTTS_[A-Z0-9_]+)"\)', open(f).read()))
    for f in glob.glob(os.path.join(ROOT, "parler_tts_amd", "*.py")) + [os.path.join(ROOT, "bench.py")]:
        read |= set(re.findall(r'environ(?:\.get|\.setdefault|\.pop)?[\(\[]\s*"(PTTS_[A-Z0-9_]+)"', open(f).read()))
    readme = open(os.path.join(ROOT, "README.md")).read()
    a = readme.index("## Environment switches")
    table = readme[a:]
    listed = set(pat.findall(table))
    internal = {"PTTS_STAMPS_OUT"}                      # diagnostic build only (tools/gpu_calls.md)
    missing = sorted(read - listed - internal)
    assert not missing, f"switches read by the code but absent from README's table: {missing}"
    stale = sorted(s for s in listed - read if s not in ("PTTS_HIP_LIB",) and not s.startswith("PTTS_DAC_TILE"))
    assert not stale, f"README lists switches nothing reads any more: {stale}"
