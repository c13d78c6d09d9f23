#!/usr/bin/env python3
"""Launch the one-off tile GEMM (ptts_test_gemm prologue 7 / 8) at a list of shapes, for a rocprofv3 kernel trace:
    rocprofv3 --kernel-trace --output-format csv -d /tmp/tg -- python3 tools/tile_gemm_probe.py  M,N,K[,pro] ...
The trace's gemm_tile_kernel rows are in call order (3 launches per shape)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from parler_tts_amd import _hip  # noqa: E402

hip = _hip.load_library()
for spec in sys.argv[1:]:
    v = [int(x) for x in spec.split(",")]
    M, N, K = v[:3]
    pro = v[3] if len(v) > 3 else 7
    W = (torch.randn(N, K) * 0.05).cuda()
    X = torch.randn(M, K).cuda()
    g = torch.ones(K).cuda()
    b = torch.zeros(K).cuda()
    bias = torch.zeros(N).cuda()
    Y = torch.empty(M, N, device="cuda")
    for _ in range(3):
        rc = hip.ptts_test_gemm(1, M, N, K, W.data_ptr(), X.data_ptr(), pro, g.data_ptr(), b.data_ptr(), C.c_float(1e-5), 0, None,
                                Y.data_ptr(), None)
        assert rc == 0, hip.ptts_last_error()
    torch.cuda.synchronize()
    print(spec, "ok", flush=True)
