#!/usr/bin/env python3
"""Stand-alone folded-LayerNorm GEMM (ptts_test_gemm prologue 4) repeated under GPU sharing: bitwise repeatable?"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from parler_tts_amd import _hip

tag = sys.argv[1] if len(sys.argv) > 1 else ""
pro = int(sys.argv[2]) if len(sys.argv) > 2 else 4
hip = _hip.load_library()
g = torch.Generator().manual_seed(5)
bad_total = 0
shapes = [(32, 4096, 1024), (32, 1024, 1024)]
for (M, N, K) in shapes:
    W = (torch.randn(N, K, generator=g) * 0.05).cuda()
    X = (0.3 + torch.randn(M, K, generator=g)).cuda()
    gamma = (1 + 0.2 * torch.randn(K, generator=g)).cuda()
    beta = (0.1 * torch.randn(K, generator=g)).cuda()
    bias = torch.zeros(N).cuda()
    ref = None
    bad = 0
    for it in range(int(sys.argv[3]) if len(sys.argv) > 3 else 400):
        Y = torch.empty(M, N, device="cuda")
        rc = hip.ptts_test_gemm(1, M, N, K, W.data_ptr(), X.data_ptr(), pro, gamma.data_ptr(), beta.data_ptr(),
                                C.c_float(1e-5), 0, bias.data_ptr(), Y.data_ptr(), None)
        assert rc == 0
        if ref is None:
            ref = Y.clone()
        elif not torch.equal(Y, ref):
            bad += 1
            if bad <= 6:
                d = (Y != ref)
                rows = d.any(1).nonzero().flatten().tolist()
                cols = d.any(0).nonzero().flatten().tolist()
                print(f"{tag} N={N} it={it}: {int(d.sum())} elements differ; rows {rows[:8]} cols {cols[:8]}..{cols[-1]} maxdiff {(Y - ref).abs().max().item():.3e}", flush=True)
    print(f"{tag} pro={pro} M={M} N={N} K={K}: {bad} repeats differ", flush=True)
    bad_total += bad
