#!/usr/bin/env python3
"""Stand-alone GEMM hook on the small shapes of the tiny test config (cannot fault: fixed buffers, no token feedback)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from parler_tts_amd import _hip

hip = _hip.load_library()
g = torch.Generator().manual_seed(11)
for dtype in ("bf16", "fp32"):
    for pro in (4, 5, 2):
        for (M, N, K) in [(2, 384, 128), (4, 128, 128), (2, 512, 128), (2, 128, 512), (4, 1152, 128), (2, 192, 64), (16, 1024, 1024),
                          (32, 1024, 4096), (32, 1024, 1024), (32, 4096, 1024)]:
            if pro == 2 and K > 2048:
                continue
            W = (torch.randn(N, K, generator=g) * 0.05).cuda()
            X = (0.2 + torch.randn(M, K, generator=g)).cuda()
            gamma = (1 + 0.2 * torch.randn(K, generator=g)).cuda()
            beta = (0.1 * torch.randn(K, generator=g)).cuda()
            bias = torch.zeros(N).cuda()
            Y = torch.full((M, N), 7.0, device="cuda")
            rc = hip.ptts_test_gemm(1 if dtype == "bf16" else 0, M, N, K, W.data_ptr(), X.data_ptr(), pro, gamma.data_ptr(),
                                    beta.data_ptr(), C.c_float(1e-5), 0, bias.data_ptr(), Y.data_ptr(), None)
            if rc != 0:
                print(dtype, pro, (M, N, K), "rc", rc, hip.ptts_last_error())
                continue
            Xd, Wd = X.double(), W.double()
            if dtype == "bf16":
                Wd = W.bfloat16().double()
                if pro != 2:
                    Xd = X.bfloat16().double()
            ref = (torch.nn.functional.layer_norm(Xd, (K,), gamma.double(), beta.double(), 1e-5) if pro in (2, 4) else Xd)
            if dtype == "bf16" and pro == 2:
                ref = ref.float().bfloat16().double()
            ref = ref @ Wd.t()
            err = (Y.double() - ref).abs().max().item()
            print(f"{dtype} pro={pro} M={M} N={N} K={K}: err {err:.3e} scale {ref.abs().max().item():.2f} nan {bool(torch.isnan(Y).any())}", flush=True)
