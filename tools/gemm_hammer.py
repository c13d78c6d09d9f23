#!/usr/bin/env python3
"""Native repeatability hammer for one GEMM instance: PTTS_TEST_GEMM_ITERS launches compared bitwise on the device.
   python3 tools/gemm_hammer.py <prologue 2|4|5|6> <M> <N> <K> <iters> [dtype]     (start two at once to share the GPU)"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pro, M, N, K, iters = (int(x) for x in sys.argv[1:6])
dtype = sys.argv[6] if len(sys.argv) > 6 else "bf16"
os.environ["PTTS_TEST_GEMM_ITERS"] = str(iters)
import torch

from parler_tts_amd import _hip

hip = _hip.load_library()
g = torch.Generator().manual_seed(5)
W = (torch.randn(N, K, generator=g) * 0.05).cuda()
X = (0.3 + torch.randn(M, K, generator=g)).cuda()
gamma = (1 + 0.2 * torch.randn(K, generator=g)).cuda()
beta = (0.1 * torch.randn(K, generator=g)).cuda()
bias = torch.zeros(N).cuda()
Y = torch.empty(M, N, device="cuda")
rc = hip.ptts_test_gemm(1 if dtype == "bf16" else 0, M, N, K, W.data_ptr(), X.data_ptr(), pro, gamma.data_ptr(), beta.data_ptr(),
                        C.c_float(1e-5), 0, bias.data_ptr(), Y.data_ptr(), None)
print("rc", rc, hip.ptts_last_error().decode() if rc else "")
