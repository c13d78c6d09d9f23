#!/usr/bin/env python3
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from parler_tts_amd import _hip
hip = _hip.load_library()
g = torch.Generator().manual_seed(11)
for (M, N, K) in [(32, 1024, 1024), (2, 384, 128), (32, 1024, 4096)]:
    W = (torch.randn(N, K, generator=g) * 0.05).cuda()
    X = (0.2 + torch.randn(M, K, generator=g)).cuda()
    z = torch.zeros(max(N, K)).cuda()
    Y = torch.full((M, N), 7.0, device="cuda")
    rc = hip.ptts_test_gemm(1, M, N, K, W.data_ptr(), X.data_ptr(), 5, z.data_ptr(), z.data_ptr(), C.c_float(1e-5), 0, z.data_ptr(), Y.data_ptr(), None)
    ref = X.bfloat16().double() @ W.bfloat16().double().t()
    bad = ~((Y.double() - ref).abs() < 1e-2)
    print((M, N, K), "rc", rc, "bad elements", int(bad.sum()), "of", M * N)
    if bad.any():
        print("  bad rows:", bad.any(1).nonzero().flatten().tolist()[:40])
        cols = bad.any(0).nonzero().flatten().tolist()
        print("  bad cols (first 40):", cols[:40], "count", len(cols))
        r, c = bad.nonzero()[0].tolist()
        print("  sample: Y=%g ref=%g" % (Y[r, c].item(), ref[r, c].item()))
        # per-wave K-slice check: which k-ranges contribute wrongly?  recompute with K-slices zeroed
        for ks in range(0, K, K // 8):
            X2 = X.clone(); X2[:, ks:ks + K // 8] = 0
            Y2 = torch.empty(M, N, device="cuda")
            hip.ptts_test_gemm(1, M, N, K, W.data_ptr(), X2.data_ptr(), 5, z.data_ptr(), z.data_ptr(), C.c_float(1e-5), 0, z.data_ptr(), Y2.data_ptr(), None)
            ref2 = X2.bfloat16().double() @ W.bfloat16().double().t()
            print(f"    zero k[{ks}:{ks + K // 8}]: bad {int((~((Y2.double() - ref2).abs() < 1e-2)).sum())}")
