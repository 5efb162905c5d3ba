"""Micro-benchmark of pp_gemm_f32 at a few shapes (device tensors, HIP-event timing)."""
import ctypes as C
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pyprob_amd import lib as L

lib = L.load()
dev = torch.device('cuda:0')


def run(M, N, K, akm=0, bkm=0, split=0, iters=50):
    A = torch.randn((K, M) if akm else (M, K), device=dev)
    B = torch.randn((K, N) if bkm else (N, K), device=dev)
    Cm = torch.zeros(M, N, device=dev)
    g = L.pp_gemm_args()
    g.A, g.lda, g.B, g.ldb, g.C, g.ldc = A.data_ptr(), A.shape[1], B.data_ptr(), B.shape[1], Cm.data_ptr(), N
    g.M, g.N, g.K, g.a_kmajor, g.b_kmajor, g.split_k, g.accumulate = M, N, K, akm, bkm, split, split
    st = L.stream_ptr()
    for _ in range(5):
        lib.pp_gemm_f32(C.byref(g), st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        lib.pp_gemm_f32(C.byref(g), st)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    print('M=%6d N=%5d K=%5d akm=%d bkm=%d split=%d : %8.1f us  %7.1f TFLOP/s' % (M, N, K, akm, bkm, split, us, 2.0 * M * N * K / us / 1e6))


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'split':
        run(1024, 212, 2048, 0, 1, 1)
        run(2048, 212, 1024, 1, 1, 1)
        sys.exit(0)
    run(1024, 2048, 212)
    run(8192, 2048, 212)
    run(131072, 2048, 212, iters=10)
    run(131072, 2048, 512, iters=10)
    run(4096, 4096, 4096, iters=5)
    run(2048, 212, 1024, 1, 1, 1)
    run(1024, 212, 2048, 0, 1, 1)
    run(1024, 271, 512)
    run(64, 64, 1024, 1, 1, 1)
    run(64, 64, 64)
