"""Does a power-of-two leading dimension of the k-major operand (dG: 4H = 2048 floats = 8 KB between consecutive k) slow
the weight-gradient product down? pp_gemm_f32 on the dW_ih shape with lda = 2048 vs padded strides."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pyprob_amd import lib as L

lib = L.load()
dev = torch.device('cuda:0')


def run(M, N, K, lda, ldb, akm, bkm, split, iters=100, label=''):
    A = torch.randn((K if akm else M) * lda + 64, device=dev)
    B = torch.randn((K if bkm else N) * ldb + 64, device=dev)
    Cm = torch.zeros(M, N, device=dev)
    g = L.pp_gemm_args()
    g.A, g.lda, g.B, g.ldb, g.C, g.ldc = A.data_ptr(), lda, B.data_ptr(), ldb, Cm.data_ptr(), N
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
    print('%-34s M=%5d N=%4d K=%5d lda=%5d ldb=%5d : %7.1f us %6.1f TFLOP/s' % (label, M, N, K, lda, ldb, us, 2.0 * M * N * K / us / 1e6))


for lda in (2048, 2052, 2080, 2112, 2176):
    run(2048, 212, 1024, lda, 212, 1, 1, 1, label='dW_ih = dG^T X (TN, split)')
for lda in (2048, 2080):
    run(1024, 212, 2048, lda, 212, 0, 1, 1, label='dX = dG W_ih (NN, split)')
for ldc in (512, 544):
    run(271, 512, 1024, 272, ldc, 1, 1, 1, label='dW1 = dZ1^T Hs (TN, split) ldb')
run(1024, 2048, 212, 212, 212, 0, 0, 0, label='fwd X W_ih^T')
