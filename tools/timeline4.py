"""Per-trip clock64 stamps of the async GEMM main loop (workgroup 0): PP_DBG_STAMP=1 python tools/timeline4.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pyprob_amd import lib as L
lib = L.load(); dev = torch.device('cuda:0')
buf = torch.zeros(128, dtype=torch.int64, device=dev)
lib.pp_debug_timeline(buf.data_ptr())
M, N, K, akm, bkm = [int(x) for x in (sys.argv[1:6] if len(sys.argv) > 5 else (1024, 212, 2048, 0, 1))]
A = torch.randn((K, M) if akm else (M, K), device=dev)
B = torch.randn((K, N) if bkm else (N, K), device=dev)
Cm = torch.zeros(M, N, device=dev)
g = L.pp_gemm_args()
g.A, g.lda, g.B, g.ldb, g.C, g.ldc = A.data_ptr(), A.shape[1], B.data_ptr(), B.shape[1], Cm.data_ptr(), N
g.M, g.N, g.K, g.a_kmajor, g.b_kmajor, g.split_k, g.accumulate = M, N, K, akm, bkm, 1, 1
for rep in range(3):
    buf.zero_()
    lib.pp_gemm_f32(C.byref(g), L.stream_ptr()); torch.cuda.synchronize()
    t = [x for x in buf.tolist()[16:] if x]
    print('trips (2 slabs each), clock64 deltas:', [b - a for a, b in zip(t, t[1:])])
lib.pp_debug_timeline(None)
