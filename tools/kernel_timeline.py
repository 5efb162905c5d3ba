"""Phase stamps of the wave-direct GEMM tile (workgroup (1,1)): PP_DBG_STAMP=1 python tools/timeline5.py M N K akm bkm"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pyprob_amd import lib as L
lib = L.load(); dev = torch.device('cuda:0')
buf = torch.zeros(128, dtype=torch.int64, device=dev)
lib.pp_debug_timeline(buf.data_ptr())
M, N, K, akm, bkm = [int(x) for x in (sys.argv[1:6] if len(sys.argv) > 5 else (1024, 271, 512, 0, 0))]
A = torch.randn((K, M) if akm else (M, K), device=dev)
B = torch.randn((K, N) if bkm else (N, K), device=dev)
ld = (N + 3) // 4 * 4
Cm = torch.zeros(M, ld, device=dev)
g = L.pp_gemm_args()
g.A, g.lda, g.B, g.ldb, g.C, g.ldc = A.data_ptr(), A.shape[1], B.data_ptr(), B.shape[1], Cm.data_ptr(), ld
g.M, g.N, g.K, g.a_kmajor, g.b_kmajor = M, N, K, akm, bkm
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for rep in range(3):
    buf.zero_()
    e0.record(); lib.pp_gemm_f32(C.byref(g), L.stream_ptr()); e1.record(); torch.cuda.synchronize()
    t = buf.tolist()[32:37]
    print('init +%d  first loads issued +%d  first slab consumed +%d  loop done +%d   (kernel incl. launch %.1f us)' % (t[1] - t[0], t[2] - t[1], t[3] - t[2], t[4] - t[3], e0.elapsed_time(e1) * 1e3))
lib.pp_debug_timeline(None)
