"""Wall time of the Marsaglia program's lock-step posterior call, un-instrumented (no event pairs armed):
   [H=512] [DEPTH=1] python tools/gumm_call_bench.py [particles] [calls]
(H = 32 / 64 / 128, DEPTH 1..4: the N-row statements are csrc/is_step_small.hip; PP_IS_STEP_FUSED=0: the chain of GEMM launches with
the rows gathered and scattered by the host)"""
import contextlib, io, os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from pyprob_amd.state import InferenceEngine, InferenceNetwork
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 12
GUM, GUMM = bench.api_models()
model = GUMM()
torch.manual_seed(123)
warnings.simplefilter('ignore')
with contextlib.redirect_stdout(io.StringIO()):
    model.learn_inference_network(num_traces=96 * 1024, inference_network=InferenceNetwork.LSTM,
                                  observe_embeddings={'obs0': {'dim': 32}, 'obs1': {'dim': 32}}, batch_size=1024, lstm_dim=int(os.environ.get('H', '512')),
                                  lstm_depth=int(os.environ.get('DEPTH', '1')), seed=1)
IC = InferenceEngine.IMPORTANCE_SAMPLING_WITH_INFERENCE_NETWORK
for i in range(4):
    post = model.posterior_results(n, IC, observe={'obs0': 4, 'obs1': 5}, lock_step=True, seed=i)
    _ = post.effective_sample_size
torch.cuda.synchronize()
ts = []
for i in range(calls):
    t0 = time.perf_counter()
    post = model.posterior_results(n, IC, observe={'obs0': 4, 'obs1': 5}, lock_step=True, seed=10 + i)
    _ = post.effective_sample_size
    torch.cuda.synchronize()
    ts.append(time.perf_counter() - t0)
ts.sort()
print('H=%s depth=%s PP_IS_STEP_FUSED=%s PP_IS_PART_POLL=%s: %d particles, %d paths: median %.3f ms/call (min %.3f) -> %.2f M particles/s' % (
    os.environ.get('H', '512'), os.environ.get('DEPTH', '1'), os.environ.get('PP_IS_STEP_FUSED', '1'), os.environ.get('PP_IS_PART_POLL', '1'), n, post.num_paths, ts[len(ts) // 2] * 1e3, ts[0] * 1e3, n / ts[len(ts) // 2] / 1e6))
