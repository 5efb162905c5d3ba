"""Kernel timeline of ONE lock-step posterior call of the Marsaglia program (H = 512): run under
  rocprofv3 --kernel-trace -d DIR -o p -- python tools/gumm_timeline.py [particles]
then  python tools/rocprof_summary.py DIR/p_results.db out.csv sequence 'is_fused_kernel<1'
lists the launches between the first statements of the last two calls with their start offsets, durations and the idle gap
before each (host time + synchronisation between dependent launches)."""
import contextlib
import io
import os
import sys
import warnings

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                                   # noqa: E402
import bench                                                   # noqa: E402
from pyprob_amd.state import InferenceEngine, InferenceNetwork  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
GUM, GUMM = bench.api_models()
model = GUMM()
torch.manual_seed(123)
warnings.simplefilter('ignore')
with contextlib.redirect_stdout(io.StringIO()):
    model.learn_inference_network(num_traces=96 * 1024, inference_network=InferenceNetwork.LSTM,
                                  observe_embeddings={'obs0': {'dim': 32}, 'obs1': {'dim': 32}}, batch_size=1024, lstm_dim=512, seed=1)
IC = InferenceEngine.IMPORTANCE_SAMPLING_WITH_INFERENCE_NETWORK
for i in range(8):
    post = model.posterior_results(n, IC, observe={'obs0': 4, 'obs1': 5}, lock_step=True, seed=i)
    _ = post.effective_sample_size
torch.cuda.synchronize()
