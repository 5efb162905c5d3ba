"""Why does a 20-step timed region after 5 warm-up steps run slower than steps 100..500?  

Cold process -> engine -> steps 1..N of the config-2 training step, each bracketed by an event pair on the launch
stream (GPU time between consecutive step boundaries) and by perf_counter on the host (enqueue time of the step's two
C calls).  Three phases are told apart by what they correlate with:
  * host enqueue  : host_us > gpu_us  (the GPU waits for the interpreter: cold code paths, first use of a batch descriptor)
  * clock ramp    : gpu_us falls smoothly over the first milliseconds although host_us is already low
  * first touch   : gpu_us of the first use of each resident batch (distinct descriptors, cold L2 / TLB) against re-use
Prints one JSON object (per-step arrays + a summary)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
from pyprob_amd.packed import ColumnarDataset


def main():
    n_steps = int(sys.argv[1]) if len(sys.argv) > 1 else 120
    n_batches = int(sys.argv[2]) if len(sys.argv) > 2 else 25       # the driver's run touches W + K = 25 descriptors
    idle_s = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
    dev = torch.device('cuda:0')
    t_proc = time.perf_counter()
    eng = bench.make_engine(512, dev, seed=123)
    obs, mu, prior = bench.synth_gum_dataset(1000000, dev, seed=1000)
    ds = ColumnarDataset(obs, mu, prior, 1024)
    cache = {}
    batches = [ds.batch(i, 0, 1, cache) for i in range(n_batches)]
    torch.cuda.synchronize()
    setup_s = time.perf_counter() - t_proc
    if idle_s:
        time.sleep(idle_s)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n_steps + 1)]
    host = np.zeros(n_steps)
    ev[0].record()
    t_all = time.perf_counter()
    for i in range(n_steps):
        t0 = time.perf_counter()
        eng.train_step(batches[i % n_batches], 1e-3)
        ev[i + 1].record()
        host[i] = (time.perf_counter() - t0) * 1e6
    enq_s = time.perf_counter() - t_all
    torch.cuda.synchronize()
    wall_s = time.perf_counter() - t_all
    gpu = np.array([ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(n_steps)])

    def seg(a, lo, hi):
        return round(float(np.mean(a[lo:hi])), 2) if hi <= len(a) else None
    summary = dict(
        n_steps=n_steps, n_batches=n_batches, setup_s=round(setup_s, 2), idle_before_s=idle_s,
        wall_us_per_step=round(wall_s / n_steps * 1e6, 2), host_enqueue_us_per_step=round(enq_s / n_steps * 1e6, 2),
        gpu_us_steps_1_5=seg(gpu, 0, 5), gpu_us_steps_6_25=seg(gpu, 5, 25), gpu_us_steps_26_60=seg(gpu, 25, 60),
        gpu_us_steps_61_end=seg(gpu, 60, n_steps),
        host_us_steps_1_5=seg(host, 0, 5), host_us_steps_6_25=seg(host, 5, 25), host_us_steps_26_60=seg(host, 25, 60),
        host_us_steps_61_end=seg(host, 60, n_steps),
        gpu_us_median_last_half=round(float(np.median(gpu[n_steps // 2:])), 2))
    print(json.dumps(dict(summary=summary, gpu_us=[round(float(x), 1) for x in gpu],
                          host_us=[round(float(x), 1) for x in host])))


if __name__ == '__main__':
    main()
