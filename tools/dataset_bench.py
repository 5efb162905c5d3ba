"""Packed on-disk dataset (pyprob_amd/dataset.py): write / open / minibatch-packing rates, and end-to-end offline
training fed by the prefetching loader. python tools/dataset_bench.py [n_traces] [gumm]"""
import os, sys, time, tempfile, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np
if os.environ.get('PP_SWITCH'):
    sys.setswitchinterval(float(os.environ['PP_SWITCH']))
import torch  # noqa: F401  (first import of a fresh box takes seconds: keep it out of the timed loops)
from pyprob_amd.dataset import PackedTraceDataset, PackedTraceWriter

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
gumm = len(sys.argv) > 2 and sys.argv[2] == 'gumm'
root = tempfile.mkdtemp(prefix='pp_ds_')
try:
    rng = np.random.default_rng(0)
    t0 = time.perf_counter()
    per = 250000
    for s in range(0, n, per):
        m = min(per, n - s)
        if gumm:
            from helpers import synthetic_gumm_arrays
            arrays, addresses = synthetic_gumm_arrays(m, seed=s, max_iter=6)
            lens, ids, value, prior, obs = (arrays[k] for k in ('trace_len', 'addr_idx', 'values', 'prior', 'obs'))
            table = [(a, 'Uniform', None) for a in addresses]
        else:
            mu = rng.normal(1.0, 5 ** 0.5, m).astype(np.float32)
            obs = (mu[:, None] + rng.normal(0, 2 ** 0.5, (m, 2))).astype(np.float32)
            lens, ids, value = np.ones(m, np.int64), np.zeros(m, np.int64), mu
            prior = np.tile(np.asarray([[1.0, 5 ** 0.5]], np.float32), (m, 1))
            table = [('16__forward__mu__Normal__1', 'Normal', None)]
        with PackedTraceWriter(os.path.join(root, 'shard_%04d' % (s // per)), ['obs0', 'obs1'], [1, 1]) as w:
            w.add_columns(lens, table, ids, value, prior, obs)
    t1 = time.perf_counter()
    size = sum(os.path.getsize(os.path.join(dp, f)) for dp, _, fs in os.walk(root) for f in fs)
    print('write: %d traces in %.2f s = %.2f M traces/s, %.1f MB on disk (%.1f B/trace)' % (n, t1 - t0, n / (t1 - t0) / 1e6, size / 1e6, size / n))
    ds = PackedTraceDataset(root)
    ds.sorted_indices()
    t2 = time.perf_counter()
    print('open + sorted index: %.3f s' % (t2 - t1))

    class Spec:
        addresses = ds.addresses
        address_id = {a[0]: i for i, a in enumerate(ds.addresses)}
    B = 1024
    sampler = ds.sampler(B, 0, 1)
    t3 = time.perf_counter()
    k = 0
    for ids_ in sampler:
        ds.batch(ids_, Spec)
        k += 1
        if k == 200:
            break
    t4 = time.perf_counter()
    print('host packing (native, straight from the mapped columns), one thread: %.2f M traces/s (%.2f ms per 1024-trace minibatch)' % (k * B / (t4 - t3) / 1e6, (t4 - t3) / k * 1e3))
    import torch
    if torch.cuda.is_available():
        import bench
        dev = torch.device('cuda:0')
        eng = bench.make_engine(512, dev, 1)
        if gumm:
            eng.add_addresses([(a[0], a[1], a[2]) for a in ds.addresses])
        for b in ds.loader(eng.spec, B, dev, epochs=1, prefetch=8):
            eng.train_step(b, 1e-3)
            k -= 1
            if k <= 150:
                break
        torch.cuda.synchronize()
        t5 = time.perf_counter()
        steps = 0
        for b in ds.loader(eng.spec, B, dev, epochs=1, prefetch=8, workers=int(os.environ.get('PP_LOADER_WORKERS', '0'))):
            eng.train_step(b, 1e-3)
            steps += 1
            if steps == 400:
                break
        torch.cuda.synchronize()
        t6 = time.perf_counter()
        print('offline training fed by the loader (disk -> host pack -> H2D -> step), Python per step: %.2f M traces/s (%.3f ms/step)' % (steps * B / (t6 - t5) / 1e6, (t6 - t5) / steps * 1e3))
        # the route learn_inference_network(dataset_dir=...) takes: runs of 64 steps inside pp_train_steps
        it = iter(ds.sampler(B, 0, 1))
        runs = [[next(it) for _ in range(64)] for _ in range(8)]
        eng.train_run(ds, runs[0], [1e-3] * 64)
        torch.cuda.synchronize()
        t7 = time.perf_counter()
        for r in runs[1:]:
            l, s_ = eng.train_run(ds, r, [1e-3] * 64)
            l.cpu()
        torch.cuda.synchronize()
        t8 = time.perf_counter()
        print('offline training, native loop (pp_train_steps, 64 steps per call): %.2f M traces/s (%.3f ms/step)' % (7 * 64 * B / (t8 - t7) / 1e6, (t8 - t7) / (7 * 64) * 1e3))
finally:
    shutil.rmtree(root, ignore_errors=True)
