"""Where a loader worker spends its time per minibatch (single thread, no training running)."""
import os, sys, time, tempfile, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pyprob_amd.dataset import PackedTraceDataset, PackedTraceWriter
import bench
root = tempfile.mkdtemp()
try:
    rng = np.random.default_rng(0); m = 200000
    mu = rng.normal(1.0, 5 ** 0.5, m).astype(np.float32)
    with PackedTraceWriter(os.path.join(root, 's'), ['obs0', 'obs1'], [1, 1]) as w:
        w.add_columns(np.ones(m, np.int64), [('16__forward__mu__Normal__1', 'Normal', None)], np.zeros(m, np.int64), mu,
                      np.tile(np.asarray([[1.0, 5 ** 0.5]], np.float32), (m, 1)), (mu[:, None] + rng.normal(0, 2 ** 0.5, (m, 2))).astype(np.float32))
    ds = PackedTraceDataset(root); dev = torch.device('cuda:0'); eng = bench.make_engine(512, dev, 1)
    ids = [b for b in ds.sampler(1024)][:100]
    for name, fn in (('gather', lambda i: ds.gather(i)), ('batch (gather+pack)', lambda i: ds.batch(i, eng.spec)),
                     ('batch + to(device)', lambda i: ds.batch(i, eng.spec).to(dev))):
        fn(ids[0]); torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in ids: fn(i)
        torch.cuda.synchronize(); print('%-22s %.1f us per minibatch' % (name, (time.perf_counter() - t0) / len(ids) * 1e6))
    st = torch.cuda.Stream()
    t0 = time.perf_counter()
    for i in ids:
        with torch.cuda.stream(st):
            b = ds.batch(i, eng.spec).to(dev); ev = torch.cuda.Event(); ev.record(st)
    torch.cuda.synchronize(); print('%-22s %.1f us per minibatch' % ('... on a side stream + event', (time.perf_counter() - t0) / len(ids) * 1e6))
    # sequential feed (no loader thread): pack + upload + enqueue the step in one thread
    import bench as _b
    ids = [b for b in ds.sampler(1024)][:150]
    for i in ids[:20]: eng.train_step(ds.batch(i, eng.spec).to(dev), 1e-3)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in ids[20:]: eng.train_step(ds.batch(i, eng.spec).to(dev), 1e-3)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print('sequential pack + upload + step: %.1f us per step = %.2f M traces/s' % (dt / 130 * 1e6, 130 * 1024 / dt / 1e6))
    t0 = time.perf_counter(); n = 0
    for b in ds.loader(eng.spec, 1024, dev, epochs=1, prefetch=8, workers=1):
        eng.train_step(b, 1e-3); n += 1
        if n == 130: break
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print('loader thread: %.1f us per step = %.2f M traces/s' % (dt / 130 * 1e6, 130 * 1024 / dt / 1e6))
finally:
    shutil.rmtree(root, ignore_errors=True)
