"""How many host cores does this box really give to a process tree? N forked processes run the same pure-Python loop."""
import multiprocessing as mp, os, time
def work(n):
    s = 0
    for i in range(n):
        s += i * i % 7
    return s
def child(conn, n):
    t0 = time.perf_counter(); work(n); conn.send(time.perf_counter() - t0); conn.close()
if __name__ == '__main__':
    print('cpu_count', os.cpu_count(), 'affinity', len(os.sched_getaffinity(0)))
    try:
        print('cgroup cpu.max:', open('/sys/fs/cgroup/cpu.max').read().strip())
    except OSError as e:
        print('cgroup cpu.max unreadable', e)
    n = 3000000
    ctx = mp.get_context('fork')
    for N in (1, 4, 8, 16, 32, 64, 128):
        pipes, procs = [], []
        t0 = time.perf_counter()
        for _ in range(N):
            a, b = ctx.Pipe(); p = ctx.Process(target=child, args=(b, n)); p.start(); pipes.append(a); procs.append(p)
        ts = [a.recv() for a in pipes]
        for p in procs: p.join()
        wall = time.perf_counter() - t0
        print('N=%3d  per-process loop %.2f s (min %.2f max %.2f)  wall %.2f s  -> %.1f loops/s aggregate' % (N, sum(ts) / N, min(ts), max(ts), wall, N / wall))
