#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --stats result (rocpd sqlite .db) as a per-kernel, per-launch-shape CSV."""
import csv
import sqlite3
import sys


def main(db, out, steps=None):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select name, grid_x, grid_y, grid_z, workgroup_x, count(*), sum(duration), avg(duration), "
                       "min(duration), max(duration) from kernels group by name, grid_x, grid_y, grid_z "
                       "order by sum(duration) desc").fetchall()
    total = sum(r[6] for r in rows)
    with open(out, 'w', newline='') as f:
        w = csv.writer(f)
        w.writerow(['kernel', 'workgroups_x', 'workgroups_y', 'workgroups_z', 'threads', 'calls', 'total_us', 'avg_us',
                    'min_us', 'max_us', 'percent'])
        for r in rows:
            w.writerow([r[0], r[1] // max(r[4], 1), r[2], r[3], r[4], r[5], round(r[6] / 1e3, 2), round(r[7] / 1e3, 3),
                        round(r[8] / 1e3, 3), round(r[9] / 1e3, 3), round(100.0 * r[6] / total, 2)])
        w.writerow(['TOTAL', '', '', '', '', sum(r[5] for r in rows), round(total / 1e3, 2), '', '', '', 100.0])
    print('wrote', out, 'kernels', len(rows), 'total_ms', total / 1e6)


def sequence(db, out, marker='adam_kernel', which=-3):
    """The kernels of ONE step in launch order (between two launches of `marker`, the `which`-th interval), with start offsets."""
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select name, grid_x, workgroup_x, start, end from kernels order by start").fetchall()
    marks = [i for i, r in enumerate(rows) if marker in r[0]]
    a, b = marks[which - 1], marks[which]
    t0 = rows[a + 1][3]
    with open(out, 'w', newline='') as f:
        w = csv.writer(f)
        w.writerow(['start_us', 'duration_us', 'gap_before_us', 'workgroups', 'kernel'])
        prev_end = rows[a][4]
        for r in rows[a + 1:b + 1]:
            w.writerow([round((r[3] - t0) / 1e3, 2), round((r[4] - r[3]) / 1e3, 2), round((r[3] - prev_end) / 1e3, 2),
                        r[1] // max(r[2], 1), r[0].split('(')[0][:70]])
            prev_end = r[4]
    print(open(out).read())


if __name__ == '__main__':
    if len(sys.argv) > 3 and sys.argv[3] == 'sequence':
        sequence(sys.argv[1], sys.argv[2], *(sys.argv[4:5]))
    else:
        main(sys.argv[1], sys.argv[2])
