#!/usr/bin/env python3
"""Per-kernel / per-launch-shape averages of rocprofv3 --pmc counters (rocpd sqlite .db) -> CSV."""
import csv
import sqlite3
import sys


def main(db, out):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select kernel_name, grid_size_x/workgroup_size_x, grid_size_y, grid_size_z, counter_name, count(*), "
                       "avg(value), avg(duration) from counters_collection group by kernel_name, grid_size_x, grid_size_y, "
                       "grid_size_z, counter_name order by avg(duration)*count(*) desc").fetchall()
    with open(out, 'w', newline='') as f:
        w = csv.writer(f)
        w.writerow(['kernel', 'wg_x', 'wg_y', 'wg_z', 'counter', 'dispatches', 'avg_value', 'avg_duration_ns'])
        for r in rows:
            w.writerow([r[0], r[1], r[2], r[3], r[4], r[5], round(r[6], 3), round(r[7], 1)])
    return rows


if __name__ == '__main__':
    rows = main(sys.argv[1], sys.argv[2])
    for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 12]:
        print('%-60s (%4d,%4d,%3d) %-26s n=%4d avg=%12.2f dur=%8.1f ns' % (r[0].replace('void pp::', '')[:60], r[1], r[2], r[3], r[4], r[5], r[6], r[7]))
