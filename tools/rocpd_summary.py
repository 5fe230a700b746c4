"""Summarise a rocprofv3 (rocpd sqlite) result database as a per-kernel stats table (the same columns as
`rocprofv3 --stats` kernel_stats.csv): name, calls, total ns, average ns, min, max, percentage.

    python tools/rocpd_summary.py gpurun_out/prof_r01/r01_results.db > profiles/r01_kernel_stats.csv
"""
import sqlite3
import sys


def main(path):
    c = sqlite3.connect(path)
    rows = c.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
                     "max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), "
                     "max(grid_x), max(grid_y), max(workgroup_x) from kernels group by name order by sum(duration) desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    print("Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs,Percentage,VGPR,AGPR,SGPR,LDS,GridX,GridY,WorkgroupX")
    for r in rows:
        print('"%s",%d,%d,%.1f,%d,%d,%.2f,%d,%d,%d,%d,%d,%d,%d' % (r[0], r[1], r[2], r[3], r[4], r[5], 100.0 * r[2] / tot, *r[6:]))


if __name__ == "__main__":
    main(sys.argv[1])
