"""Per-kernel stats of a rocprofv3 (rocpd sqlite) database split by launch geometry (the same kernel name serves several problem
shapes: ToMP's GEMMs):  python tools/rocpd_by_grid.py <results.db> [min_total_us]"""
import sqlite3
import sys


def main(path, min_us=0.0):
    c = sqlite3.connect(path)
    rows = c.execute("select name, grid_x, grid_y, grid_z, workgroup_x, count(*), sum(duration), avg(duration), min(duration) "
                     "from kernels group by name, grid_x, grid_y, grid_z, workgroup_x order by sum(duration) desc").fetchall()
    print("name | grid (threads) | wg | calls | total_us | avg_us | min_us")
    for r in rows:
        if r[6] / 1e3 < min_us:
            continue
        name = r[0].replace("(anonymous namespace)::", "")[:56]
        print(f"{name:58s} {r[1]:>7d}x{r[2]:<5d}x{r[3]:<3d} {r[4]:>5d} {r[5]:>6d} {r[6] / 1e3:>10.1f} {r[7] / 1e3:>8.2f} {r[8] / 1e3:>8.2f}")


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 0.0)
