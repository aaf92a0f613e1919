"""Summarise a rocprofv3 rocpd database (ROCm 7.2 default output) as a per-kernel stats table.

    python scripts/rocpd_summary.py gpurun_out/prof/x_results.db > profiles/rNN_name.txt
"""
import sqlite3
import sys


def main(path):
    con = sqlite3.connect(path)
    cur = con.cursor()
    rows = cur.execute(
        "select name, count(*), sum(end-start)/1e6, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3, "
        "max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(grid_x), max(workgroup_x) "
        "from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    print(f"# rocprofv3 --kernel-trace --stats summary of {path}")
    print(f"{'kernel':72s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>11s} {'min_us':>10s} {'max_us':>10s} {'%':>6s} "
          f"{'vgpr':>5s} {'agpr':>5s} {'sgpr':>5s} {'lds':>7s} {'grid_x':>10s} {'wg':>4s}")
    for r in rows:
        print(f"{r[0][:72]:72s} {r[1]:6d} {r[2]:10.3f} {r[3]:11.2f} {r[4]:10.2f} {r[5]:10.2f} {100 * r[2] / tot:6.2f} "
              f"{r[6]:5d} {r[7]:5d} {r[8]:5d} {r[9]:7d} {r[10]:10d} {r[11]:4d}")


if __name__ == "__main__":
    main(sys.argv[1])
