"""Summarise rocprofv3 (ROCm 7.2 rocpd sqlite) outputs into text that can be committed under profiles/.

    python tools/rocpd_summary.py gpurun_out/prof_r01/trace_results.db                   # --kernel-trace --stats
    python tools/rocpd_summary.py --pmc gpurun_out/pmc_fetch_r01/pmc_results.db          # --pmc passes
"""
import sqlite3
import sys


def short(name):
    name = name.replace("void ", "")
    return name.split("(")[0][:60]


def kernel_stats(db):
    cur = sqlite3.connect(db).cursor()
    rows = list(cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
    print("# rocprofv3 --kernel-trace --stats : %s" % db)
    print("%-62s %8s %14s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
    for n, c, t, a, p in rows:
        print("%-62s %8d %14.3f %12.3f %7.2f" % (short(n), c, t, a, p))


def pmc(db):
    cur = sqlite3.connect(db).cursor()
    q = ("select kernel_name, counter_name, count(*), avg(value), min(value), max(value), avg(duration) "
         "from counters_collection group by kernel_name, counter_name order by avg(value) desc")
    print("# rocprofv3 --pmc : %s   (FETCH_SIZE / WRITE_SIZE are in KiB per dispatch)" % db)
    print("%-62s %-12s %6s %14s %14s %14s %10s" % ("kernel", "counter", "n", "avg", "min", "max", "avg_ns"))
    for k, c, n, a, lo, hi, d in cur.execute(q):
        print("%-62s %-12s %6d %14.3f %14.3f %14.3f %10.0f" % (short(k), c, n, a, lo, hi, d))


if __name__ == "__main__":
    if sys.argv[1] == "--pmc":
        for db in sys.argv[2:]:
            pmc(db)
    else:
        for db in sys.argv[1:]:
            kernel_stats(db)
