"""Print per-kernel averages of the PMC counters in a rocprofv3 rocpd database (compact, one line per counter)."""
import glob
import sqlite3
import sys

path = sys.argv[1]
like = sys.argv[2] if len(sys.argv) > 2 else "%pw_%"      # SQL LIKE filter on the kernel name ("%" = every kernel)
dbs = glob.glob(path + "/**/*.db", recursive=True) if not path.endswith(".db") else [path]
for db in dbs:
    con = sqlite3.connect(db)
    q = ("select kernel_name, grid_size_x, counter_name, count(*), avg(value) from counters_collection "
         "where kernel_name like '{0}' group by kernel_name, grid_size_x, counter_name order by kernel_name, grid_size_x, counter_name").format(like)
    last = None
    for name, grid, ctr, n, avg in con.execute(q):
        short = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]
        key = (short, grid)
        if key != last:
            print("== {} grid={} dispatches={}".format(short, grid, n))
            last = key
        print("   {:32s} {:16.0f}".format(ctr, avg))
