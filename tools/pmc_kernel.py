"""print the counters of the kernels matching a name from a rocprofv3 --pmc results db: per dispatch averages"""
import glob, sqlite3, sys
f = glob.glob(sys.argv[1] + "/**/*.db", recursive=True)[0]
pick = sys.argv[2]
db = sqlite3.connect(f)
for k, c, v, n in db.execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection "
                             "group by kernel_name, counter_name"):
    if pick in k:
        print("%-40s %-28s per dispatch %16.0f  (%d dispatches)" % (k[:40], c, v / n, n))
