import csv, glob, sys, collections
d = sys.argv[1]
rows = collections.defaultdict(list)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "resize_sym" in r["Kernel_Name"]:
            rows[r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
for name, v in sorted(rows.items()):
    v.sort(); last = [x[1] for x in v[-20:]]
    g = [sum(last[i:i + 5]) / 5 for i in range(0, 20, 5)]
    print("%-36s slow %.4g  fast %.4g  slow %.4g  fast %.4g   slow/fast %.3f" % (name, g[0], g[1], g[2], g[3], (g[0] + g[2]) / max(1e-9, g[1] + g[3])))
