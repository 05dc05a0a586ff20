"""Summarise a rocprofv3 --pmc csv run: mean counter value per kernel per dispatch."""
import csv, glob, sys, collections, re
d = sys.argv[1]
files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in files:
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        m = re.search(r"(drone_\w+|car_\w+|hopper_\w+|rs_\w+|sum_partials_kernel)(<\d+>)?", k)
        if not m:
            continue
        k = m.group(0)
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print("   %-28s n=%3d mean=%.6g" % (c, len(v), sum(v) / len(v)))
