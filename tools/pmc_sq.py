"""Average SQ counters per launch of one kernel from rocprofv3 --pmc csv passes: python tools/pmc_sq.py <kernel substr> <dir> [<dir> ...]"""
import collections, csv, glob, sys
sub = sys.argv[1]
for d in sys.argv[2:]:
    for f in glob.glob(f"{d}/*/*_counter_collection.csv"):
        agg = collections.defaultdict(lambda: [0, 0.0])
        for r in csv.DictReader(open(f)):
            if sub in r["Kernel_Name"]:
                agg[r["Counter_Name"]][0] += 1
                agg[r["Counter_Name"]][1] += float(r["Counter_Value"])
        for k, v in sorted(agg.items()):
            print(f"{k:32s} launches {v[0]:4d}  avg {v[1] / v[0]:16.1f}")
