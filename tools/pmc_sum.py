"""Sum rocprofv3 --pmc counter_collection.csv values per counter for kernels matching a pattern.
usage: python tools/pmc_sum.py <dir> [kernel-substring]"""
import csv, glob, sys, collections
d = sys.argv[1]; pat = sys.argv[2] if len(sys.argv) > 2 else 'k_fused'
acc = collections.defaultdict(float); nd = collections.defaultdict(set)
for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if pat in r['Kernel_Name']:
            acc[r['Counter_Name']] += float(r['Counter_Value'])
            nd[r['Counter_Name']].add(r['Dispatch_Id'])
for k in sorted(acc):
    n = max(1, len(nd[k]))
    print(f"{k:34s} {acc[k] / n:16.0f}  per dispatch ({n} dispatches)")
