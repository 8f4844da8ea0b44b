"""Aggregate a rocprofv3 --pmc counter_collection CSV per kernel: python tools/pmc_summary.py in.csv out.csv"""
import csv, sys, collections
rows = csv.DictReader(open(sys.argv[1]))
agg = collections.OrderedDict()
disp = collections.defaultdict(set)
for r in rows:
    k = r['Kernel_Name']
    c = r['Counter_Name']
    agg.setdefault(k, collections.OrderedDict()).setdefault(c, 0.0)
    agg[k][c] += float(r['Counter_Value'])
    disp[k].add(r['Dispatch_Id'])
cs = sorted({c for v in agg.values() for c in v})
w = csv.writer(open(sys.argv[2], 'w'))
w.writerow(['kernel', 'dispatches'] + cs)
for k, v in agg.items():
    w.writerow([k, len(disp[k])] + ['%.0f' % v.get(c, 0) for c in cs])
