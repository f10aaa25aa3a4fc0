"""Condense rocprofv3 --pmc counter_collection.csv files into a small per-kernel table (mean per dispatch)."""
import csv, glob, sys, collections
out = collections.OrderedDict()
for d in sys.argv[2:]:
    for f in glob.glob(f'{d}/**/*_counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            k = (r['Kernel_Name'].split('(')[0][:60], r['Counter_Name'])
            out.setdefault(k, []).append(float(r['Counter_Value']))
with open(sys.argv[1], 'w') as fh:
    fh.write('kernel,counter,dispatches,mean_value,min_value,max_value\n')
    for (k, c), v in out.items():
        fh.write(f'"{k}",{c},{len(v)},{sum(v)/len(v):.1f},{min(v):.1f},{max(v):.1f}\n')
print(open(sys.argv[1]).read())
