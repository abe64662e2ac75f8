#!/usr/bin/env python3
"""Print the kws:: rows of a rocprofv3 kernel_stats CSV found under a directory."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'kws::' in r['Name']:
        print(r['Name'][:72].ljust(74), r['Calls'].rjust(5), f"{float(r['TotalDurationNs'])/1e6:9.2f} ms", f"{float(r['AverageNs'])/1e3:9.1f} us avg", r['Percentage'])
