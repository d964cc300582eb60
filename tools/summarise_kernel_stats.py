#!/usr/bin/env python3
"""Development aid: a rocprofv3 *kernel_stats.csv with readable kernel names."""
import csv
import sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Name']
    if 'rocprim' in n:
        short = 'rocprim:' + ('onesweep_iteration' if 'onesweep_iteration' in n else 'onesweep_histogram' if 'global_offsets' in n
                              else 'scan' if 'scan_impl' in n else 'other')
    else:
        short = n.split('(')[0].replace('void ', '').replace('msspe::', '')
        if '<' in n.split('(')[0]:
            short = n.split('(')[0].replace('void ', '').replace('msspe::', '')
        if short.strip() == '' or short.endswith('::'):
            short = n.replace('msspe::(anonymous namespace)::', '').split('(')[0].replace('void ', '')
    print(f"{short[:58]:58s} calls {r['Calls']:>6s} total {float(r['TotalDurationNs'])/1e6:9.3f} ms avg {float(r['AverageNs'])/1e3:10.2f} us max {float(r['MaxNs'])/1e3:9.1f}")
