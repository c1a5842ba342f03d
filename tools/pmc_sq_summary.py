#!/usr/bin/env python3
"""Per kernel name: mean of every counter of a rocprofv3 --pmc counter_collection.csv, and each as a share of SQ_WAVE_CYCLES.

usage: pmc_sq_summary.py counter_collection.csv [substring of kernel name ...]"""
import csv
import sys
from collections import defaultdict


def main():
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    with open(sys.argv[1]) as f:
        for row in csv.DictReader(f):
            a = acc[row['Kernel_Name']][row['Counter_Name']]
            a[0] += float(row['Counter_Value'])
            a[1] += 1
    pats = sys.argv[2:]
    for name, counters in acc.items():
        if pats and not any(p in name for p in pats):
            continue
        mean = {k: v[0] / v[1] for k, v in counters.items()}
        wc = mean.get('SQ_WAVE_CYCLES', 0.0)
        n = next(iter(counters.values()))[1]
        print(f'{name[:100]}  ({n} dispatches)')
        for k in sorted(mean):
            share = f'{100.0 * mean[k] / wc:6.1f} % of wave cycles' if wc else ''
            print(f'    {k:28s} {mean[k]:16.0f} {share}')


if __name__ == '__main__':
    main()
