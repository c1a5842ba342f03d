#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: per kernel (name, grid size) the mean counter value per dispatch.

FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB.  On gfx950 FETCH_SIZE tallies 128-byte read requests as 64
bytes for wide coalesced reads, so read bytes = 2 x FETCH_SIZE; WRITE_SIZE is exact for 16-byte-per-lane stores.

usage: pmc_summary.py FETCH.csv WRITE.csv [substring of kernel name]
"""
import csv
import json
import sys
from collections import defaultdict


def load(path):
    acc = defaultdict(lambda: [0.0, 0])
    with open(path) as f:
        for row in csv.DictReader(f):
            key = (row['Kernel_Name'], int(row['Grid_Size']), int(row['Workgroup_Size']))
            a = acc[key]
            a[0] += float(row['Counter_Value'])
            a[1] += 1
    return {k: (v[0] / v[1], v[1]) for k, v in acc.items()}


def main():
    fetch, write = load(sys.argv[1]), load(sys.argv[2])
    pat = sys.argv[3] if len(sys.argv) > 3 else ''
    out = []
    for key in sorted(fetch, key=lambda k: -fetch[k][0] * fetch[k][1]):
        if pat not in key[0] or key not in write:
            continue
        rd, wr = 2.0 * fetch[key][0] * 1024.0, write[key][0] * 1024.0
        out.append({'kernel': key[0][:90], 'grid': key[1], 'wg': key[2], 'dispatches': fetch[key][1],
                    'read_bytes': rd, 'write_bytes': wr, 'hbm_bytes': rd + wr})
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
