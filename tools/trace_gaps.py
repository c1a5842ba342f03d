#!/usr/bin/env python3
"""GPU busy share from a rocprofv3 --kernel-trace CSV: union of the kernel intervals against the span they cover, for the
last `frac` of the trace (the timed steps), plus the launch count.

usage: trace_gaps.py kernel_trace.csv [frac]"""
import csv
import sys


def main():
    frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
    iv = []
    with open(sys.argv[1]) as f:
        for r in csv.DictReader(f):
            iv.append((int(r['Start_Timestamp']), int(r['End_Timestamp'])))
    iv.sort()
    t0, t1 = iv[0][0], max(e for _, e in iv)
    cut = t1 - (t1 - t0) * frac
    iv = [(s, e) for s, e in iv if s >= cut]
    busy, cur_s, cur_e = 0, iv[0][0], iv[0][1]
    for s, e in iv[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    span = max(e for _, e in iv) - iv[0][0]
    print(f'{len(iv)} launches over {span / 1e6:.2f} ms: GPU busy {busy / 1e6:.2f} ms = {100.0 * busy / span:.1f} %, '
          f'sum of kernel times {sum(e - s for s, e in iv) / 1e6:.2f} ms')


if __name__ == '__main__':
    main()
