#!/usr/bin/env python3
"""Lists, for every kernel of a gfx950 assembly file (hipcc -S --cuda-device-only), its loops with the global loads issued
inside and the `s_waitcnt vmcnt(N)` values the compiler placed there.  A software-pipelined loop whose waits read vmcnt(0)
(or fewer than the loads of the younger register set) waits for the loads it has just requested: the compiler's counts are
static, so one path on which the younger set was not requested (a load under `if (kt + 2 < nk)`, a different load order in
the loop's preheader) makes every iteration wait for everything in flight.

  hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -o /tmp/k.s csrc/gemm.hip && tools/isa_loop_waits.py /tmp/k.s [filter]"""
import re
import subprocess
import sys


def demangle(names):
    try:
        out = subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-cxxfilt'], input='\n'.join(names), capture_output=True, text=True).stdout
        return out.splitlines()
    except OSError:
        return names


def main():
    path = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ''
    kernels, cur = [], None
    for line in open(path):
        m = re.match(r'^(_Z\w+):', line)
        if m:
            cur = [m.group(1), []]
            kernels.append(cur)
            continue
        if cur is not None:
            if line.startswith('\t.end_amdhsa_kernel') or '.Lfunc_end' in line:
                cur = None
                continue
            cur[1].append(line.rstrip())
    names = demangle([k[0] for k in kernels])
    for (mangled, body), name in zip(kernels, names):
        if flt and flt not in name:
            continue
        # blocks by label; a block belongs to a loop when its label comment says "in Loop" or "Loop Header"
        blocks, lab, in_loop = [], None, False
        for ln in body:
            m = re.match(r'^(\.LBB\d+_\d+):(.*)', ln)
            if m:
                lab = m.group(1)
                in_loop = 'Loop' in m.group(2)
                blocks.append([lab, in_loop, m.group(2).strip(), []])
            elif re.match(r'^; %bb\.\d+:(.*)', ln):
                m2 = re.match(r'^; (%bb\.\d+):(.*)', ln)
                in_loop = 'Loop' in m2.group(2)
                blocks.append([m2.group(1), in_loop, m2.group(2).strip(), []])
            elif blocks:
                blocks[-1][3].append(ln)
        rows = []
        for lab, in_loop, note, ins in blocks:
            if not in_loop:
                continue
            ev = []
            for i in ins:
                s = i.strip()
                if s.startswith(('global_load', 'buffer_load')):
                    ev.append('L')
                elif s.startswith('s_waitcnt') and 'vmcnt' in s:
                    ev.append('w' + re.search(r'vmcnt\((\d+)\)', s).group(1))
                elif s.startswith('s_barrier'):
                    ev.append('|')
                elif s.startswith('v_mfma'):
                    if not ev or not ev[-1].startswith('M'):
                        ev.append('M1')
                    else:
                        ev[-1] = 'M%d' % (int(ev[-1][1:]) + 1)
            if any(e == 'L' or e.startswith('w') for e in ev):
                rows.append(f'   {lab:12s} {" ".join(ev)}   ; {note[:60]}')
        if rows:
            print(name[:160])
            print('\n'.join(rows))


if __name__ == '__main__':
    main()
