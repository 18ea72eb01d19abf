"""Where a training step's wall time goes, from a rocprofv3 kernel trace (kernel_trace.csv of tools/profile_bench.sh):
per step (delimited by the adam_kernel launches) the time covered by a persistent scan, the time covered only by other
kernels (exposed GEMMs / elementwise work), and the time with no kernel running at all (launch gaps, host-bound).
usage: python tools/timeline_gaps.py <kernel_trace.csv>"""
import csv, sys


def union(iv):
    iv = sorted(iv)
    out = []
    for a, b in iv:
        if out and a <= out[-1][1]:
            out[-1][1] = max(out[-1][1], b)
        else:
            out.append([a, b])
    return out


def length(iv):
    return sum(b - a for a, b in iv)


def subtract(a, b):        # parts of the intervals a (a union) not covered by b (a union)
    out = []
    j = 0
    for s, e in a:
        cur = s
        while j < len(b) and b[j][1] <= cur:
            j += 1
        k = j
        while k < len(b) and b[k][0] < e:
            if b[k][0] > cur:
                out.append([cur, b[k][0]])
            cur = max(cur, b[k][1])
            k += 1
        if cur < e:
            out.append([cur, e])
    return out


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows)
    adam = [e for e in ev if e[2].startswith('adam_kernel')]
    if len(adam) < 2:
        print('need at least two optimiser steps in the trace'); return
    for i in range(len(adam) - 1):
        t0, t1 = adam[i][1], adam[i + 1][1]
        step = [e for e in ev if e[0] >= t0 and e[1] <= t1]
        scan = union([(a, b) for a, b, n in step if 'scan' in n])
        allk = union([(a, b) for a, b, n in step])
        other_only = subtract(allk, scan)
        idle = subtract([[t0, t1]], allk)
        gemm_only = {}
        for a, b, n in step:
            if 'scan' in n:
                continue
            ex = length(subtract([[a, b]], scan))
            key = n.split('(')[0][:60]
            gemm_only[key] = gemm_only.get(key, 0) + ex
        print('step %d: wall %.2f ms | under a scan %.2f | other kernels only %.2f | no kernel %.2f (%d gaps, longest %.0f us)'
              % (i, (t1 - t0) / 1e6, length(scan) / 1e6, length(other_only) / 1e6, length(idle) / 1e6, len(idle),
                 max([b - a for a, b in idle] + [0]) / 1e3))
        top = sorted(gemm_only.items(), key=lambda kv: -kv[1])[:8]
        print('   exposed (not under a scan; overlapping kernels counted each): ' +
              ', '.join('%s %.2f ms' % (k, v / 1e6) for k, v in top))


if __name__ == '__main__':
    main()
