"""GPU occupancy of a traced run: python tools/trace_busy.py <kernel_trace.csv> [skip_frac]
Union of the kernel intervals (all queues), per-queue busy time and the idle gaps, over the middle of the trace (the first / last `skip_frac` of
the time span are dropped: warm-up, pipeline fill and drain).  Says whether a pipeline is bound by the GPU or by what feeds it."""
import csv, sys
import numpy as np
rows = list(csv.DictReader(open(sys.argv[1])))
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.25
s = np.array([int(r['Start_Timestamp']) for r in rows], np.int64)
e = np.array([int(r['End_Timestamp']) for r in rows], np.int64)
q = np.array([r.get('Queue_Id', '0') for r in rows])
names = np.array([r['Kernel_Name'][:60] for r in rows])
own = np.flatnonzero(np.char.find(names, 'pcc') >= 0)          # the library's kernels mark the timed region: its middle, by launch index
t0, t1 = s[own[int(skip * len(own))]], e[own[int((1 - skip) * len(own)) - 1]]
lo, hi = t0, t1
m = (s >= lo) & (e <= hi)
s, e, q, names = s[m], e[m], q[m], names[m]
order = np.argsort(s)
s, e, q, names = s[order], e[order], q[order], names[order]
span = e.max() - s.min()
busy, cur_s, cur_e, gaps = 0, s[0], e[0], []
for a, b, n in zip(s[1:], e[1:], names[1:]):
    if a > cur_e:
        busy += cur_e - cur_s
        gaps.append((a - cur_e, n))
        cur_s, cur_e = a, b
    else:
        cur_e = max(cur_e, b)
busy += cur_e - cur_s
print(f'window {span / 1e6:.2f} ms, {len(s)} kernels; some kernel running {100.0 * busy / span:.1f} % of it; sum of kernel time {100.0 * (e - s).sum() / span:.1f} %')
for qq in sorted(set(q)):
    k = q == qq
    print(f'  queue {qq}: {k.sum()} kernels, busy {100.0 * (e[k] - s[k]).sum() / span:.1f} %')
g = np.array([x[0] for x in gaps])
if len(g):
    print(f'idle gaps: {len(g)}, total {g.sum() / 1e6:.2f} ms; > 20 us: {(g > 20000).sum()} totalling {g[g > 20000].sum() / 1e6:.2f} ms; > 100 us: {(g > 100000).sum()} totalling {g[g > 100000].sum() / 1e6:.2f} ms')
    big = sorted(gaps, key=lambda x: -x[0])[:8]
    print('largest gaps (us, kernel that ended the gap):', [(round(a / 1e3), n[:40]) for a, n in big])
