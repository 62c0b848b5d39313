"""GPU busy/idle analysis of a rocprofv3 kernel trace (+ memory-copy trace if present)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:50]) for r in rows)
# restrict to the last 60 % of the trace (steady state)
t0, t1 = ev[0][0], ev[-1][1]
lo = t0 + int(0.4 * (t1 - t0))
ev = [e for e in ev if e[0] >= lo]
busy, cur_end, gaps = 0, ev[0][0], []
for s, e, n in ev:
    if s > cur_end:
        gaps.append((s - cur_end, n))
        cur_end = s
    if e > cur_end:
        busy += e - cur_end
        cur_end = e
span = ev[-1][1] - ev[0][0]
print(f'span {span/1e6:.2f} ms busy {busy/1e6:.2f} ms ({100*busy/span:.1f}%) idle {(span-busy)/1e6:.2f} ms in {len(gaps)} gaps')
gaps.sort(reverse=True)
import collections
agg = collections.Counter()
for g, n in gaps:
    agg[n] += g
print('idle time before kernel (top):')
for n, g in agg.most_common(12):
    print(f'  {g/1e6:8.3f} ms  before {n}')
print('largest gaps (us):', [round(g/1e3) for g, _ in gaps[:15]])
