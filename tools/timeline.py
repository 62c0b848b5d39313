"""One steady-state step of a rocprofv3 kernel trace as a timeline: start (us, relative), duration, overlap marker, grid, kernel.
usage: python tools/timeline.py <t_kernel_trace.csv> [anchor substring] [which occurrence] [how many rows]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
anchor = sys.argv[2] if len(sys.argv) > 2 else 'conv_cin1'
occ = int(sys.argv[3]) if len(sys.argv) > 3 else 4
nrows = int(sys.argv[4]) if len(sys.argv) > 4 else 120
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Grid_Size', r.get('Grid_Size_X', '')),
             r.get('Queue_Id', '')) for r in rows)
hits = [i for i, e in enumerate(ev) if anchor in e[2]]
i0 = hits[min(occ, len(hits) - 1)]
t0 = ev[i0][0]
prev_end = t0
for s, e, n, g, q in ev[i0:i0 + nrows]:
    mark = '||' if s < prev_end else f'+{(s - prev_end) / 1e3:.1f}'
    print(f'{(s - t0) / 1e3:9.1f} us  {(e - s) / 1e3:8.1f} us  {mark:>8}  q{q:>3} grid {g:>8}  {n[:90]}')
    prev_end = max(prev_end, e)
