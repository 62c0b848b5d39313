"""Idle gaps of the main queue inside one steady-state step of a rocprofv3 kernel trace: python tools/timeline_gaps.py <csv> [occurrence]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
occ = int(sys.argv[2]) if len(sys.argv) > 2 else 5
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', '')) for r in rows)
mainq = collections.Counter(e[3] for e in ev if 'conv16_wino' in e[2]).most_common(1)[0][0]
ev = [e for e in ev if e[3] == mainq]
hits = [i for i, e in enumerate(ev) if 'conv_cin1' in e[2]]
i0, i1 = hits[occ], hits[occ + 1]
span = ev[i1][0] - ev[i0][0]
busy = sum(e[1] - e[0] for e in ev[i0:i1])
print(f'step {span / 1e3:.1f} us, kernels {busy / 1e3:.1f} us, idle {(span - busy) / 1e3:.1f} us, {i1 - i0} kernels')
prev = ev[i0][1]
for s, e, n, q in ev[i0 + 1:i1 + 1]:
    g = (s - prev) / 1e3
    if g > 1.0:
        print(f'  gap {g:7.1f} us before {n[:70]}')
    prev = e
