"""Summarise a rocprofv3 --kernel-trace CSV per (kernel, grid size): calls, total, average."""
import collections, csv, sys
path, steps = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 1
rows = list(csv.DictReader(open(path)))
agg = collections.OrderedDict()
for r in rows:
    n = r['Kernel_Name'].replace('void (anonymous namespace)::', '').replace('(anonymous namespace)::', '')
    key = (n[:72], r.get('Grid_Size_X', r.get('Grid_Size', '')))
    d = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    a = agg.setdefault(key, [0, 0]); a[0] += 1; a[1] += d
tot = sum(v[1] for v in agg.values())
print(f'| calls | total ms | avg us | % | grid | kernel |\n|---|---|---|---|---|---|')
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[3]) if len(sys.argv) > 3 else 32]:
    print(f'| {v[0]} | {v[1]/1e6:.2f} | {v[1]/v[0]/1e3:.1f} | {100*v[1]/tot:.1f} | {k[1]} | `{k[0]}` |')
print(f'\ntotal kernel time {tot/1e6:.2f} ms over {steps} steps = {tot/1e6/steps:.2f} ms/step')
