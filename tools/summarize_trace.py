"""Summarise a rocprofv3 --kernel-trace CSV per (kernel, grid size): calls, total, average, MEDIAN, trimmed mean, max.

    python tools/summarize_trace.py <kernel_trace.csv> <steps the traced command ran> [rows]

The average of a traced run is not a launch time: the first launch of a kernel pays its code-object load (round 4: one 28.8 ms launch
among 64 of 0.36 ms doubled the average of the dominant kernel).  The median / the mean without the top and bottom 10 % are what a
launch costs in the steady state; a row whose max exceeds 10 x its median is flagged.  `steps` is every step the traced command ran
(set-up priming + warm-up + timed)."""
import collections, csv, sys


def load(path):
    agg = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        n = r['Kernel_Name'].replace('void (anonymous namespace)::', '').replace('(anonymous namespace)::', '')
        key = (n[:72], r.get('Grid_Size_X', r.get('Grid_Size', '')))
        agg.setdefault(key, []).append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
    return agg


def stats(d):
    d = sorted(d)
    n = len(d)
    cut = n // 10
    core = d[cut:n - cut] if n - 2 * cut > 0 else d
    return dict(calls=n, total=sum(d), avg=sum(d) / n, median=d[n // 2] if n % 2 else 0.5 * (d[n // 2 - 1] + d[n // 2]),
                trimmed=sum(core) / len(core), max=d[-1], outlier=d[-1] > 10 * d[n // 2])


def table(agg, steps, rows=32):
    st = {k: stats(v) for k, v in agg.items()}
    tot = sum(s['total'] for s in st.values())
    out = ['| calls | total ms | avg us | median us | trimmed-mean us | max us | % | grid | kernel |', '|---|---|---|---|---|---|---|---|---|']
    for k, s in sorted(st.items(), key=lambda kv: -kv[1]['total'])[:rows]:
        flag = ' **(max > 10 x median: first-launch / outlier in the average)**' if s['outlier'] else ''
        out.append(f"| {s['calls']} | {s['total']/1e6:.2f} | {s['avg']/1e3:.1f} | {s['median']/1e3:.1f} | {s['trimmed']/1e3:.1f} | {s['max']/1e3:.1f} | "
                   f"{100*s['total']/tot:.1f} | {k[1]} | `{k[0]}`{flag} |")
    steady = sum(s['median'] * s['calls'] for s in st.values())
    out.append('')
    out.append(f'total kernel time {tot/1e6:.2f} ms over {steps} steps = {tot/1e6/steps:.2f} ms/step including first launches; '
               f'sum of (median x calls) = {steady/1e6:.2f} ms = **{steady/1e6/steps:.2f} ms/step** in the steady state '
               '(kernels on side streams overlap, so this is an upper bound of the step time)')
    return '\n'.join(out), st


if __name__ == '__main__':
    path, steps = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 1
    print(table(load(path), steps, int(sys.argv[3]) if len(sys.argv) > 3 else 32)[0])
