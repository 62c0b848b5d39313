"""Instruction mix of a kernel's hottest loop (the longest backward-branch span) from `hipcc -S --cuda-device-only` output.
usage: python tools/isa_count.py file.s <mangled kernel name substring> [steps per loop body]"""
import collections, re, sys
s = open(sys.argv[1]).read()
name = sys.argv[2]
div = int(sys.argv[3]) if len(sys.argv) > 3 else 1
a = s.index(name + ':') if (name + ':') in s else s.index(re.search(r'^(\S*%s\S*):' % re.escape(name), s, re.M).group(1) + ':')
k = s[a:s.index('.Lfunc_end', a)].splitlines()
labels = {m.group(1): i for i, l in enumerate(k) for m in [re.match(r'^(\.LBB\d+_\d+):', l)] if m}
loops = []
for i, l in enumerate(k):
    m = re.search(r's_c?branch\w*\s+(\.LBB\d+_\d+)', l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        loops.append((labels[m.group(1)], i))
lo, hi = max(loops, key=lambda x: x[1] - x[0])
c = collections.Counter()
for l in k[lo:hi]:
    l = l.strip()
    if not l or l.startswith(('.', ';', '//')) or l.endswith(':'):
        continue
    c[l.split()[0]] += 1
print('loop', lo, hi, 'instructions', sum(c.values()), 'per step', sum(c.values()) / div)
for op, n in c.most_common(30):
    print(f'  {op:30s}{n / div:8.1f}')
