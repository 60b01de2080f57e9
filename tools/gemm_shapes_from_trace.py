"""Per-launch-shape GEMM durations from a kernel trace (last step only)."""
import csv
import sys
from collections import defaultdict

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append(r)
rows.sort(key=lambda r: int(r['Start_Timestamp']))
ends = [int(r['End_Timestamp']) for r in rows if 'sgdm_kernel' in r['Kernel_Name']]
t0, t1 = ends[-2], ends[-1]
agg = defaultdict(list)
for r in rows:
    s = int(r['Start_Timestamp'])
    if s < t0 or s > t1:
        continue
    n = r['Kernel_Name']
    if 'gemm_bf16_kernel' in n or 'splitk' in n:
        key = (n[:n.index('(')][-24:], int(r['Grid_Size_X']) // 256, int(r['Grid_Size_Y']), int(r['Grid_Size_Z']))
        agg[key].append((int(r['End_Timestamp']) - s) / 1e3)
tot = 0
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    tot += sum(v)
    print(k, len(v), 'avg %.1f us total %.0f us' % (sum(v) / len(v), sum(v)))
print('total %.0f us' % tot)
