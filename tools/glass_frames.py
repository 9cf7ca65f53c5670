"""Kernel durations of each level-by-level frame in a rocprofv3 kernel trace (directory as argument)."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/*/*_kernel_trace.csv')[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
frames, cur = [], None
for r in rows:
    n = r['Kernel_Name']
    if 'queue_primary' in n or 'wf_primary' in n:
        cur = []
        frames.append(cur)
    if cur is not None and 'rtk::' in n:
        cur.append((n.split('(')[0].split('::')[-1][:14], (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6))
for fr in frames:
    print(' | '.join(f"{n} {t:.3f}" for n, t in fr if t > 0.01), ' sum', round(sum(t for _, t in fr), 3))
