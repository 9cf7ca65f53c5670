"""Durations of the launches of the last level-by-level frame in a rocprofv3 kernel trace (directory as argument)."""
import csv,sys,glob
f=glob.glob(sys.argv[1]+'/*/*_kernel_trace.csv')[0]
rows=list(csv.DictReader(open(f)))
# last frame: from last wf_primary to end
idx=max(i for i,r in enumerate(rows) if 'wf_primary_kernel' in r['Kernel_Name'] or 'queue_primary_kernel' in r['Kernel_Name'])
for r in rows[idx-1:idx+12]:
    print(r['Kernel_Name'][:40].ljust(41), f"{(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6:.3f} ms")
