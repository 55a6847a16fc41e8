"""dev probe: the kernel timeline of the last call in a rocprofv3 kernel trace (trace_show.py DIR [first kernel of a call])"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
first = sys.argv[2] if len(sys.argv) > 2 else "k_aggregate_raw_d"
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith(first)]
i0 = max(0, idx[-1] - 4)
t0 = int(rows[idx[-1]]['Start_Timestamp'])
for r in rows[i0:]:
    print(r['Kernel_Name'].split('(')[0][:38].ljust(38), r['Queue_Id'], str(round((int(r['Start_Timestamp']) - t0) / 1e3)).rjust(6), str(round((int(r['End_Timestamp']) - t0) / 1e3)).rjust(6), r['Grid_Size_X'])
