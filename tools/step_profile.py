"""Per-launch durations of one graph-replayed step from a rocprofv3 kernel trace csv (local helper)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'select_normalize' in r['Kernel_Name']]
a, b = idx[-4], idx[-3]
step = rows[a:b]
print("step wall %.1f us, launches %d" % ((int(rows[b]['Start_Timestamp']) - int(step[0]['Start_Timestamp'])) / 1e3, len(step)))
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 18
for r in step:
    n = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    if d > thr:
        print("%-52s %7.1f us  grid %s x %s" % (n[:52], d, r['Grid_Size_X'], r['Grid_Size_Y']))
