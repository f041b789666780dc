"""dev aid: kernel sequence of the last DBSCAN call in a rocprofv3 --kernel-trace CSV
(python tools/kernel_timeline.py gpurun_out/<dir>/<name>_kernel_trace.csv)"""
import csv
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'].split('(')[0].replace('pyqsm::', '') for r in rows]
first = sys.argv[2] if len(sys.argv) > 2 else 'k_bbox'
last = len(names) - 1 - names[::-1].index(first)
t0 = int(rows[last]['Start_Timestamp'])
for r, nm in list(zip(rows, names))[last:last + int(sys.argv[3]) if len(sys.argv) > 3 else last + 45]:
    print('%-30s start %8.1f us  duration %7.1f us' % (nm[:30], (int(r['Start_Timestamp']) - t0) / 1e3,
                                                      (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3))
