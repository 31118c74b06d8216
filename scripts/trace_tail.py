"""Per-iteration view of a rocprofv3 --kernel-trace CSV of one render (last render in the file)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
gi = [i for i, r in enumerate(rows) if 'k_generate' in r['Kernel_Name']]
rs = rows[gi[-1]:]
t0 = int(rs[0]['Start_Timestamp'])
def sel(k): return [(int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - int(r['Start_Timestamp'])) for r in rs if k in r['Kernel_Name']]
ext, shd = sel('k_extend'), sel('k_shade')
wall = (int(rs[-1]['End_Timestamp']) - t0) / 1e6
print(len(ext), len(shd), 'total ext %.1f ms shade %.1f ms' % (sum(d for _, d in ext) / 1e6, sum(d for _, d in shd) / 1e6), 'wall %.1f ms' % wall)
for i in [i for i in list(range(0, 100, 10)) + list(range(100, len(ext), 25)) if i < min(len(ext), len(shd))]:
    print(i, 'start %.2f ms' % (ext[i][0] / 1e6), 'ext %.1f us' % (ext[i][1] / 1e3), 'shade %.1f us' % (shd[i][1] / 1e3))
allk = [(int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in rs]
gap = sum(max(0, allk[i + 1][0] - allk[i][1]) for i in range(len(allk) - 1))
print('sum of gaps %.2f ms' % (gap / 1e6))
