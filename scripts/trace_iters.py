"""First iterations of the last render in a rocprofv3 --kernel-trace CSV: k_extend and k_shade durations side by side. k_shade streams
~124 B per ray at a steady rate, so its duration stands in for the number of rays of the iteration."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
gi = [i for i, r in enumerate(rows) if 'k_generate' in r['Kernel_Name']]
rs = rows[gi[-1]:]
def sel(k): return [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in rs if k in r['Kernel_Name']]
ext, shd = sel('k_extend'), sel('k_shade')
for i in range(min(int(sys.argv[2]) if len(sys.argv) > 2 else 14, len(shd))):
    print("iteration %2d  k_extend %9.1f us  k_shade %8.1f us  ratio %.1f" % (i, ext[i], shd[i], ext[i] / shd[i]))
