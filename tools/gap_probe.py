"""Development aid: idle gaps between kernels in a rocprofv3 --kernel-trace run (csv) of bench.py.
usage: gap_probe.py <trace dir> [min gap us]   Prints busy / idle time of the span between the
first and the last kernel named like an optimizer / LSTM step and the largest gaps with their
neighbours."""
import csv, glob, sys
d = sys.argv[1]
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
f = sorted(glob.glob(d + '/**/*kernel_trace.csv', recursive=True))[0]
rows = []
for r in csv.DictReader(open(f)):
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
rows.sort()
# the timed region: from the first band lattice launch to the last one
idx = [i for i, r in enumerate(rows) if 'lattice_fwbw_band' in r[2]]
lo, hi = idx[2], idx[-1]
span = rows[lo:hi + 1]
nsteps = len([1 for r in span if 'lattice_fwbw_band' in r[2]]) - 1
wall = span[-1][0] - span[0][0]
busy, cur_end, gaps = 0, span[0][0], []
for i, (s, e, n) in enumerate(span[:-1]):
    if s > cur_end:
        gaps.append((s - cur_end, span[i - 1][2][:60] if i else '', n[:60]))
    busy += max(0, e - max(s, cur_end))
    cur_end = max(cur_end, e)
print('steps %d  wall %.3f ms/step  busy %.3f ms/step  idle %.3f ms/step  kernels/step %.0f' % (
    nsteps, wall / 1e6 / nsteps, busy / 1e6 / nsteps, (wall - busy) / 1e6 / nsteps, len(span) / nsteps))
small = sum(g[0] for g in gaps if g[0] / 1e3 < thr)
print('gaps below %.0f us: %.3f ms/step (%d per step)' % (thr, small / 1e6 / nsteps, len([1 for g in gaps if g[0] / 1e3 < thr]) / nsteps))
for g in sorted(gaps, reverse=True)[:int(6 * nsteps)]:
    if g[0] / 1e3 >= thr:
        print('%8.1f us   after %-60s before %s' % (g[0] / 1e3, g[1], g[2]))
