"""Per-kernel summary of a rocprofv3 --kernel-trace --stats run (csv output): prints the top
kernels and optionally writes the summary csv that is committed under profiles/."""
import csv
import glob
import sys


def main():
    d = sys.argv[1]
    steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    f = sorted(glob.glob(d + '/**/*kernel_stats.csv', recursive=True))[0]
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r['TotalDurationNs']) for r in rows)
    print('total kernel time %.2f ms (%.2f ms per step over %g steps)' % (tot / 1e6, tot / 1e6 / steps, steps))
    for r in rows[:int(sys.argv[4]) if len(sys.argv) > 4 else 40]:
        print('%-70s calls %5s  %8.3f ms/step  avg %9.1f us' % (
            r['Name'][:70], r['Calls'], float(r['TotalDurationNs']) / 1e6 / steps, float(r['AverageNs']) / 1e3))
    if len(sys.argv) > 3 and sys.argv[3] != '-':
        with open(sys.argv[3], 'w') as o:
            o.write(open(f).read())


if __name__ == '__main__':
    main()
