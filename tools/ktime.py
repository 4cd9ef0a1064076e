import csv,glob,sys
for f in glob.glob(sys.argv[1]+'/*/*kernel_stats.csv'):
    for r in csv.DictReader(open(f)):
        if 'sl_kernel' in r['Name'] or 'mitm' in r['Name']: print('   ', r['Name'][28:60], r['Calls'], float(r['AverageNs'])/1000, 'us')
