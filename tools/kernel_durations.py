import csv, glob, sys, collections
f = glob.glob('gpurun_out/kt/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
by = collections.defaultdict(list)
for r in rows:
    by[r['Kernel_Name'][:70]].append((int(r['Start_Timestamp']), int(r['End_Timestamp'])))
for k, v in by.items():
    if 'tree5r_kernel<0>' in k or 'tree5r_kernel<2>' in k or 'ntt_tile12_kernel<3' in k:
        v.sort()
        d = [(e - s) / 1e3 for s, e in v]
        print(k, len(d))
        print('  ', ' '.join(f'{x:.0f}' for x in d))
