"""Diagnostics (GPU box): histogram of a rocprofv3 PC-sampling run.  pc_hist.py DIR OUT.csv
Aggregates every *pc_sampling*.csv under DIR by the columns that identify the instruction and its issue / stall state
(dropping timestamps, ids of waves / dispatches), so that only a few thousand rows travel back."""
import csv, collections, glob, os, sys
src, out = sys.argv[1], sys.argv[2]
files = [f for f in glob.glob(os.path.join(src, '**', '*.csv'), recursive=True) if 'pc_sampling' in os.path.basename(f)]
print('files:', files)
drop = ('timestamp', 'exec_mask', 'dispatch', 'correlation', 'wave_id', 'wave_in_group', 'chiplet', 'hw_id', 'workgroup', 'sample')
with open(out, 'w', newline='') as fo:
    w = csv.writer(fo)
    for f in files:
        hist, keys = collections.Counter(), None
        with open(f, newline='') as fi:
            rd = csv.DictReader(fi)
            keys = [k for k in rd.fieldnames if not any(d in k.lower() for d in drop)]
            first = []
            for row in rd:
                if len(first) < 3: first.append(dict(row))
                hist[tuple(row[k] for k in keys)] += 1
        w.writerow(['# file', os.path.basename(f), 'all columns', ' '.join(rd.fieldnames)])
        for r in first: w.writerow(['# example'] + [f'{k}={v}' for k, v in r.items()])
        w.writerow(['count'] + keys)
        for key, n in hist.most_common(6000): w.writerow([n] + list(key))
        print(os.path.basename(f), 'samples', sum(hist.values()), 'distinct', len(hist))
