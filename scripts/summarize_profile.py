"""Condense a scripts/profile_gpu.sh output directory (gpurun_out/prof_<tag>/) into the two small files that are kept
under profiles/: the kernel-trace statistics with the template-heavy PyTorch kernel names shortened, and the PMC
counters of the dominant kernel averaged over its launches (FETCH_SIZE doubled, MI355X_MICROARCH.md gfx950 note).
usage: python scripts/summarize_profile.py gpurun_out/prof_r02 profiles/r02 [kernel substring] [forwards per launch]"""
import csv
import glob
import json
import os
import sys

src, dst = sys.argv[1], sys.argv[2]
kernel = sys.argv[3] if len(sys.argv) > 3 else 'sample_chain_fc_kernel'
forwards = int(sys.argv[4]) if len(sys.argv) > 4 else 501
tag = f'T{forwards - 1}'
os.makedirs(dst, exist_ok=True)

stats = glob.glob(os.path.join(src, 'trace_*_kernel_stats.csv'))[0]
rows = list(csv.reader(open(stats)))
with open(os.path.join(dst, f'kernel_stats_{tag}.csv'), 'w', newline='') as f:
    w = csv.writer(f)
    for r in rows:
        r[0] = r[0] if len(r[0]) < 140 else r[0][:100] + ' ... ' + r[0][-30:]
        w.writerow(r)
# a chain may be TWO launches (EDM.split_chain: <.., false, ..> for every molecule, then <.., true, ..> for the big ones on teams):
# per-chain figures = totals over both kernels / number of chains (= launches of the kernel with the largest total)
match = [r for r in rows[1:] if kernel in r[0]]
chains = int(max(match, key=lambda r: float(r[2]))[1])
avg_ns = sum(float(r[2]) for r in match) / chains

sums, counts, meta = {}, {}, {}
for path in glob.glob(os.path.join(src, 'pmc_*_counter_collection.csv')):
    for r in csv.DictReader(open(path)):
        if kernel not in r['Kernel_Name']:
            continue
        sums[r['Counter_Name']] = sums.get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
        counts[r['Counter_Name']] = counts.get(r['Counter_Name'], 0) + 1
        meta = {k: r[k] for k in ('Grid_Size', 'Workgroup_Size', 'LDS_Block_Size', 'Scratch_Size', 'VGPR_Count',
                                  'Accum_VGPR_Count', 'SGPR_Count')}
out = {k: sums[k] / chains for k in sorted(sums)}       # per chain (see above)
out['_kernel_meta'] = meta
d = {'kernel_avg_ms_rocprof': avg_ns / 1e6, 'forwards_per_launch': forwards, 'launches_per_chain': {r[0][:90]: int(r[1]) / chains for r in match}}
if 'GRBM_GUI_ACTIVE' in out:
    d['gui_active_per_xcd_cycles'] = out['GRBM_GUI_ACTIVE'] / 8
    d['effective_clock_GHz'] = d['gui_active_per_xcd_cycles'] / avg_ns
if 'SQ_VALU_MFMA_BUSY_CYCLES' in out and 'GRBM_GUI_ACTIVE' in out:
    d['MfmaUtil_pct'] = 100.0 * out['SQ_VALU_MFMA_BUSY_CYCLES'] / (d['gui_active_per_xcd_cycles'] * 256 * 4)
if 'FETCH_SIZE' in out:
    d['hbm_fetch_bytes_per_launch_x2_corrected'] = out['FETCH_SIZE'] * 1024 * 2
if 'WRITE_SIZE' in out:
    d['hbm_write_bytes_per_launch'] = out['WRITE_SIZE'] * 1024
d['note'] = ('per launch of ' + kernel + ' (' + tag + ': ' + str(forwards) + ' forwards, C2 ragged batch, f16x3); FETCH_SIZE/WRITE_SIZE are in KiB, '
             'FETCH doubled per MI355X_MICROARCH.md (gfx950 counts 128-B requests as 64 B); fabric-side counters, '
             'Infinity-Cache hits included: the fp32 node-feature tiles a workgroup parks in its HBM scratch and the L2 misses of the weight stream; the algorithmic '
             'bytes are ~2 MB per forward')
out['_derived'] = d
json.dump(out, open(os.path.join(dst, f'pmc_chain_kernel_{tag}.json'), 'w'), indent=1)
print(json.dumps(d, indent=1))
