"""rocprofv3 --pmc CSVs -> JSON for a kernel: the dispatches with the largest grid (default), or the mean over ALL its
dispatches (mode `all`: the per-launch figure that pairs with a kernel's average duration in rocprofv3 --stats).
usage: pmc_dominant.py OUT.json KERNEL_SUBSTR EXEC_GFLOP FETCH_DIR WRITE_DIR MFMA_DIR [label] [largest|all]"""
import csv, glob, json, sys
out, sub, gflop, dF, dW, dM = sys.argv[1], sys.argv[2], float(sys.argv[3]), sys.argv[4], sys.argv[5], sys.argv[6]
label = sys.argv[7] if len(sys.argv) > 7 else ''
mode = sys.argv[8] if len(sys.argv) > 8 else 'largest'


def rows(d):
    r = []
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        r += [x for x in csv.DictReader(open(f)) if sub in x['Kernel_Name']]
    return r


def per_launch(d, counter):
    rs = [x for x in rows(d) if x['Counter_Name'] == counter]
    if not rs:
        return None, 0, 0
    g = max(int(x['Grid_Size']) for x in rs)
    sel = [float(x['Counter_Value']) for x in rs if mode == 'all' or int(x['Grid_Size']) == g]
    return sum(sel) / len(sel), len(sel), g


f, nf, grid = per_launch(dF, 'FETCH_SIZE')
w, nw, _ = per_launch(dW, 'WRITE_SIZE')
mf, _, _ = per_launch(dM, 'SQ_VALU_MFMA_BUSY_CYCLES')
ga, _, _ = per_launch(dM, 'GRBM_GUI_ACTIVE')
res = {'kernel': sub, 'kernel_family': 'igemm deep-K' if ('igemm_kernel<128, 128, 64, 64, 0, 0, 2, false>' in sub or 'igemm_h16_kernel' in sub) else sub.split('<')[0], 'what': label, 'launches': mode, 'lazy_finest': True, 'launches_averaged': nf, 'grid_size_threads': grid, 'executed_GFLOP': gflop,
       'FETCH_SIZE_KB_per_launch': f, 'WRITE_SIZE_KB_per_launch': w,
       'fetch_bytes_raw': f * 1024, 'fetch_bytes_corrected_x2': 2 * f * 1024, 'write_bytes': w * 1024,
       'traffic_bytes_per_launch': 2 * f * 1024 + w * 1024,
       'mfma_busy_cycles': mf, 'grbm_gui_active_sum8xcd': ga,
       'mfma_busy_frac': (mf / (1024 * ga / 8)) if mf and ga else None,
       'note': 'separate rocprofv3 --pmc passes (FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE) '
               'of the bench command; launches: the largest Grid_Size of the kernel, or all of them (`launches`); gfx950: FETCH_SIZE reports 1/2 of '
               'wide coalesced 16 B/lane reads -> doubled (MI355X_MICROARCH.md, HBM) and counts L2 -> fabric requests, i.e. '
               'Infinity-Cache hits too; WRITE_SIZE exact; mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE/8)'}
json.dump(res, open(out, 'w'), indent=1)
print(json.dumps(res))
