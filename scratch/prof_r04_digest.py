# digest of scratch/prof_r04.sh's rocprofv3 output -> summary_rocprofv3.txt, traffic.json, linalg_pmc.json (run before the raw
# CSVs are deleted).  Rules (MI355X_MICROARCH.md): FETCH_SIZE / WRITE_SIZE are KiB per dispatch, FETCH_SIZE is doubled on
# gfx950; SQ_INSTS_VALU_MFMA_MOPS_F64 counts matrix operations in units of 512 flops.
import csv, glob, collections, os, json, sys
O = sys.argv[1]
out = open(os.path.join(O, 'summary_rocprofv3.txt'), 'w')
def P(*a):
    s = ' '.join(str(x) for x in a); print(s); out.write(s + '\n')
dur = {}
for tag in ('trace_c3', 'trace_c2', 'trace_c4', 'trace_c5', 'trace_c3_f64', 'trace_linalg', 'trace_grad_c3', 'trace_grad_c2', 'trace_grad_c5',
            'trace_grad_c4'):
    for f in glob.glob(O + '/' + tag + '/*/*kernel_stats.csv') + glob.glob(O + '/' + tag + '_kernel_stats.csv'):
        P('==', tag, '(rocprofv3 --kernel-trace --stats)')
        for r in list(csv.DictReader(open(f)))[:(24 if 'grad' in tag else 12)]:
            P('  %-70s calls %4s avg %10.1f us  %6s%%' % (r['Name'][:70], r['Calls'], float(r['AverageNs']) / 1e3, r['Percentage']))
            dur[(tag, r['Name'].split('(')[0][:60])] = float(r['AverageNs']) / 1e3
        if os.path.abspath(f) != os.path.abspath(os.path.join(O, tag + '_kernel_stats.csv')):
            os.replace(f, os.path.join(O, tag + '_kernel_stats.csv'))
# the pass kernel of stage B runs four times per step under one name: the two Psi2 passes and the two short Psi1 passes — told apart
# by their grids (per-dispatch rows of the kernel trace / the counter collection)
P('== pg_pass_kernel by dispatch (kernel trace, per-dispatch rows): grid size -> calls, average us')
for tag in ('trace_grad_c3', 'trace_grad_c2', 'trace_grad_c5', 'trace_grad_c4'):
    by = collections.defaultdict(list)
    for f in glob.glob(O + '/' + tag + '/*/*kernel_trace.csv'):
        for r in csv.DictReader(open(f)):
            if 'pg_pass_kernel' in r.get('Kernel_Name', ''):
                by[r.get('Grid_Size', r.get('Grid_Size_X', '?'))].append((float(r['End_Timestamp']) - float(r['Start_Timestamp'])) / 1e3)
    for g, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
        P('  %-14s grid %-10s calls %3d avg %10.1f us' % (tag, g, len(v), sum(v) / len(v)))
def agg_pmc(tag, want):
    res = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(O + '/' + tag + '/*/*counter_collection.csv'):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'].split('(')[0]
            if any(x in k for x in want):
                if 'pg_pass_kernel' in k:
                    k = k[:40] + ' grid ' + str(r.get('Grid_Size', '?'))
                res[k[:60]][r['Counter_Name']].append(float(r['Counter_Value']))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in res.items()}
P('== PMC of the bench (python3 bench.py --steps 4 --warmup 1), averages per dispatch')
bench_k = ('psi2_pairs', 'chain_b', 'psi1T_y', 'elbo_front', 'model_prepare')
vals = collections.defaultdict(dict)
for tag in ('pmc_fetch', 'pmc_write', 'pmc1', 'pmc_c3_sq'):
    for k, d in agg_pmc(tag, bench_k).items():
        for c, v in d.items():
            P('  %-50s %-30s %.6g' % (k, c, v)); vals[k][c] = v
rec = {}
for k, v in vals.items():
    if 'psi2_pairs' in k and 'FETCH_SIZE' in v and 'WRITE_SIZE' in v:
        rec['config3_mixed'] = {'kernel': k.strip(), 'fetch_size_kib': v['FETCH_SIZE'], 'write_size_kib': v['WRITE_SIZE'],
                                'bytes_per_launch': 2 * 1024 * v['FETCH_SIZE'] + 1024 * v['WRITE_SIZE'],
                                'rule': '2 x FETCH_SIZE + WRITE_SIZE (KiB -> bytes), separate --pmc passes, MI355X_MICROARCH.md HBM section'}
if rec:
    json.dump(rec, open(os.path.join(O, 'traffic.json'), 'w'), indent=1)
    P('traffic.json:', json.dumps(rec))
P('== PMC of the training step at configs 3 and 5 (model.gradients() x 2), averages per dispatch: the pass kernel of stage B')
gp = collections.defaultdict(dict)
for tag in ('pmc_grad1', 'pmc_grad2', 'pmc_grad5_1', 'pmc_grad5_2'):          # (..5_: the same at config 5, the Q = 20 form of the kernel)
    for k, d in agg_pmc(tag, ('pg_pass_kernel',)).items():
        k = k + (' [config 5]' if '5_' in tag else '')
        for c, v in d.items():
            P('  %-50s %-30s %.6g' % (k, c, v)); gp[k][c] = v
for k, v in gp.items():
    if 'GRBM_GUI_ACTIVE' in v and 'SQ_VALU_MFMA_BUSY_CYCLES' in v:
        cyc = v['GRBM_GUI_ACTIVE'] / 8.0                          # (the counter sums the 8 XCDs)
        e = {'cycles_per_dispatch': cyc, 'mfma_busy_frac': v['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024.0 / cyc,
             'valu_active_frac': 4.0 * v['SQ_ACTIVE_INST_VALU'] / 1024.0 / cyc, 'coexec_frac': v['SQ_VALU_MFMA_COEXEC_CYCLES'] / 1024.0 / cyc,
             'wait_any_frac_of_wave_cycles': v['SQ_WAIT_ANY'] / v['SQ_WAVE_CYCLES'],
             'wait_inst_any_frac_of_wave_cycles': v['SQ_WAIT_INST_ANY'] / v['SQ_WAVE_CYCLES'],
             'note': 'per grid: the two large grids are the Psi2 passes, the two small ones the Psi1 passes'}
        P('  ->', k.strip(), json.dumps(e))
        pgj = os.path.join(O, 'pg_pass_pmc.json')
        allp = json.load(open(pgj)) if os.path.exists(pgj) else {}
        allp[k.strip()] = {'counters': v, 'derived': e}
        json.dump(allp, open(pgj, 'w'), indent=1)
P('== PMC of the gram / Cholesky workloads (python3 scratch/prof_linalg.py 3), averages per dispatch')
la_k = ('gram', 'potrf', 'pbig', 'pleft')
la = collections.defaultdict(dict)
for tag in ('pmc_la_fetch', 'pmc_la_write', 'pmc_la_sq'):
    for k, d in agg_pmc(tag, la_k).items():
        for c, v in d.items():
            P('  %-50s %-30s %.6g' % (k, c, v)); la[k][c] = v
digest = {}
for k, v in la.items():
    us = [t for (tag, name), t in dur.items() if tag == 'trace_linalg' and name.strip() == k.strip()]
    e = {'counters': v, 'avg_us_kernel_trace': us[0] if us else None}
    if 'WRITE_SIZE' in v and us:
        e['write_gbps'] = 1024 * v['WRITE_SIZE'] / (us[0] * 1e-6) / 1e9
        e['hbm_bytes'] = 2 * 1024 * v.get('FETCH_SIZE', 0.0) + 1024 * v['WRITE_SIZE']
    if 'SQ_INSTS_VALU_MFMA_MOPS_F64' in v and us:
        fl = 512.0 * v['SQ_INSTS_VALU_MFMA_MOPS_F64']
        e['mfma_flops_executed'] = fl
        e['mfma_tflops'] = fl / (us[0] * 1e-6) / 1e12
        e['mfma_utilisation_of_78.6'] = e['mfma_tflops'] / 78.6
    digest[k.strip()] = e
    P('  ->', k.strip(), json.dumps({a: b for a, b in e.items() if a != 'counters'}))
json.dump(digest, open(os.path.join(O, 'linalg_pmc.json'), 'w'), indent=1)
out.close()
