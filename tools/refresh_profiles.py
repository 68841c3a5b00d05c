"""Copies the newest rocprofv3 summaries from gpurun_out/prof (tools/profile.sh) into profiles/round1."""
import csv, glob, json, os, shutil
P, D = 'gpurun_out/prof', 'profiles/round1'
newest = lambda pat: sorted(glob.glob(pat), key=os.path.getmtime)[-1]
shutil.copy(newest(f'{P}/trace/runc/*_kernel_stats.csv'), f'{D}/kernel_stats.csv')
q = lambda r: ','.join('"%s"' % c for c in r)
rows = list(csv.reader(open(newest(f'{P}/trace/runc/*_kernel_trace.csv'))))
open(f'{D}/kernel_trace_preissmann.csv', 'w').write('\n'.join(q(r) for r in rows if r and (r[0] == 'Kind' or 'preissmann' in ','.join(r))) + '\n')
vals = {}
for name in ('pmc_fetch', 'pmc_write', 'pmc_sq', 'pmc_flops'):
    rows = list(csv.reader(open(newest(f'{P}/{name}/runc/*_counter_collection.csv'))))
    keep = [r for r in rows if r and (r[0] == 'Correlation_Id' or 'preissmann' in ','.join(r))]
    open(f'{D}/{name}_preissmann.csv', 'w').write('\n'.join(q(r) for r in keep) + '\n')
    h = keep[0]; ci, cv, di = h.index('Counter_Name'), h.index('Counter_Value'), h.index('Dispatch_Id')
    last = max(int(r[di]) for r in keep[1:])
    vals.update({r[ci]: float(r[cv]) for r in keep[1:] if int(r[di]) == last})
B, K = 65536, 8
fetch, write = vals['FETCH_SIZE'] * 1024, vals['WRITE_SIZE'] * 1024
json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), tools/profile.sh, round 1 (final kernel)",
           "workload": "65536 reaches x 4096 nodes, 8 steps in one launch", "fetch_bytes_per_launch": fetch,
           "write_bytes_per_launch": write,
           "fetch_note": "raw counter x 1024; MI355X_MICROARCH.md: FETCH_SIZE under-reports wide coalesced reads by 2x on gfx950, this kernel reads 8 B/lane (uncalibrated)",
           "hbm_bytes_per_reach_timestep": (fetch + write) / (B * K)}, open(f'{D}/hbm_traffic.json', 'w'), indent=1)
print(open(f'{D}/kernel_stats.csv').read().split('\n')[1][:200])
print('fetch GB', fetch / 1e9, 'write GB', write / 1e9, 'per reach-timestep', (fetch + write) / (B * K))
w = vals['SQ_WAVES']
print({k: f'{v / w:.4g}' for k, v in vals.items() if k.startswith('SQ_INSTS')}, 'valu active', vals['SQ_ACTIVE_INST_VALU'] / vals['SQ_WAVE_CYCLES'])

# fp64 work per Newton iteration of one reach (wave-instruction counts x 64 lanes; an FMA is two flops)
bj = json.loads(open(f'{P}/bench_flops.json').read().strip().split('\n')[-1])
its = bj['config']['mean_newton_iterations_per_step']
flops = 64 * (2 * vals['SQ_INSTS_VALU_FMA_F64'] + vals['SQ_INSTS_VALU_MUL_F64'] + vals['SQ_INSTS_VALU_ADD_F64'] + vals['SQ_INSTS_VALU_TRANS_F64'])
json.dump({"source": "rocprofv3 --pmc SQ_INSTS_VALU_{FMA,MUL,ADD,TRANS}_F64 (tools/profile.sh), last launch of the run",
           "workload": "65536 reaches x 4096 nodes, 8 steps in one launch", "mean_newton_iterations_per_step": its,
           "fp64_flops_per_launch": flops, "fp64_flops_per_reach_iteration": flops / (B * K * its),
           "valu_instructions_per_wave_iteration": vals['SQ_INSTS_VALU'] / vals['SQ_WAVES'] / (K * its)},
          open(f'{D}/fp64_flops.json', 'w'), indent=1)
print('fp64 flops per reach-iteration', flops / (B * K * its))
