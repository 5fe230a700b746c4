#!/bin/bash
# copies the evidence of tools/r06_run6.sh (gpurun_out/r06f) into profiles/ under the r06h names and regenerates profiles/pmc_traffic.json
S=gpurun_out/r06f
cp $S/bench.json profiles/r06h_bench.json; cp $S/bench_steps20_1.json profiles/r06h_bench_steps20_warmup5.json
cp $S/driver_style_three_runs.txt profiles/r06h_driver_style_three_runs.txt
cp $S/kernel_stats_graph.csv profiles/r06h_kernel_stats_graph.csv; cp $S/kernel_stats_eager.csv profiles/r06h_kernel_stats_eager.csv
cp $S/pmc/pmc_counters.txt profiles/r06_pmc_counters.txt; cp $S/trackers_on_device.txt profiles/r06h_trackers_on_device.txt
cp $S/frame_extended.json profiles/r06h_frame_extended.json; cp $S/tomp_kernels.txt profiles/r06h_tomp_kernel_stats.txt
cp $S/atom_kernels.csv profiles/r06h_atom_kernel_stats.csv; cp $S/iou_kernels.csv profiles/r06h_iou_kernel_stats.csv
tail -4 $S/pytest.log > profiles/r06h_gpu_tests_tail.txt
cp $S/graph_kernel_stats_dimp50.csv profiles/r06h_graph_kernel_stats_dimp50.csv; cp $S/graph_kernel_stats_prdimp50.csv profiles/r06h_graph_kernel_stats_prdimp50.csv
cp $S/smoke.txt profiles/r06h_smoke.txt
python tools/pmc_traffic.py profiles/r06_pmc_counters.txt > /dev/null
python - <<'PY'
import json
d=json.loads(open('profiles/r06h_bench.json').read().strip().splitlines()[-1])
r=d['roofline']
print('value',d['value'], d['ms_per_step'], 'frac',r['frac'], 'avg',r['avg_launch_us'], 'floor',r['stream_floor_us'], r['frac_of_floor'], 'traffic',r['traffic'], 'solve',r['solve_level']['frac'], r['solve_level']['achieved_GBs'], 'adj',r['kernels']['k_adj2']['avg_launch_us'], r['kernels']['k_adj2']['achieved_GBs'], 'corr it', r['kernels']['k_corr2']['avg_launch_us_in_iteration'], r['achieved'])
print('cpu',d['cpu_baseline']['value'], d['cpu_baseline']['port']['value'], d['cpu_baseline']['one_thread']['value'], 'gpu',d['gpu_stock_baseline']['value'], 'head',d['head_inclusive']['value'], d['head_inclusive']['us_per_frame'], 'ms2',d['multi_sequence']['2']['frames_per_s'])
o=d['other_workloads']
print('prdimp',o['prdimp50_frame']['frames_per_s'], o['prdimp50_frame']['ms'], o['prdimp50_frame']['frac'], 'tomp',o['tomp_predict']['ms'], o['tomp_predict']['frac'], 'lwl',o['lwl_n32_it3']['ms'], o['lwl_n32_it3']['frac'], o['lwl_n8_it3']['ms'], o['lwl_n8_it3']['frac'], 'atom',o['atom_cg_n250']['ms'], o['atom_cg_n250']['frac'], o['atom_cg_n250']['survey_8d_count']['frac'], o['iou_refine'])
print('e2e',d['end_to_end']['frames_per_s'])
print(open('profiles/r06h_driver_style_three_runs.txt').read())
j=json.load(open('profiles/pmc_traffic.json')); print({k:{kk:vv['hbm_bytes_per_launch'] for kk,vv in v.items()} for k,v in j.items() if k!='_note'})
PY
