# the four single-GPU workloads, one JSON line each into gpurun_out/bench_<w>.json
for w in shell diel er rmat; do
  timeout -k 10 300 python bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline --no-also > gpurun_out/bench_$w.json 2> gpurun_out/bench_$w.err || { tail -3 gpurun_out/bench_$w.err; exit 1; }
  python - <<PY
import json
d=json.load(open('gpurun_out/bench_$w.json'))
wm=d['whole_multiply']
print('$w', round(d['ms_per_step'],2),'ms', round(d['value'],1),'GF/s model_no_reuse_frac', round(wm['frac_model_no_reuse'],3), 'comp_frac', round(wm['compulsory_frac_of_hbm_peak'],3), 'copy', round(wm['measured_copy_GBs']), {k:round(v,2) for k,v in wm['phase_ms'].items()})
print('   dom', d['roofline']['kernel'], round(d['roofline']['kernel_ms'],2), round(d['roofline']['frac'],3))
PY
done
