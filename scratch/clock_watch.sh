#!/bin/bash
# sample sclk / power while bench runs
python bench.py --steps 60 --warmup 5 --no_cpu_baseline > gpurun_out/bench_long.json 2>/dev/null &
BP=$!
for i in $(seq 1 40); do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' '; echo
  sleep 0.25
  if ! kill -0 $BP 2>/dev/null; then break; fi
done
wait $BP
python -c "
import json; d=json.load(open('gpurun_out/bench_long.json')); print(d['ms_per_step'], {k:round(v['ms_per_step'],2) for k,v in d['roofline']['families'].items()})"
