#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
for shp in "8418 128 128" "600 256 256" "262144 128 128"; do
  tag=$(echo $shp | tr ' ' '_')
  rm -rf gpurun_out/prof_dense_$tag
  timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_dense_$tag -o d --output-format csv -- python tools/bench_dense.py $shp > gpurun_out/prof_dense_$tag.log 2>&1
  echo "== $shp"; grep "^M=" gpurun_out/prof_dense_$tag.log
  python - <<PY
import csv
rows=list(csv.DictReader(open('gpurun_out/prof_dense_$tag/d_kernel_stats.csv')))
for r in rows:
    if 'k_linear' in r['Name'] or 'k_dense' in r['Name'] or 'Cijk' in r['Name']:
        print(f"  {r['Name'][:60]:60s} calls {r['Calls']:>4} avg {float(r['AverageNs'])/1e3:8.1f} us min {float(r['MinNs'])/1e3:8.1f}")
PY
  rm -f gpurun_out/prof_dense_$tag/d_kernel_trace.csv
done
rocm-smi --showclocks 2>/dev/null | head -20
