#!/bin/bash
# gpurun_out/ (scratch) -> profiles/<tag>_* (tracked): the files of one `tools/gpu_visit.sh lease bench quick:* prof:* seq:* roofprof`
# visit under the round's names.  usage: bash tools/collect_profiles.sh r06
t=${1:-r06}; o=gpurun_out; p=profiles
line() { grep '"metric"' "$1" 2>/dev/null | tail -1; }
[ -f $o/bench_spherenet_qm9.log ] && line $o/bench_spherenet_qm9.log > $p/${t}_bench_line_default.json
for w in schnet_qm9 dimenetpp_md17_force spherenet_oc20 comenet_128 comenet_qm9; do
  [ -f $o/quick_$w.log ] && line $o/quick_$w.log > $p/${t}_bench_line_$w.json
done
for w in spherenet_qm9 dimenetpp_md17_force spherenet_oc20 comenet_128 comenet_qm9 schnet_qm9; do
  [ -f $o/kernel_stats_$w.csv ] && cp $o/kernel_stats_$w.csv $p/${t}_${w}_kernel_stats.csv
  [ -f $o/kernel_sequence_$w.txt ] && cp $o/kernel_sequence_$w.txt $p/${t}_kernel_sequence_$w.txt
done
[ -f $o/kernel_stats_roofline.csv ] && cp $o/kernel_stats_roofline.csv $p/${t}_scatter_roofline_kernel_stats.csv
[ -f $o/bench_line_under_rocprof.json ] && cp $o/bench_line_under_rocprof.json $p/${t}_scatter_roofline_bench_line_under_rocprof.json
mkdir -p $p/${t}_leases
l=$(ls -t $o/leases/lease_*.json 2>/dev/null | head -1)
if [ -n "$l" ]; then
  ts=$(basename $l .json | sed 's/lease_//')
  cp $l $p/${t}_leases/; cp $o/leases/smoke_$ts.log $p/${t}_leases/ 2>/dev/null
  tail -5 $o/leases/pytest_$ts.log > $p/${t}_leases/pytest_tail_$ts.txt
  [ -f $o/leases/parity_$ts.json ] && cp $o/leases/parity_$ts.json $p/${t}_parity_report.json
fi
ls -la $p | grep " ${t}_" | wc -l
