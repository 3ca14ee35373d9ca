#!/bin/bash
# ONE parametrised GPU-box visit (replaces the gpu_visitN.sh one-offs of rounds 1-2).  Stages run in the order given:
#   gpurun --timeout 900 -- 'bash tools/gpu_visit.sh tests smoke bench prof:spherenet_qm9'
# stages:
#   tests[:expr]        pytest -m gpu (optionally -k expr)                      -> gpurun_out/pytest_gpu.log
#   lease               box id + smoke() + the whole GPU suite as the driver runs them -> gpurun_out/leases/lease_<utc>.json
#   smoke               __graft_entry__.smoke()                                  -> gpurun_out/smoke.log
#   bench[:workload]    full bench line (rooflines + PMC + cpu baseline for the headline) -> gpurun_out/bench_<w>.log
#   quick[:workload]    bench line without rooflines / cpu baseline              -> gpurun_out/quick_<w>.log
#   ab:NAME=VAL[:workload] quick bench with one kernel route flipped (bench.py --route; same-box A/B) -> gpurun_out/ab_<NAME>_<w>.log
#   prof[:workload]     rocprofv3 --kernel-trace --stats of a short bench        -> gpurun_out/prof_<w>/ + kernel stats csv
#   roofprof            rocprofv3 --kernel-trace --stats of the roofline launches -> gpurun_out/prof_roofline/
#   counters[:workload] MFMA / VALU busy counters (eager step), two --pmc passes  -> gpurun_out/pmc_<w>_{1,2}.csv
#   dense:M,K,N         tools/bench_dense.py M K N                               -> gpurun_out/dense_M_K_N.log
#   py:path             python <path> (a tools/ script)                          -> gpurun_out/<basename>.log
mkdir -p gpurun_out; export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
B="--steps ${STEPS:-20} --warmup ${WARMUP:-10}"
# every visit starts with the framework-only box probe: ~1 lease in 8 of this pool faults inside torch's own first copies
# (r04); such a box is characterised (which operation, which workaround) and the visit ends there
if [ -z "$SKIP_PROBE" ]; then
  timeout 900 python tools/box_probe.py || { echo "[visit] box unusable (probe rc=$?): stopping"; exit 3; }
fi
for st in "$@"; do
  IFS=':' read -r name a1 a2 <<< "$st"
  case $name in
    tests)
      if [ -n "$a1" ]; then K=(-k "$a1"); else K=(); fi
      DIG3D_PARITY_REPORT=gpurun_out/parity_report.json timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=8 "${K[@]}" > gpurun_out/pytest_gpu.log 2>&1
      echo "[tests] rc=$?"; tail -4 gpurun_out/pytest_gpu.log | cut -c1-300 ;;
    lease)    # one fresh-lease record: box id + smoke() + the whole GPU suite, as the driver runs them (-x, no cache provider)
      mkdir -p gpurun_out/leases; ts=$(date -u +%Y%m%dT%H%M%SZ)
      uuid=$(rocminfo 2>/dev/null | grep -m1 -E "Uuid: +GPU" | awk '{print $2}')
      timeout 300 python3 -c 'import sys; sys.path.insert(0, "."); import __graft_entry__ as e; e.smoke(); print("__SMOKE_OK__")' > gpurun_out/leases/smoke_$ts.log 2>&1; src=$?
      DIG3D_PARITY_REPORT=gpurun_out/leases/parity_$ts.json timeout 1500 python3 -m pytest tests/ -x -q -m gpu -p no:cacheprovider > gpurun_out/leases/pytest_$ts.log 2>&1; prc=$?
      python3 - "$ts" "$uuid" "$src" "$prc" <<'PY'
import json, subprocess, sys
ts, uuid, src, prc = sys.argv[1:5]
tail = open(f'gpurun_out/leases/pytest_{ts}.log').read().strip().splitlines()[-1]
smoke = [l for l in open(f'gpurun_out/leases/smoke_{ts}.log').read().splitlines() if l.startswith(('smoke:', '[smoke] device'))]
head = subprocess.run(['git', 'rev-parse', '--short', 'HEAD'], capture_output=True, text=True).stdout.strip()
rec = dict(utc=ts, gpu_uuid=uuid, kernel=open('/proc/sys/kernel/osrelease').read().strip(), smoke_rc=int(src), pytest_rc=int(prc),
           pytest_summary=tail, smoke=smoke)
json.dump(rec, open(f'gpurun_out/leases/lease_{ts}.json', 'w'))
print('[lease]', json.dumps(rec)[:400])
PY
      ;;
    smoke)
      timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "[smoke] rc=$?"; tail -2 gpurun_out/smoke.log ;;
    bench)
      w=${a1:-spherenet_qm9}
      timeout 900 python bench.py --workload $w $B ${BENCH_ARGS} > gpurun_out/bench_$w.log 2>&1; echo "[bench $w] rc=$?"
      grep '"metric"' gpurun_out/bench_$w.log | cut -c1-700 ;;
    quick)
      w=${a1:-spherenet_qm9}
      timeout 600 python bench.py --workload $w $B --no-roofline --no-cpu-baseline --no-through-loader ${BENCH_ARGS} > gpurun_out/quick_$w.log 2>&1
      echo "[quick $w] rc=$?"; grep '"metric"' gpurun_out/quick_$w.log | python -c "import sys,json; [print({k:d[k] for k in ('value','ms_per_step','ms_p10','ms_p90')}) for d in map(json.loads, sys.stdin)]" ;;
    ab)       # ab:NAME=VAL[:workload] — quick bench with one kernel route flipped (bench.py --route), same box as the other stages
      w=${a2:-spherenet_qm9}; tag=$(echo $a1 | tr '=' '_')
      timeout 600 python bench.py --workload $w $B --no-roofline --no-cpu-baseline --no-through-loader --route $a1 ${BENCH_ARGS} > gpurun_out/ab_${tag}_$w.log 2>&1
      echo "[ab $a1 $w] rc=$?"; grep '"metric"' gpurun_out/ab_${tag}_$w.log | python -c "import sys,json; [print({k:d[k] for k in ('value','ms_per_step','ms_p10','ms_p90')}) for d in map(json.loads, sys.stdin)]" ;;
    prof)
      w=${a1:-spherenet_qm9}; rm -rf gpurun_out/prof_$w
      timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$w -o p --output-format csv -- python bench.py --workload $w --steps 10 --warmup 5 --windows 1 --no-roofline --no-cpu-baseline --no-through-loader ${BENCH_ARGS} > gpurun_out/prof_$w.log 2>&1
      echo "[prof $w] rc=$?"; find gpurun_out/prof_$w -name '*kernel_trace.csv' -delete
      f=$(find gpurun_out/prof_$w -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f gpurun_out/kernel_stats_$w.csv && head -12 $f | cut -c1-160 ;;
    seq)      # seq[:workload] — the ordered kernel list of ONE replayed step (between two k_adam_flat launches)
      w=${a1:-spherenet_qm9}; rm -rf /tmp/prof_seq
      (cd /tmp && timeout 900 rocprofv3 --kernel-trace -d /tmp/prof_seq -o p --output-format csv -- python $R/bench.py --workload $w --steps 4 --warmup 6 --windows 1 --no-roofline --no-cpu-baseline --no-through-loader ${BENCH_ARGS} > $R/gpurun_out/seq_$w.log 2>&1); echo "[seq $w] rc=$?"
      f=$(find /tmp/prof_seq -name '*kernel_trace.csv' | head -1)
      [ -n "$f" ] && python - "$f" gpurun_out/kernel_sequence_$w.txt <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
ad = [i for i, r in enumerate(rows) if 'k_adam_flat' in r['Kernel_Name']]
a, b = ad[-2], ad[-1]
with open(sys.argv[2], 'w') as f:
    for r in rows[a:b + 1]:
        f.write(f"{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:8.1f} us  {r['Kernel_Name'][:110]}\n")
print('kernels in one step:', b - a)
PY
      ;;
    roofprof)
      rm -rf gpurun_out/prof_roofline
      timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_roofline -o p --output-format csv -- python bench.py --steps 5 --warmup 5 --windows 1 --no-pmc --no-cpu-baseline --no-through-loader > gpurun_out/prof_roofline.log 2>&1
      echo "[roofprof] rc=$?"; find gpurun_out/prof_roofline -name '*kernel_trace.csv' -delete
      f=$(find gpurun_out/prof_roofline -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f gpurun_out/kernel_stats_roofline.csv
      grep '"metric"' gpurun_out/prof_roofline.log > gpurun_out/bench_line_under_rocprof.json
      grep -E "k_segsum_sorted|k_seg_fused|k_featconv|k_trip_fwd" gpurun_out/kernel_stats_roofline.csv | cut -c1-200 ;;
    counters)
      w=${a1:-spherenet_qm9}; i=0
      for pm in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "MfmaUtil VALUBusy"; do
        i=$((i+1)); rm -rf /tmp/pmc_$i
        (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $pm -d /tmp/pmc_$i -o p --output-format csv -- python $R/bench.py --workload $w --steps 3 --warmup 5 --windows 1 --no-cpu-baseline --no-roofline --no-through-loader --eager > $R/gpurun_out/pmc_${w}_$i.log 2>&1); echo "[counters $w pass $i] rc=$?"
        f=$(find /tmp/pmc_$i -name '*counter_collection.csv' | head -1); [ -n "$f" ] && cp $f gpurun_out/pmc_${w}_$i.csv
      done ;;
    stall)    # stall[:workload] — wave-cycle breakdown / instruction mix / memory-pipe counters of one eager step, three --pmc passes
      w=${a1:-spherenet_qm9}; i=0
      for pm in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" \
                "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM_RD SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE" \
                "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE"; do
        i=$((i+1)); rm -rf /tmp/stall_$i
        (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $pm -d /tmp/stall_$i -o p --output-format csv -- python $R/bench.py --workload $w --steps 3 --warmup 4 --windows 1 --no-cpu-baseline --no-roofline --no-through-loader --eager > $R/gpurun_out/stall_${w}_$i.log 2>&1); echo "[stall $w pass $i] rc=$?"
        f=$(find /tmp/stall_$i -name '*counter_collection.csv' | head -1); [ -n "$f" ] && cp $f gpurun_out/stall_${w}_$i.csv
      done
      python tools/summarize_counters.py gpurun_out/stall_${w}_*.csv > gpurun_out/stall_counters_$w.json && python tools/stall_report.py gpurun_out/stall_counters_$w.json ;;
    chain)    # chain:M,K0  — csrc/chain.hip stand-alone: kernel stats + the wave-cycle breakdown counters
      shp=$(echo ${a1:-8704,64} | tr ',' ' '); tag=$(echo ${a1:-8704,64} | tr ',' '_'); rm -rf gpurun_out/prof_chain_$tag /tmp/pmc_chain
      timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_chain_$tag -o p --output-format csv -- python tools/bench_chain.py $shp > gpurun_out/chain_$tag.log 2>&1; echo "[chain $shp] rc=$?"
      grep "^M=" gpurun_out/chain_$tag.log; find gpurun_out/prof_chain_$tag -name '*kernel_trace.csv' -delete
      f=$(find gpurun_out/prof_chain_$tag -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && grep -E "chain|reduce" $f | cut -d, -f1-4 | cut -c1-50,80-
      (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d /tmp/pmc_chain -o p --output-format csv -- python $R/tools/bench_chain.py $shp 10 > $R/gpurun_out/pmc_chain_$tag.log 2>&1); echo "[chain counters] rc=$?"
      f=$(find /tmp/pmc_chain -name '*counter_collection.csv' | head -1); [ -n "$f" ] && cp $f gpurun_out/pmc_chain_$tag.csv && python tools/summarize_counters.py gpurun_out/pmc_chain_$tag.csv > gpurun_out/pmc_chain_$tag.json && python -c "
import json; d=json.load(open('gpurun_out/pmc_chain_$tag.json'))
for k,v in d.items():
    c=v['counters']; w=c.get('SQ_WAVE_CYCLES',1)
    print(k, {n: round(c[n]/w,3) for n in c if n.startswith('SQ_') and n!='SQ_WAVE_CYCLES'}, 'mfma_busy', v.get('mfma_busy_frac'), 'us', v.get('kernel_us_at_2.4GHz'))
" ;;
    dense)
      shp=$(echo $a1 | tr ',' ' '); tag=$(echo $a1 | tr ',' '_')
      timeout 600 python tools/bench_dense.py $shp > gpurun_out/dense_$tag.log 2>&1; echo "[dense $shp] rc=$?"; grep "^M=" gpurun_out/dense_$tag.log ;;
    py)
      timeout 900 python $a1 ${a2} > gpurun_out/$(basename $a1 .py).log 2>&1; echo "[py $a1] rc=$?"; tail -25 gpurun_out/$(basename $a1 .py).log | cut -c1-250 ;;
    *) echo "unknown stage $st" ;;
  esac
done
