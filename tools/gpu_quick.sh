#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/pmc1 /tmp/pmc2
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU -d /tmp/pmc1 -o p --output-format csv -- python $GRAFT_REPO_ROOT/tools/bench_dense.py 8418 128 128 > /tmp/pmc1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS -d /tmp/pmc2 -o p --output-format csv -- python $GRAFT_REPO_ROOT/tools/bench_dense.py 8418 128 128 > /tmp/pmc2.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, collections
for d in ('/tmp/pmc1','/tmp/pmc2'):
    f = glob.glob(d+'/**/*counter_collection.csv', recursive=True)
    if not f:
        print('no counter file in', d, glob.glob(d+'/**', recursive=True)[:10]); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        k = r['Kernel_Name'][:40]
        if 'k_linear' in k or 'k_dense' in k:
            acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, c in acc.items():
        print(k, {n: round(sum(v)/len(v)) for n, v in c.items()}, 'n=', len(next(iter(c.values()))))
PY
tail -3 /tmp/pmc1.log | cut -c1-300
