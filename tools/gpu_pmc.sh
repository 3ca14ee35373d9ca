#!/bin/bash
# HBM traffic of the scatter_add roofline kernel from PMC counters: separate passes for FETCH_SIZE and WRITE_SIZE
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
cd /tmp; rm -rf /tmp/pf /tmp/pw
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pf -o p --output-format csv -- python $R/tools/pmc_scatter.py > /tmp/pf.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pw -o p --output-format csv -- python $R/tools/pmc_scatter.py > /tmp/pw.log 2>&1
cd $R
grep "^M " /tmp/pf.log
python - <<'PY'
import csv, glob, json, collections
res = {}
for d, cn in (('/tmp/pf', 'FETCH_SIZE'), ('/tmp/pw', 'WRITE_SIZE')):
    f = glob.glob(d + '/**/*counter_collection.csv', recursive=True)[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] == cn:
            k = 'segsum' if 'k_segsum_sorted' in r['Kernel_Name'] else ('calib' if 'k_gather_mul' in r['Kernel_Name'] else None)
            if k: acc[k].append(float(r['Counter_Value']))
    res[cn] = {k: sum(v[2:]) / len(v[2:]) for k, v in acc.items()}
print(json.dumps(res))
json.dump(res, open('gpurun_out/scatter_pmc_raw.json', 'w'))
PY
