#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_models.py -q -p no:cacheprovider -x -k "force" > gpurun_out/pytest_c.log 2>&1; echo "pytest rc=$?"; tail -12 gpurun_out/pytest_c.log | cut -c1-300
for i in 1 2; do
timeout 400 python bench.py --workload dimenetpp_md17_force --steps 10 --warmup 5 --no-cpu-baseline --no-pmc --no-roofline > gpurun_out/bench_c3.log 2>&1; echo "[config3 heads2] $(tail -1 gpurun_out/bench_c3.log | cut -c60-200)"
DIG3D_NO_HEADS2=1 timeout 400 python bench.py --workload dimenetpp_md17_force --steps 10 --warmup 5 --no-cpu-baseline --no-pmc --no-roofline > gpurun_out/bench_c3n.log 2>&1; echo "[config3 torch heads] $(tail -1 gpurun_out/bench_c3n.log | cut -c60-200)"
done
python - <<'PY'
import torch
x = torch.randn(1 << 22, 128, device='cuda')
for _ in range(3): x.sum()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20): x.sum()
b.record(); torch.cuda.synchronize()
t = a.elapsed_time(b) / 20
print(f'torch.sum of 2.147 GB: {t*1e3:.1f} us -> {x.numel()*4/t/1e9:.2f} TB/s')
PY
