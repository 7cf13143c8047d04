#!/bin/bash
# first GPU session: parity tests, smoke, bench, rocprof kernel stats
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
rocm-smi --showproductname 2>&1 | head -8 > gpurun_out/gpu_info.log
nproc >> gpurun_out/gpu_info.log; lscpu | grep -E "Model name|^CPU\(s\)|Thread|Socket" >> gpurun_out/gpu_info.log
timeout 1500 python -m pytest tests -m gpu -q --maxfail=40 -p no:cacheprovider -s > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1
echo "smoke exit $?" >> gpurun_out/smoke.log
timeout 600 python bench.py --steps 5 --warmup 2 > gpurun_out/bench.log 2>&1
echo "bench exit $?" >> gpurun_out/bench.log
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/prof_r1" -o r1 -- python "$GRAFT_REPO_ROOT/bench.py" --steps 5 --warmup 2 --no-cpu-baseline > "$GRAFT_REPO_ROOT/gpurun_out/rocprof_bench.log" 2>&1
echo "rocprof exit $?" >> "$GRAFT_REPO_ROOT/gpurun_out/rocprof_bench.log"
cd "$GRAFT_REPO_ROOT"
find gpurun_out/prof_r1 -name "*.db" -size +20M -delete 2>/dev/null
tail -5 gpurun_out/pytest_gpu.log; tail -3 gpurun_out/smoke.log; tail -2 gpurun_out/bench.log
