#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 700 python bench.py > gpurun_out/r03_bench_S1.json 2> gpurun_out/r03_bench_S1.err; echo "bench exit $?" >> gpurun_out/r03_bench_S1.err
for W in S0_pubmed S2_gamlp; do timeout 600 python bench.py --workload $W --steps 5 2>/dev/null | tail -1 > gpurun_out/r03_bench_$W.json; done
timeout 900 python bench.py --workload S3_papers_shard --steps 5 2>/dev/null | tail -1 > gpurun_out/r03_bench_S3_shard.json
timeout 900 python tools/pasca_sweep.py > gpurun_out/r03_pasca_sweep.log 2>&1
timeout 1500 python tools/pasca_sweep_papers.py > gpurun_out/r03_pasca_sweep_papers_shard.log 2>&1
grep "msg_op\|graph_op" gpurun_out/r03_pasca_sweep_papers_shard.log | head -30
timeout 600 python examples/gamlp_label_reuse_synthetic.py > gpurun_out/r03_example_gamlp_label_reuse.log 2>&1; tail -5 gpurun_out/r03_example_gamlp_label_reuse.log
timeout 600 python examples/sgc_synthetic.py > gpurun_out/r03_example_sgc.log 2>&1; tail -3 gpurun_out/r03_example_sgc.log
for f in gpurun_out/r03_bench_S0_pubmed.json gpurun_out/r03_bench_S2_gamlp.json gpurun_out/r03_bench_S3_shard.json; do python -c "import json,sys;j=json.load(open('$f'));print('$f', round(j['ms_per_step'],3), round(j['roofline']['frac'],3))"; done
