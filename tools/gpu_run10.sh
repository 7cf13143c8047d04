#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python examples/sgc_synthetic.py > gpurun_out/ex_sgc.log 2>&1; echo "exit $?" >> gpurun_out/ex_sgc.log
timeout 900 python examples/gamlp_label_reuse_synthetic.py > gpurun_out/ex_gamlp.log 2>&1; echo "exit $?" >> gpurun_out/ex_gamlp.log
tail -4 gpurun_out/ex_sgc.log; tail -8 gpurun_out/ex_gamlp.log
