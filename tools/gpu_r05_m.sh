#!/bin/bash
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_parity_gpu.py -q -m gpu -s -k "activation or predict_epsilon or inpaint or long_ddim or footwork or honours" 2>&1 | grep -E "activation=|inpaint|long|ootwork|ddim_sample clip|passed|failed|rror" | tail -24
timeout 900 python -m pytest tests/test_train_step_gpu.py tests/test_train_gpu.py -q -m gpu -s -k "activation or restart" 2>&1 | grep -E "activation=|passed|failed|rror" | tail -8
