#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06o; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_engine.py tests/test_gpu_parity.py tests/test_gpu_attention_fullgeom.py tests/test_gpu_fullsize_oracle.py tests/test_gpu_api.py -q -m gpu --maxfail=10 -k "not prefill_gemm" > $O/pytest.txt 2>&1; echo "pytest rc=$?"; grep -n "passed\|failed\|FAILED" $O/pytest.txt | tail -8 | cut -c1-200
