#!/bin/bash
# r05d: fp8 decode at K = 11008 (parity), act-order layers in the engine (parity + tokens/s vs int4, same box)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05d; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_wtypes_decode.py tests/test_gpu_engine.py -q -m gpu -k "fp8_weight_types_decode_kernel or act_order" > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
timeout 300 python tools/visits/r05d.py > $O/act_order.txt 2> $O/act_order.err; echo "act_order rc=$?"; cat $O/act_order.txt; tail -2 $O/act_order.err
WOQ_ENGINE_XQ=0 timeout 300 python tools/visits/r05d.py > $O/act_order_xq0.txt 2> $O/act_order_xq0.err; cat $O/act_order_xq0.txt
WOQ_SHUFFLE_GENERIC=1 timeout 300 python tools/visits/r05d.py > $O/act_order_generic.txt 2> $O/act_order_generic.err; cat $O/act_order_generic.txt
