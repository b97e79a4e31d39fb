#!/bin/bash
mkdir -p gpurun_out
timeout 600 python tools/visits/prefill_stress3.py 2>&1 | grep -E "equal|Error|error"
