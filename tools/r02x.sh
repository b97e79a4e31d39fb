#!/bin/bash
mkdir -p gpurun_out
timeout 600 python tools/prefill_stress3.py 2>&1 | grep -E "equal|Error|error"
