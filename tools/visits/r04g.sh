#!/bin/bash
# r04g: graph re-instantiation vs engine instance: where does the 1 us-per-launch bimodality live?
set -u
TAG=r04g; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
for i in 1 2 3; do timeout 300 python tools/graph_mode_probe.py mistral 8 2>&1 | grep -v amdgpu.ids; echo ---; done | tee $OUT/mistral.txt
for i in 1 2; do timeout 300 python tools/graph_mode_probe.py 7b 8 2>&1 | grep -v amdgpu.ids; echo ---; done | tee $OUT/7b.txt
