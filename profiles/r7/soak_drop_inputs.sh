#!/bin/bash
# Round 7: tests/tools/soak_node.py with every view's inputs dropped right after its call (views still queued on their side
# streams), with the binding's stream record and -- to see that the soak can tell -- without it.
cd "$(dirname "$0")/../.."
O=gpurun_out/r7_soak_drop; mkdir -p $O
timeout 600 python tests/tools/soak_node.py --seconds 150 --seed 11 2>&1 | grep -v amdgpu | tail -3 | tee $O/soak.txt
timeout 600 python tests/tools/soak_node.py --seconds 150 --seed 12 --drop-inputs 2>&1 | grep -v amdgpu | tail -3 | tee -a $O/soak.txt
echo "== DGR_RECORD_INPUT_STREAMS=0 --drop-inputs (mismatches expected)" | tee -a $O/soak.txt
DGR_RECORD_INPUT_STREAMS=0 timeout 300 python tests/tools/soak_node.py --seconds 60 --seed 12 --drop-inputs 2>&1 | grep -v amdgpu | tail -4 | cut -c1-400 | tee -a $O/soak.txt
