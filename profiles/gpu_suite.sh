#!/bin/bash
# The whole GPU suite, without -x so that one failure does not hide the rest; log under gpurun_out/suite/.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/suite
timeout 3300 python -m pytest tests -q -m gpu -s "$@" 2>&1 | grep -v "^$\|amdgpu.ids" > gpurun_out/suite/pytest.log
grep -n "flip log\|passed\|failed\|^FAILED\|^ERROR" gpurun_out/suite/pytest.log | tail -40
