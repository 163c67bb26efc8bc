#!/bin/bash
# Round 7: the forward records its inputs on a side stream (csrc/torch_ext.cpp: keep_until_read) -- the test with and without it,
# what it costs on the host, and the timeline of the driver's command.
cd "$(dirname "$0")/../.."
O=gpurun_out/r7_lifetime; mkdir -p $O
T=tests/test_hip_front_end.py::test_inputs_made_on_the_callers_stream_and_dropped_after_the_call_stay_valid
echo "== with the record" | tee $O/test.txt
python -m pytest -x -q -m gpu $T 2>&1 | tail -3 | tee -a $O/test.txt
echo "== DGR_RECORD_INPUT_STREAMS=0 (the test is expected to FAIL here)" | tee -a $O/test.txt
DGR_RECORD_INPUT_STREAMS=0 python -m pytest -x -q -m gpu $T 2>&1 | tail -4 | tee -a $O/test.txt
for rec in 1 0; do
  for i in 1 2; do
    echo "record=$rec config2 full: $(DGR_RECORD_INPUT_STREAMS=$rec python bench.py --workload config2 --variant full --no-cpu-baseline | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d.get('ms_per_view_one_stream'))")" | tee -a $O/cost.txt
    echo "record=$rec config3 driver cmd: $(DGR_RECORD_INPUT_STREAMS=$rec python bench.py --steps 20 --warmup 5 --no-cpu-baseline | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")" | tee -a $O/cost.txt
  done
done
bash profiles/r7/timeline.sh 20 20
