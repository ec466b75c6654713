#!/bin/bash
out=${1:-gpurun_out/r7l}
mkdir -p $out
timeout 1500 python -m pytest tests -q -m gpu --timeout 400 -x --deselect tests/test_rcnn_gpu.py::test_r101_batch16_full_proposals_against_the_vectorised_oracle 2>&1 | tail -25 > $out/pytest_gpu.txt
tail -12 $out/pytest_gpu.txt
timeout 900 python bench.py > $out/bench_default.json 2> $out/bench_default.err
python -c "
import json
d=json.loads(open('$out/bench_default.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step']); print(json.dumps(d['roofline'])[:1500]); print({k:(v.get('value'), v.get('roofline',{}).get('frac')) for k,v in d['configs'].items() if isinstance(v,dict)})"
