#!/bin/bash
# The measurements DESIGN.md sec. 6 quotes, in one pass on the MI355X box:  tools/final_measure.sh <outdir>
set -u
OUT=${1:-gpurun_out/final}
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/$OUT
cd $R
if [ "${SKIP_PYTEST:-0}" != "1" ]; then
python -m pytest tests -m gpu -q -s 2>&1 | grep -E "passed|failed|max-abs|max \|GPU|final map|vs oracle|fp64|worst" > $OUT/pytest_gpu.txt
fi
# the driver's own command form first (compact contract line, config 4 only), then the full record (all modes, all configurations)
python bench.py --gpus 1 --steps 20 --warmup 5 --detail $OUT/bench_detail_driver_form.json > $OUT/bench_line.json 2> $OUT/bench_stderr.txt
python bench.py --steps 20 --warmup 5 --also all --configs all --op-table $OUT/op_table_fp32.json --detail $OUT/bench_detail.json > $OUT/bench_line_full.json 2>> $OUT/bench_stderr.txt
for m in bf16x6 fp16x3 bf16x3; do python bench.py --precision $m --steps 10 --warmup 3 --no-cpu-baseline --also "" --traffic measure --op-table $OUT/op_table_$m.json 2>/dev/null | grep "^{" ; done > $OUT/bench_modes.json
python bench.py --config 5 --steps 5 --warmup 2 --no-cpu-baseline --also "bf16x6,fp16x3" --traffic none 2>/dev/null | grep "^{" > $OUT/bench_config5.json
for m in fp32 bf16x6 fp16x3; do
python bench.py --batch 1 --size 240 --precision $m --steps 50 --warmup 5 --no-cpu-baseline --also "" --traffic none --no-probe 2>/dev/null | grep "^{"
python bench.py --batch 1 --size 720 --precision $m --steps 50 --warmup 5 --no-cpu-baseline --also "" --traffic none --no-probe 2>/dev/null | grep "^{"
done > $OUT/bench_b1.jsonl
python tools/bench_rcnn.py 16 2>/dev/null | grep "^{" > $OUT/rcnn_b16.jsonl
python tools/bench_rcnn.py 1 2>/dev/null | grep "^{" > $OUT/rcnn_b1.jsonl
: > $OUT/pipeline_config4.jsonl
python tools/bench_pipeline.py --episodes 2 --frames 60 2>/dev/null | grep "^{" >> $OUT/pipeline_config4.jsonl
python tools/bench_pipeline.py --episodes 2 --frames 60 --precision bf16x6 2>/dev/null | grep "^{" >> $OUT/pipeline_config4.jsonl
python tools/bench_pipeline.py --episodes 2 --frames 60 --detector 2>/dev/null | grep "^{" >> $OUT/pipeline_config4.jsonl
python tools/bench_pipeline.py --episodes 2 --frames 40 --detector 2>/dev/null | grep "^{" >> $OUT/pipeline_config4.jsonl          # the bench line's own shape
python tools/bench_pipeline.py --episodes 2 --frames 60 --detector --serial-goal 2>/dev/null | grep "^{" >> $OUT/pipeline_config4.jsonl   # goal selection after the forward, timed on its own
python tools/bench_pipeline.py --episodes 2 --frames 60 --detector --precision bf16x6 2>/dev/null | grep "^{" >> $OUT/pipeline_config4.jsonl
python tools/bench_pipeline.py --episodes 2 --frames 60 --detector --precision fp16x3 2>/dev/null | grep "^{" >> $OUT/pipeline_config4.jsonl
python tools/bench_pipeline.py --episodes 2 --frames 60 --detector --precision bf16x3 2>/dev/null | grep "^{" >> $OUT/pipeline_config4.jsonl
cd /tmp; export TMPDIR=/tmp
for m in fp32 bf16x6 fp16x3; do
rm -rf /tmp/fm_trace
rocprofv3 --kernel-trace --stats -d /tmp/fm_trace -- python $R/bench.py --precision $m --steps 10 --warmup 3 --no-cpu-baseline --no-probe --also "" --traffic none --configs "" > /tmp/fm_trace.log 2>&1
db=$(find /tmp/fm_trace -name '*.db' | head -1); [ -n "$db" ] && python $R/tools/rocpd_summary.py $db > $R/$OUT/trace_$m.txt
done
rm -rf /tmp/fm_trace2
rocprofv3 --kernel-trace -d /tmp/fm_trace2 -- python $R/tools/measure_mapping.py > $R/$OUT/mapping_measure.json 2>/dev/null
db=$(find /tmp/fm_trace2 -name '*.db' | head -1); [ -n "$db" ] && python $R/tools/rocpd_summary.py $db > $R/$OUT/mapping_trace.txt
cd $R
# batch-1 gap table (round 5): per op of one 720 x 720 map / one detector frame against the batch rates + the launch timeline
cd /tmp; rm -rf /tmp/fm_tl
rocprofv3 --kernel-trace -d /tmp/fm_tl -- python $R/bench.py --batch 1 --size 720 --steps 30 --warmup 5 --no-cpu-baseline --no-probe --also "" --traffic none --configs "" > /tmp/fm_tl.log 2>&1
db=$(find /tmp/fm_tl -name '*.db' | head -1)
cd $R
python tools/gap_b1.py $OUT/gap_b1.json ${db:+--timeline-db $db --timeline-forwards 20} > $OUT/gap_b1_summary.txt 2>/dev/null
python tools/step_profile.py 60 --fine 2>/dev/null | grep "^{" > $OUT/step_profile.json
bash tools/detector_timeline.sh $OUT > /dev/null 2>&1
bash tools/pred_timeline.sh $OUT 1 720 > /dev/null 2>&1
bash tools/pipeline_timeline.sh $OUT > /dev/null 2>&1
tools/pmc_passes.sh $OUT/pmc_fp32 --precision fp32
tools/pmc_passes.sh $OUT/pmc_bf16x6 --precision bf16x6
tools/pmc_passes.sh $OUT/pmc_fp16x3 --precision fp16x3
ls $OUT
