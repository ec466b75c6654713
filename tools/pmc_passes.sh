#!/bin/bash
# rocprofv3 PMC passes over a short bench.py run (one pass per counter group, --kernel-trace only, as
# MI355X_MICROARCH.md prescribes); summaries via tools/rocpd_summary.py.
#   tools/pmc_passes.sh <outdir> <bench args...>
set -u
OUT=$1; shift
mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
run() {  # name, counters...
  local name=$1; shift
  rm -rf /tmp/pmc_$name
  timeout 600 rocprofv3 --kernel-trace --pmc "$@" -d /tmp/pmc_$name -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-probe --also "" --traffic none --configs "" $BENCH_ARGS > /tmp/pmc_$name.log 2>&1
  local db=$(find /tmp/pmc_$name -name '*.db' | head -1)
  if [ -n "$db" ]; then python $REPO/tools/rocpd_summary.py "$db" > "$REPO/$OUT/pmc_$name.txt" 2>&1; else tail -5 /tmp/pmc_$name.log > "$REPO/$OUT/pmc_$name.txt"; fi
}
BENCH_ARGS="$*"
run sq SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcc TCC_HIT_sum TCC_MISS_sum
if [ "${PMC_EXTRA:-0}" = "1" ]; then
  run lds SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU GRBM_GUI_ACTIVE
  run valu SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE
fi
