set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r1a
mkdir -p $O
rocprofv3 -L > $O/counters.txt 2>&1 || true
grep -c "" $O/counters.txt
( cd $R && rocprofv3 --kernel-trace --stats -d $O/trace -o bench -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_trace.log 2>&1 )
tail -2 $O/bench_trace.log
( cd $R && rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o bench -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-probe > $O/bench_fetch.log 2>&1 )
( cd $R && rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o bench -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-probe > $O/bench_write.log 2>&1 )
( cd $R && rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE -d $O/pmc_sq -o bench -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-probe > $O/bench_sq.log 2>&1 )
tail -3 $O/bench_sq.log
find $O -type f | head -40
du -sh $O
