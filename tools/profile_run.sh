#!/bin/bash
# tools/profile_run.sh ROUND -- on the GPU box: kernel-trace stats, the HBM-traffic PMC passes and the SQ counter passes of
# the default bench.py command; raw output lands in gpurun_out/prof_ROUND/ (tools/make_profile_summary.py copies the
# summaries into profiles/).  Each --pmc set is its own run, never combined with a trace (MI355X_MICROARCH.md, rocprofv3).
R=$1
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$R
rm -rf $OUT; mkdir -p $OUT
B="$GRAFT_REPO_ROOT/bench.py --no-cpu --no-natural --no-copy --no-configs $BENCH_ARGS"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $B --steps 10 --warmup 3 > $OUT/bench_under_rocprof.log 2>&1
pmc() { # name counters...
  local name=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 $B --steps 2 --warmup 1 --no-events > $OUT/$name.log 2>&1
}
pmc pmc_fetch FETCH_SIZE
pmc pmc_write WRITE_SIZE
pmc pmc_l2 TCC_HIT_sum TCC_MISS_sum
pmc sq_a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS
pmc sq_b SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SALU
pmc sq_d SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES SQ_WAVES SQ_BUSY_CU_CYCLES
pmc sq_c SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE SQ_INSTS_SMEM
python3 $GRAFT_REPO_ROOT/bench.py --no-cpu > $OUT/bench.json 2> $OUT/bench.err
cat $OUT/bench.json
find $OUT -name "*kernel_stats.csv" | head -1 | xargs head -3 | cut -c1-200
