#!/bin/bash
# SQ-counter passes of the persistent CG kernel itself (what binds an iteration: VALU issue, parked waves, issue stalls, LDS conflicts).
# One --pmc group per pass, --kernel-trace only, the program directly after "--" (gpurun refuses anything else).
#   gpurun -- 'bash scripts/profile_sq.sh r06 2048'    -> gpurun_out/prof/<tag>_cg_persist1_<n>_sq_counters.txt + sq_counters.json (copy both into profiles/)
R=${GRAFT_REPO_ROOT:?run through gpurun}
TAG=${1:-rXX}
N=${2:-2048}
OUT=$R/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/${TAG}_rocprofv3_counter_list.txt 2>&1
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU"
P2="SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS"
P3="SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CU_CYCLES SQ_CYCLES"
P4="GRBM_GUI_ACTIVE GRBM_COUNT"
# L2 (TCC) passes: hit rate of the XCDs' L2s and the share of their misses that goes to DRAM (the rest is served by the Infinity Cache)
P5="TCC_HIT_sum TCC_MISS_sum"
P6="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum"
i=0
for P in "$P1" "$P2" "$P3" "$P4" "$P5" "$P6"; do
  i=$((i+1))
  rm -rf /tmp/sq_$i
  rocprofv3 --kernel-trace --pmc $P --output-format csv -d /tmp/sq_$i -o sq -- python3 $R/scripts/bench_cg.py $N > $OUT/${TAG}_sq_pass${i}_run.log 2>&1
  echo "pass $i rc $?"
done
python3 $R/scripts/summarize_sq.py $N /tmp/sq_1 /tmp/sq_2 /tmp/sq_3 /tmp/sq_4 /tmp/sq_5 /tmp/sq_6 --json $OUT/sq_counters.json > $OUT/${TAG}_cg_persist1_${N}_sq_counters.txt 2>&1
cat $OUT/${TAG}_cg_persist1_${N}_sq_counters.txt
