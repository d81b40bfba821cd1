#!/bin/bash
# PMC passes over the fused-MLP forward kernel (separate --pmc runs, no trace domains):  mlp_pmc.sh lib rows c outdir
lib=$1; rows=$2; c=$3; out=$4; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set -d $out/raw$i -o p -- python $R/tools/ubench/mlp_time.py $lib $rows $c > $out/raw$i.log 2>&1
  db=$(ls $out/raw$i/*results.db 2>/dev/null | head -1)
  [ -n "$db" ] && python $R/tools/pmc_kernel.py $db mlp_fwd
done
