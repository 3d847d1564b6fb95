cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
P2="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"
i=0
for cfg in "0 16 32 512 64 2" "0 64 32 512 64 2" "1 128 32 512 64 2"; do
  set -- $cfg; i=$((i+1))
  USSEG_BIG=$1 rocprofv3 --pmc $P1 -d gpurun_out/pmcA$i -o r --output-format csv -- python3 tools/pmc_conv.py $2 $3 $4 $5 $6 > gpurun_out/pmcA$i.log 2>&1
  USSEG_BIG=$1 rocprofv3 --pmc $P2 -d gpurun_out/pmcB$i -o r --output-format csv -- python3 tools/pmc_conv.py $2 $3 $4 $5 $6 > gpurun_out/pmcB$i.log 2>&1
done
ls gpurun_out/pmcA1
