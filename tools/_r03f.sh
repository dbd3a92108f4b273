cd $GRAFT_REPO_ROOT
for s in 60 120; do
echo "=== $s"
python tools/pk_probe.py $s 16384
bash tools/pmc_any.sh preproc_fused "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU" -- python $GRAFT_REPO_ROOT/tools/pk_probe.py $s 16384
bash tools/pmc_any.sh preproc_fused "SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_WAIT_INST_ANY" -- python $GRAFT_REPO_ROOT/tools/pk_probe.py $s 16384
bash tools/pmc_any.sh preproc_fused "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_SMEM SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES" -- python $GRAFT_REPO_ROOT/tools/pk_probe.py $s 16384
done
