cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
for i in 1 2 3; do
  case $i in
    1) C="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY";;
    2) C="SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_WAIT_ANY";;
    3) C="SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_LEVEL_WAVES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_CYCLES";;
  esac
  timeout -k 10 200 rocprofv3 --pmc $C -d gpurun_out/pmcA$i -o run -- python3 scripts/gpu_render_once.py book1 1200 800 500 1 > gpurun_out/pmcA$i.log 2>&1 || exit 1
done
python3 scripts/pmc_summary.py gpurun_out/pmcA1 gpurun_out/pmcA2 gpurun_out/pmcA3
