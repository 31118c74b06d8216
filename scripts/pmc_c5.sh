cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAIT_ANY --output-format csv -d gpurun_out/pmcC5 -o run -- python3 scripts/gpu_c5.py 32 > gpurun_out/pmcC5.log 2>&1
python3 scripts/pmc_summary.py gpurun_out/pmcC5 | grep -E "k_extend|k_shade" | cut -c1-500
