# usage: bash scripts/pmc_passes.sh <tag> <scene> <W> <H> <spp>   (on the GPU box; one rocprofv3 --pmc pass per counter group)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
TAG=${1:-A}; SCENE=${2:-book1}; W=${3:-1200}; H=${4:-800}; SPP=${5:-500}
python3 -c "import sys; sys.path.insert(0,'.'); import importlib.util as u; sp=u.spec_from_file_location('b','ray-tracer-archive_amd/build.py'); b=u.module_from_spec(sp); sp.loader.exec_module(b); print(b.source_hash())" > gpurun_out/pmc${TAG}_source_hash.txt
for i in 1 2 3 4 5; do
  case $i in
    1) C="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY";;
    2) C="SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_WAIT_ANY";;
    3) C="SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_CYCLES SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F64";;
    4) C="FETCH_SIZE";;
    5) C="WRITE_SIZE";;
  esac
  timeout -k 10 200 rocprofv3 --pmc $C --output-format csv -d gpurun_out/pmc${TAG}$i -o run -- python3 scripts/gpu_render_once.py $SCENE $W $H $SPP 1 > gpurun_out/pmc${TAG}$i.log 2>&1 || exit 1
done
python3 scripts/pmc_summary.py gpurun_out/pmc${TAG}1 gpurun_out/pmc${TAG}2 gpurun_out/pmc${TAG}3 gpurun_out/pmc${TAG}4 gpurun_out/pmc${TAG}5
