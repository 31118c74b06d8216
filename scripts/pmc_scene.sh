# On the GPU box: PMC passes of one scene/size (like pmc_passes.sh, any host scene): bash scripts/pmc_scene.sh <tag> <scene> <W> <H> <spp>
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
TAG=$1; SCENE=$2; W=$3; H=$4; SPP=$5
python3 -c "import sys; sys.path.insert(0,'.'); import importlib.util as u; sp=u.spec_from_file_location('b','ray-tracer-archive_amd/build.py'); b=u.module_from_spec(sp); sp.loader.exec_module(b); print(b.source_hash())" > gpurun_out/pmc${TAG}_source_hash.txt
for i in 1 2 3 4; do
  case $i in
    1) C="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY";;
    2) C="SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_WAIT_ANY";;
    3) C="FETCH_SIZE";;
    4) C="WRITE_SIZE";;
  esac
  timeout -k 10 200 rocprofv3 --pmc $C --output-format csv -d gpurun_out/pmc${TAG}$i -o run -- python3 scripts/gpu_render_once.py $SCENE $W $H $SPP 1 > gpurun_out/pmc${TAG}$i.log 2>&1 || exit 1
done
