# On the GPU box: PMC passes of the config-5 scene (1 M spheres + 262 K triangles, SAH tree, BVH in HBM), 2048x2048.
# FETCH_SIZE and WRITE_SIZE in passes of their own (MI355X_MICROARCH.md: TCC slots), SQ groups apart.
# usage: bash scripts/pmc_c5.sh <tag> [spp]
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
TAG=${1:-C5}; SPP=${2:-16}
python3 -c "import sys; sys.path.insert(0,'.'); import importlib.util as u; sp=u.spec_from_file_location('b','ray-tracer-archive_amd/build.py'); b=u.module_from_spec(sp); sp.loader.exec_module(b); print(b.source_hash())" > gpurun_out/pmc${TAG}_source_hash.txt
for i in 1 2 3 4; do
  case $i in
    1) C="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAIT_ANY";;
    2) C="FETCH_SIZE";;
    3) C="WRITE_SIZE";;
    4) C="TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE SQ_INSTS_VMEM SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU";;
  esac
  timeout -k 10 300 rocprofv3 --pmc $C --output-format csv -d gpurun_out/pmc${TAG}$i -o run -- python3 scripts/gpu_c5.py $SPP > gpurun_out/pmc${TAG}$i.log 2>&1 || exit 1
done
python3 scripts/pmc_summary.py gpurun_out/pmc${TAG}1 gpurun_out/pmc${TAG}2 gpurun_out/pmc${TAG}3 gpurun_out/pmc${TAG}4
