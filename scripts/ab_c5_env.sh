# config-5 stand-in under both settings of an upload-time switch: bash scripts/ab_c5_env.sh VAR [spp]
for v in 0 1 0 1; do
  env $1=$v timeout -k 10 300 python3 scripts/gpu_c5_count.py ${2:-16} 2>&1 | sed -n 1p | sed "s/^/$1=$v /"
  env $1=$v timeout -k 10 300 python3 scripts/gpu_c5.py ${2:-32} 2>&1 | sed -n 2p | sed "s/^/$1=$v /"
done
