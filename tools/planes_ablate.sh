#!/bin/bash
# Resource ablations of k_gemm_planes_ws (probe builds; the results are WRONG numbers, only the time is read):
#   bash tools/planes_ablate.sh build | run [level Ka N B]
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
L=$R/pose2mesh_release_amd/lib
V="NOMFMA NOSLICE NOLOAD NOSTORE NOMFMA_NOSLICE NOMFMA_NOSLICE_NOSTORE NOLOAD_NOSTORE"
if [ "$1" = build ]; then
  mkdir -p $L/abl
  objs=$(ls $L/obj/*.o | grep -v "/gemm.o")
  for v in $V; do
    defs=$(echo $v | sed 's/_/ /g' | sed 's/\([A-Z]*\)/-DP2M_PL_ABL_\1/g')
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c $defs -o $L/abl/gemm_$v.o $R/pose2mesh_release_amd/csrc/gemm.hip 2>/dev/null || exit 1
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $L/abl/libp2m_hip_PL_$v.so $objs $L/abl/gemm_$v.o && rm -f $L/abl/gemm_$v.o
  done
  ls $L/abl
else
  for v in product $V; do
    lib=$L/abl/libp2m_hip_PL_$v.so; [ $v = product ] && lib=
    echo "== $v"
    P2M_HIP_LIB=$lib P2M_GEMM_ARITH=bf16x3 python $R/tools/probes/planes_abl_probe.py "${@:2}" 2>&1 | grep "^B="
  done
fi
