#!/bin/bash
# Resource ablations of k_gemm_tn_ws (probe builds; results are WRONG numbers, only the time is read):
#   bash tools/tn_ablate.sh build    -> lib/abl/libp2m_hip_TN_{NOMFMA,NOSLICE,NOLOAD}.so
#   bash tools/tn_ablate.sh run      -> time of the finest level's weight gradient under each
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
L=$R/pose2mesh_release_amd/lib
if [ "$1" = build ]; then
  mkdir -p $L/abl
  objs=$(ls $L/obj/*.o | grep -v "/gemm.o")
  for v in NOMFMA NOSLICE NOLOAD "NOMFMA -DP2M_TN_ABL_NOSLICE"; do
    tag=$(echo $v | tr -d ' ' | sed 's/-DP2M_TN_ABL_/_/g')
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c -DP2M_TN_ABL_$v -o $L/abl/gemm_$tag.o $R/pose2mesh_release_amd/csrc/gemm.hip 2>/dev/null || exit 1
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $L/abl/libp2m_hip_TN_$tag.so $objs $L/abl/gemm_$tag.o && rm -f $L/abl/gemm_$tag.o
  done
  ls $L/abl
else
  for B in 64 256; do
    for lib in "" abl/libp2m_hip_TN_NOMFMA.so abl/libp2m_hip_TN_NOSLICE.so abl/libp2m_hip_TN_NOLOAD.so abl/libp2m_hip_TN_NOMFMA_NOSLICE.so; do
      echo -n "${lib:-product} : "
      P2M_HIP_LIB=${lib:+$L/$lib} P2M_GEMM_ARITH=bf16x3 python $R/tools/probes/tn_trace_probe.py 0 128 128 $B 2>&1 | grep "^B="
    done
  done
fi
