#!/bin/bash
# Attribution of SQ_LDS_BANK_CONFLICT in k_cheb_mg_gemm: builds variants of libp2m_hip.so with ONE class of
# LDS accesses compiled out / linearised (P2M_ABL_* in csrc/chebtile.hip) into pose2mesh_release_amd/lib/abl/, then (on the GPU
# box, `run` argument) one rocprofv3 --pmc pass per variant over the finest-level 128 -> 128 forward launch of the f16x2 kernel.
#   bash tools/lds_conflict_ablation.sh build      (here: hipcc cross-compiles)
#   bash tools/lds_conflict_ablation.sh run <tag>  (GPU box)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
L=$R/pose2mesh_release_amd/lib
VARIANTS="NO_XU NO_P0 NO_CONV LINEAR_X LINEAR_LT LINEAR_A"
if [ "$1" = build ]; then
  mkdir -p $L/abl
  for v in $VARIANTS; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c -fno-slp-vectorize -DP2M_ABL_$v -o $L/abl/chebtile_$v.o $R/pose2mesh_release_amd/csrc/chebtile.hip &
  done
  wait
  for v in $VARIANTS; do
    objs=$(ls $L/obj/*.o | grep -v chebtile.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $L/abl/libp2m_hip_$v.so $objs $L/abl/chebtile_$v.o
    rm -f $L/abl/chebtile_$v.o
  done
  ls -la $L/abl
else
  T=$2
  CTR="SQ_INSTS_LDS SQ_INSTS_MFMA SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"
  for v in BASE $VARIANTS; do
    lib=$L/libp2m_hip.so; [ $v != BASE ] && lib=$L/abl/libp2m_hip_$v.so
    P2M_HIP_LIB=$lib P2M_GEMM_ARITH=f16x2 PROBE_CASE=0,128,128,0 PROBE_ONLY_TILE=1 bash $R/tools/rocprof_pmc.sh ${T}_abl_$v "$CTR" python $R/tools/probes/tile_gemm_probe.py > /dev/null 2>&1
    echo "== $v"; grep "k_cheb_mg_gemm" $R/gpurun_out/${T}_abl_${v}_pmc.csv | cut -d, -f2-6
  done
fi
