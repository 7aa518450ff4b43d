#!/bin/bash
# Same-box A/B of two builds of the HIP library on the tile-kernel probe (tools/probes/tile_gemm_probe.py):
#   bash tools/ab_probe.sh [all|finest] LIB_A LIB_B ...   (paths relative to pose2mesh_release_amd/lib)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
which=$1; shift
for rep in 1 2; do
  for lib in "$@"; do
    echo "== $lib (rep $rep)"
    P2M_HIP_LIB=$R/pose2mesh_release_amd/lib/$lib P2M_GEMM_ARITH=${P2M_GEMM_ARITH:-bf16x3} PROBE_ONLY_TILE=1 python $R/tools/probes/tile_gemm_probe.py $which 2>&1 | grep -v amdgpu | grep -E "TOTAL|plan" | sed 's/basis+gemm   0.000 ms | //'
  done
done
