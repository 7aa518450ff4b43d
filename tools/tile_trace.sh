#!/bin/bash
# Probe build of csrc/chebtile.hip with phase stamps (-DP2M_TILE_TRACE=<logical block id>) and its run: see
# tools/probes/tile_trace_probe.py.   bash tools/tile_trace.sh build | run
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
L=$R/pose2mesh_release_amd/lib
if [ "$1" = build ]; then
  mkdir -p $L/abl
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c -fno-slp-vectorize -DP2M_TILE_TRACE=${2:-1500} -o $L/abl/chebtile_TRACE.o $R/pose2mesh_release_amd/csrc/chebtile.hip || exit 1
  objs=$(ls $L/obj/*.o | grep -v chebtile.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $L/abl/libp2m_hip_TRACE.so $objs $L/abl/chebtile_TRACE.o && rm -f $L/abl/chebtile_TRACE.o
  ls -la $L/abl
else
  for form in fwd planes; do
    P2M_HIP_LIB=$L/abl/libp2m_hip_TRACE.so P2M_GEMM_ARITH=${P2M_GEMM_ARITH:-bf16x3} PROBE_FORM=$form python $R/tools/probes/tile_trace_probe.py 2>&1 | grep -v amdgpu
  done
fi
