#!/bin/bash
# Probe build of csrc/gemm.hip with phase stamps in k_gemm_tn_ws (-DP2M_TN_TRACE=<block id>) and its run: see
# tools/probes/tn_trace_probe.py.   bash tools/tn_trace.sh build [block] | run
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
L=$R/pose2mesh_release_amd/lib
if [ "$1" = build ]; then
  mkdir -p $L/abl
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c -DP2M_TN_TRACE=${2:-1000} -o $L/abl/gemm_TNTRACE.o $R/pose2mesh_release_amd/csrc/gemm.hip || exit 1
  objs=$(ls $L/obj/*.o | grep -v "/gemm.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $L/abl/libp2m_hip_TNTRACE.so $objs $L/abl/gemm_TNTRACE.o && rm -f $L/abl/gemm_TNTRACE.o
  ls -la $L/abl
else
  P2M_HIP_LIB=$L/abl/libp2m_hip_TNTRACE.so P2M_GEMM_ARITH=${P2M_GEMM_ARITH:-bf16x3} python $R/tools/probes/tn_trace_probe.py "${@:2}" 2>&1 | grep -v amdgpu
fi
