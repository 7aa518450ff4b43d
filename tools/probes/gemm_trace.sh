#!/bin/bash
# Builds the s_memtime-instrumented probe library (pose2mesh_release_amd/lib/libp2m_hip_trace.so, -DP2M_GEMM_TRACE=1,
# optionally -DP2M_PRODUCER_PRIO=n) next to the product library.  Run on the GPU box with
#   PROBE_SHAPE=2944,256,256 P2M_HIP_LIB=$GRAFT_REPO_ROOT/pose2mesh_release_amd/lib/libp2m_hip_trace.so \
#       python tools/probes/gemm_trace.py
set -e
REPO=$(cd "$(dirname "$0")/../.." && pwd)
python -m pose2mesh_release_amd.build > /dev/null
T=$(mktemp -d)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c -DP2M_GEMM_TRACE=1 ${P2M_TRACE_FLAGS:-} -o $T/gemm.o "$REPO/pose2mesh_release_amd/csrc/gemm.hip"
O="$REPO/pose2mesh_release_amd/lib/obj"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$REPO/pose2mesh_release_amd/lib/libp2m_hip_trace.so" $T/gemm.o $O/capi.o $O/basis.o $O/bn.o $O/optim.o $O/fused.o $O/loss.o
echo "$REPO/pose2mesh_release_amd/lib/libp2m_hip_trace.so"
