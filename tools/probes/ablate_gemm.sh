#!/bin/bash
# What is the chunk time of k_gemm_planes_ws made of?  Builds libp2m_hip.so variants with parts of the kernel removed
# (-DP2M_ABLATE=mask: 1 A loads, 2 B loads, 4 slice arithmetic, 8 LDS stores, 16 fragment reads, 32 MFMAs; results are
# wrong by construction) and times them on two layer shapes.
#   tools/probes/ablate_gemm.sh build     (here: hipcc cross-compiles, ~2 min, writes build_ablate/lib_<mask>.so)
#   tools/probes/ablate_gemm.sh run       (on the GPU box, e.g. through gpurun)
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
MASKS="0 1 2 3 4 8 16 32"
if [ "${1:-run}" = build ]; then
  mkdir -p "$REPO/build_ablate"
  cd "$REPO/pose2mesh_release_amd/csrc" || exit 1
  for m in $MASKS; do
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DP2M_ABLATE=$m -o "$REPO/build_ablate/lib_$m.so" \
      capi.hip basis.hip gemm.hip bn.hip optim.hip fused.hip loss.hip &
  done
  wait
  exit 0
fi
export PROBE_MODES=bf16x3 P2M_GEMM_WS=${P2M_GEMM_WS:-2}
for m in $MASKS; do
  for sh in 5888,128,128 2944,256,256; do
    P2M_HIP_LIB="$REPO/build_ablate/lib_$m.so" PROBE_SHAPE=$sh timeout 100 python "$REPO/tools/probes/gemm_probe.py" 2>&1 |
      tail -1 | cut -c1-90 | sed "s/^/mask=$m /"
  done
done
