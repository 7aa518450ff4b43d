#!/bin/bash
# usage (on the GPU box, from the repo root): bash tools/profile_round.sh <round tag, e.g. r02>
# Produces under gpurun_out/: full -m gpu test log + parity maxima, the default bench line (train, with CPU baseline),
# the inference and MANO bench lines, rocprofv3 kernel-trace stats of the train and inference benches, the HBM-traffic
# passes, and PMC passes (with kernel durations and derived clock) of the two dominant kernels.
set -u
T=$1
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
# round 5: bench.py's main measurement is the exact fp32 emulation (bf16x3); f16x2 (the package's fast mode) is measured beside
# it (`arith_ab`, `also.*.fast_mode_f16x2`, ${T}_bench_f16x2.json).  SKIP_TESTS=1 skips the 8-minute -m gpu run.
if [ "${SKIP_TESTS:-0}" != 1 ]; then
  python -m pytest tests -m gpu -q --durations=10 > gpurun_out/${T}_gputests.log 2>&1
  tail -4 gpurun_out/${T}_gputests.log
  cp gpurun_out/parity_maxima.json gpurun_out/${T}_parity_maxima.json 2>/dev/null
fi
python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
python bench.py --mode infer > gpurun_out/${T}_bench_infer.json 2>> gpurun_out/${T}_bench.err
python bench.py --joint-set mano --batch 512 --no-cpu-baseline --no-arith-ab > gpurun_out/${T}_bench_mano.json 2>> gpurun_out/${T}_bench.err
python bench.py --optimizer rmsprop --no-cpu-baseline --no-arith-ab --steps 10 --warmup 5 > gpurun_out/${T}_bench_rmsprop.json 2>> gpurun_out/${T}_bench.err
python bench.py --arith f16x2 --no-cpu-baseline --no-arith-ab --also none > gpurun_out/${T}_bench_f16x2.json 2>> gpurun_out/${T}_bench.err
for f in bench bench_infer bench_mano bench_rmsprop bench_f16x2; do python -c "
import json;j=json.load(open('gpurun_out/${T}_$f.json'));print('$f',j['value'],j['ms_per_step'],j.get('roofline',{}).get('frac'),j.get('roofline_sparse',{}).get('frac'),j.get('cpu_baseline',{}).get('value'))"; done
bash tools/rocprof_stats.sh ${T}_train --steps 6 --warmup 3 --no-kernel-timing --no-cpu-baseline --no-arith-ab --also none > /dev/null 2>&1
bash tools/rocprof_stats.sh ${T}_infer --mode infer --steps 20 --warmup 5 --no-kernel-timing --no-cpu-baseline > /dev/null 2>&1
bash tools/trace_step.sh ${T} > gpurun_out/${T}_trace.out 2>&1
bash tools/rocprof_traffic.sh ${T} > gpurun_out/${T}_traffic.out 2>&1
tail -12 gpurun_out/${T}_traffic.out
PROBE_SHAPE=5888,128,128 PROBE_MODES=bf16x3 bash tools/rocprof_pmc.sh ${T}_pmc_gemm_a "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" python $R/tools/probes/gemm_probe.py > /dev/null 2>&1
PROBE_SHAPE=5888,128,128 PROBE_MODES=bf16x3 bash tools/rocprof_pmc.sh ${T}_pmc_gemm_b "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_MFMA SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE" python $R/tools/probes/gemm_probe.py > /dev/null 2>&1
# the basis-inside-the-contraction kernel (finest level, 128 -> 128, forward form): MFMA busy / VALU / waits, LDS conflicts,
# HBM bytes
for spec in "a:SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
            "b:SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_MFMA SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE" \
            "c:FETCH_SIZE GRBM_GUI_ACTIVE" "d:WRITE_SIZE GRBM_GUI_ACTIVE"; do
  P2M_GEMM_ARITH=bf16x3 PROBE_CASE=0,128,128,0 PROBE_ONLY_TILE=1 bash tools/rocprof_pmc.sh ${T}_pmc_tile_${spec%%:*} "${spec#*:}" python $R/tools/probes/tile_gemm_probe.py > /dev/null 2>&1
done
# ... and the fast mode's matrix-core-gather kernel: LDS counters (round 5: SQ_LDS_BANK_CONFLICT 24.1 M -> 0)
P2M_GEMM_ARITH=f16x2 PROBE_CASE=0,128,128,0 PROBE_ONLY_TILE=1 bash tools/rocprof_pmc.sh ${T}_pmc_mg_f16x2_b "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_MFMA SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE" python $R/tools/probes/tile_gemm_probe.py > /dev/null 2>&1
# the weight-gradient contraction of the same conv (finest level, 128 x 3*128): MFMA / VALU / LDS counters
P2M_GEMM_ARITH=bf16x3 PROBE_CASE=0,128,128,0 bash tools/rocprof_pmc.sh ${T}_pmc_tn_a "SQ_INSTS_MFMA SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" python $R/tools/probes/tile_gemm_probe.py > /dev/null 2>&1
bash tools/rocprof_pmc.sh ${T}_pmc_basis_a "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES" python $R/tools/probes/basis_probe.py finest > /dev/null 2>&1
bash tools/rocprof_pmc.sh ${T}_pmc_basis_b "FETCH_SIZE GRBM_GUI_ACTIVE" python $R/tools/probes/basis_probe.py finest > /dev/null 2>&1
bash tools/rocprof_pmc.sh ${T}_pmc_basis_c "WRITE_SIZE GRBM_GUI_ACTIVE" python $R/tools/probes/basis_probe.py finest > /dev/null 2>&1
# in-kernel phase / per-wave stamps of the tile kernel (probe build made by `bash tools/tile_trace.sh build` before the call)
if [ -f pose2mesh_release_amd/lib/abl/libp2m_hip_TRACE.so ]; then bash tools/tile_trace.sh run > gpurun_out/${T}_tile_phase_trace.txt 2>&1; fi
P2M_GEMM_ARITH=bf16x3 python tools/probes/tile_gemm_probe.py all > gpurun_out/${T}_probe_tile.txt 2>&1
grep -h "gemm_planes_ws\|k_basis_tile\|k_cheb_tile_gemm" gpurun_out/${T}_pmc_*.csv | cut -c1-60,150-260
