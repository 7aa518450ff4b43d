mkdir -p gpurun_out
R=$(pwd)
python -m pytest tests/test_gpu_parity_full.py tests/test_gpu_train.py tests/test_gpu_boundary.py -m gpu -q -x -k "full_gradients or bitwise or three_adam or graphed or boundary or algebraic" > gpurun_out/c5_parity.log 2>&1; echo "parity rc=$?"; tail -6 gpurun_out/c5_parity.log
Q="--no-cpu-baseline --no-kernel-timing --no-arith-ab --also none --steps 30 --warmup 10"
python bench.py $Q > gpurun_out/c5_bench.json 2> gpurun_out/c5_bench.err; python -c "
import json;j=json.load(open('gpurun_out/c5_bench.json'));print('bench',j['value'],j['ms_per_step'],j['ms_per_step_stats']['median'])"
python bench.py --joint-set mano --batch 64 $Q > gpurun_out/c5_mano64.json 2>> gpurun_out/c5_bench.err; python -c "
import json;j=json.load(open('gpurun_out/c5_mano64.json'));print('mano64',j['value'],j['ms_per_step'],j['ms_per_step_stats']['median'])"
bash tools/trace_step.sh c5 > gpurun_out/c5_trace.out 2>&1; grep "^# [0-9]* launches\|^# busy\|^# idle" gpurun_out/c5_step_trace.txt
PROBE_CASE=0,128,128,0 bash tools/rocprof_pmc.sh r04_b_pmc_tn_a "SQ_INSTS_MFMA SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" python $R/tools/probes/tile_gemm_probe.py > /dev/null 2>&1
grep "tn_ws" gpurun_out/r04_b_pmc_tn_a_pmc.csv
