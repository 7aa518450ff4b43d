mkdir -p gpurun_out
python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "activation_on_load" > gpurun_out/c4_ops.log 2>&1; echo "ops rc=$?"; tail -25 gpurun_out/c4_ops.log
python -m pytest tests/test_gpu_parity_full.py tests/test_gpu_parity.py tests/test_gpu_train.py -m gpu -q -k "not baseline_sizes_default" > gpurun_out/c4_parity.log 2>&1; echo "parity rc=$?"; tail -12 gpurun_out/c4_parity.log
Q="--no-cpu-baseline --no-kernel-timing --no-arith-ab --also none --steps 30 --warmup 10"
for v in 0 1 0 1; do P2M_FOLD_ACT=$v python bench.py $Q > gpurun_out/c4_fold_$v.json 2> gpurun_out/c4_fold_$v.err; python -c "
import json;j=json.load(open('gpurun_out/c4_fold_$v.json'));print('FOLD_ACT=$v',j['value'],j['ms_per_step'],j['ms_per_step_stats']['median'])"; done
