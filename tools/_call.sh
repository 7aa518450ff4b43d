mkdir -p gpurun_out
Q="--no-cpu-baseline --no-kernel-timing --no-arith-ab --also none --steps 30 --warmup 10"
for v in 0 1 0 1; do P2M_LIFTER_STREAM=$v python bench.py $Q > gpurun_out/c3_lift_$v.json 2> gpurun_out/c3_lift_$v.err; python -c "
import json;j=json.load(open('gpurun_out/c3_lift_$v.json'));print('LIFTER_STREAM=$v',j['value'],j['ms_per_step'],j['ms_per_step_stats']['median'])"; done
python -m pytest tests -m gpu -q --durations=5 > gpurun_out/c3_gputests.log 2>&1; echo "pytest rc=$?" ; tail -12 gpurun_out/c3_gputests.log
bash tools/trace_step.sh c3 > gpurun_out/c3_trace.out 2>&1
