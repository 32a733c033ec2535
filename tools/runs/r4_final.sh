cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/r04; mkdir -p $out
timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu > $out/pytest_gpu.log 2>&1
echo "pytest exit $?"; tail -5 $out/pytest_gpu.log
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke exit $?"; tail -1 $out/smoke.log
timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_final.json 2> $out/bench_final.err; echo "bench exit $?"
python3 -c "
import json; d=json.load(open('gpurun_out/r04/bench_final.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['settled']['ms_per_step'], d['cpu_baseline']['value'])"
