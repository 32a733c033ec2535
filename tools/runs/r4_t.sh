cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/r04; mkdir -p $out
for v in 0 1 0 1; do
  WS_SLAB_EXACT_ONE_RANK=$v WS_BENCH_FORCE_SLAB=1 timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-ieee --no-readback --no-north-star > $out/bench_slab_one_rank_waits_$v.json 2> $out/bench_slab_one_rank_waits_$v.err
  python3 -c "
import json; d=json.load(open('$out/bench_slab_one_rank_waits_$v.json')); print('waits=$v', d['stats'].get('size_waits'), round(d['ms_per_step'],4), [round(x,4) for x in d['repetitions']['ms_per_step']], round(d['settled']['ms_per_step'],4))"
done
