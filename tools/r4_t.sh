cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/r04p; mkdir -p $out
WS_BENCH_FORCE_SLAB=1 timeout -k 10 400 python3 bench.py --no-cpu-baseline > $out/bench_c3_cloud_slab_one_rank.json 2> $out/bench_c3_cloud_slab_one_rank.err; echo "one-rank slab exit $?"
WS_BENCH_BACKEND=gloo timeout -k 10 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29571 bench.py --gpus 2 --steps 20 --warmup 5 --reps 1 > $out/bench_gloo_rehearsal_2ranks.json 2> $out/bench_gloo_rehearsal_2ranks.err; echo "gloo rehearsal exit $?"
WS_BENCH_BACKEND=gloo timeout -k 10 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29572 bench.py --gpus 2 --steps 20 --warmup 5 --reps 1 --lagged-messages > $out/bench_gloo_rehearsal_2ranks_lagged.json 2> $out/bench_gloo_rehearsal_2ranks_lagged.err; echo "gloo rehearsal lagged exit $?"
python3 - <<'P'
import json
for f in ("bench_c3_cloud_slab_one_rank","bench_gloo_rehearsal_2ranks","bench_gloo_rehearsal_2ranks_lagged"):
    d=json.load(open("gpurun_out/r04p/%s.json"%f)); print(f, round(d["ms_per_step"],3), d.get("settled",{}).get("ms_per_step"), d.get("messages_rank0"))
P
