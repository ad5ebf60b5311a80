export SMG_EXPECT_GPU=1
python -m pytest tests/test_gpu_dist.py -x -q 2>&1 | tail -15
python -m pytest tests/test_gpu_parity.py -x -q -k "allreduce or split_phase or speculative" 2>&1 | tail -5
python bench.py --steps 300 --warmup 50 --no-cpu > gpurun_out/b1.json 2> gpurun_out/b1.err; tail -c 3000 gpurun_out/b1.json | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step']); print(json.dumps(d['c4_k64_sharded'], indent=1)); print(json.dumps(d['smoothers'], indent=1))"
tail -3 gpurun_out/b1.err
echo "=== 2 ranks on one GPU over gloo"
SMG_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29655 bench.py --gpus 2 --steps 100 --warmup 20 > gpurun_out/b2.json 2> gpurun_out/b2.err; tail -n 1 gpurun_out/b2.json | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['n_gpus']); print(json.dumps(d['c4_k64_sharded'], indent=1))"
tail -5 gpurun_out/b2.err
echo "=== force split, RCCL world 1"
SMG_BENCH_FORCE_SPLIT=1 python bench.py --steps 300 --warmup 50 --no-cpu 2> gpurun_out/b3.err | tail -n 1 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['allreduce'])"
tail -3 gpurun_out/b3.err
