export SMG_EXPECT_GPU=1
python -m pytest tests -m gpu -q -x 2>&1 | grep -v "^\[W\|amdgpu.ids\|Gloo" | tail -4
python bench.py --steps 400 --warmup 50 > gpurun_out/b_full.json 2> gpurun_out/b_full.err; tail -2 gpurun_out/b_full.err
SMG_BENCH_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29671 bench.py --gpus 2 --steps 100 --warmup 20 > gpurun_out/b_g2.json 2> gpurun_out/b_g2.err; tail -2 gpurun_out/b_g2.err
