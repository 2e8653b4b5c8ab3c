set -x
mkdir -p gpurun_out/r2a
python -m pytest tests -m gpu -x -q > gpurun_out/r2a/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2a/pytest.log
tail -5 gpurun_out/r2a/pytest.log
timeout -k 10 600 python bench.py --steps 3 --warmup 1 > gpurun_out/r2a/bench1.json 2> gpurun_out/r2a/bench1.err; echo "bench rc=$?"
tail -c 3000 gpurun_out/r2a/bench1.json
timeout -k 10 300 python bench.py --gpus 2 --nuclei 20000 --steps 2 --warmup 1 > gpurun_out/r2a/bench2r.json 2> gpurun_out/r2a/bench2r.err; echo "bench2 rc=$?"
tail -c 1500 gpurun_out/r2a/bench2r.json; tail -5 gpurun_out/r2a/bench2r.err
NFX_BENCH_FORCE_DIST=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29555 bench.py --nuclei 100000 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r2a/bench_dist1.json 2> gpurun_out/r2a/bench_dist1.err; echo "dist1 rc=$?"
tail -c 600 gpurun_out/r2a/bench_dist1.json
timeout -k 10 300 python scripts/err_profile.py 3000 > gpurun_out/r2a/err_profile.txt 2>&1; echo "err rc=$?"
