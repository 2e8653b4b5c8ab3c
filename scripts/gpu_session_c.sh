set -x
mkdir -p gpurun_out/r2c
python -m pytest tests -m gpu -x -q > gpurun_out/r2c/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2c/pytest.log
tail -5 gpurun_out/r2c/pytest.log
python bench.py --workload config2 --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r2c/c2.json 2>gpurun_out/r2c/c2.err; tail -c 700 gpurun_out/r2c/c2.json
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-also > gpurun_out/r2c/c3.json 2>gpurun_out/r2c/c3.err; tail -c 500 gpurun_out/r2c/c3.json
