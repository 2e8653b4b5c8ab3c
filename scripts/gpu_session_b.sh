set -x
mkdir -p gpurun_out/r2b
python -m pytest tests -m gpu -x -q > gpurun_out/r2b/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2b/pytest.log
tail -5 gpurun_out/r2b/pytest.log
python scripts/lib_exp.py --workload config2 --steps 20 --warmup 3 > gpurun_out/r2b/lpr_c2.txt 2>&1; cat gpurun_out/r2b/lpr_c2.txt
python scripts/lib_exp.py --workload config3 --steps 3 --warmup 1 --no-also > gpurun_out/r2b/lpr_c3.txt 2>&1; cat gpurun_out/r2b/lpr_c3.txt
timeout -k 10 300 python bench.py --gpus 2 --nuclei 20000 --strong-nuclei 20000 --full-set-nuclei 5000 --steps 2 --warmup 1 > gpurun_out/r2b/bench2r.json 2> gpurun_out/r2b/bench2r.err; echo "bench2 rc=$?"
tail -c 2500 gpurun_out/r2b/bench2r.json; tail -3 gpurun_out/r2b/bench2r.err
timeout -k 10 300 python scripts/err_profile.py 3000 > gpurun_out/r2b/err_profile.txt 2>&1; echo "err rc=$?"
grep -E "==|gabor (mean|variance)|set " gpurun_out/r2b/err_profile.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
