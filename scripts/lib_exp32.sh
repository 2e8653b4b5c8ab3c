python scripts/lib_exp.py --workload config2 --nuclei 300000 --steps 2 --warmup 1 --sets glcm
python scripts/lib_exp.py --workload config3 --nuclei 300000 --steps 2 --warmup 1 --sets glcm
