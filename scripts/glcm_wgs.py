import os, sys, subprocess, json
# GLCM class-10 kernel: throughput vs workgroups (waves) per CU; ps 32 keeps every nucleus in the first class
for w in ["4", "6", "8", "10"]:
    env = dict(os.environ, NFX_GLCM_WGS=w)
    out = subprocess.run([sys.executable, "bench.py", "--workload", "config2", "--nuclei", "300000", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--sets", "glcm"], env=env, capture_output=True, text=True)
    try:
        d = json.loads(out.stdout.strip().splitlines()[-1]); print(w, d["roofline"]["kernels_ms_per_step"])
    except Exception:
        print(w, "ERR", out.stderr[-400:])
