import importlib, os, sys, time, subprocess, json
# runs bench config5 gabor-only with different debug flags
for flag in ["0","1","2","3","4"]:
    env=dict(os.environ, NFX_GABOR_DBG=flag)
    out=subprocess.run([sys.executable,"bench.py","--workload","config5","--nuclei","300000","--steps","2","--warmup","1","--no-cpu-baseline","--sets","gabor"],env=env,capture_output=True,text=True)
    try:
        d=json.loads(out.stdout.strip().splitlines()[-1]); print(flag, d["roofline"]["kernels_ms_per_step"])
    except Exception as e:
        print(flag,"ERR",out.stderr[-500:])
