import os, sys, subprocess, json
# phase-skipping timing experiment for k_glcm (config3, 300k nuclei)
for flag in ["0","1","2","3","4","7","8"]:
    env=dict(os.environ, NFX_GLCM_DBG=flag)
    out=subprocess.run([sys.executable,"bench.py","--workload","config3","--nuclei","300000","--steps","2","--warmup","1","--no-cpu-baseline","--sets","glcm"],env=env,capture_output=True,text=True)
    try:
        d=json.loads(out.stdout.strip().splitlines()[-1]); print(flag, d["roofline"]["kernels_ms_per_step"])
    except Exception as e:
        print(flag,"ERR",out.stderr[-500:])
