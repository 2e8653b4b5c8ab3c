import os, sys, subprocess, json
# compares alternative builds of libnfx.so (scratch_exp/*.so) on one workload
libs = [None] + sorted(p for p in os.listdir("scratch_exp") if p.endswith(".so"))
args = sys.argv[1:] or ["--workload", "config3", "--inputs", "resident", "--nuclei", "300000", "--steps", "2", "--warmup", "1", "--sets", "glcm", "--no-also"]
for lib in libs:
    env = dict(os.environ)
    if lib: env["NFX_LIB_PATH"] = os.path.abspath(os.path.join("scratch_exp", lib))
    out = subprocess.run([sys.executable, "bench.py", "--no-cpu-baseline", *args], env=env, capture_output=True, text=True)
    try:
        d = json.loads(out.stdout.strip().splitlines()[-1]); print(lib or "default", d["ms_per_step"], d["roofline"]["kernels_ms_per_step"])
    except Exception:
        print(lib, "ERR", out.stderr[-400:])
