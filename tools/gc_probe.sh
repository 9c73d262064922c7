python - <<'PY'
import gc, json, subprocess, sys, os
for mode in ("default", "nogc"):
    env = dict(os.environ, PCRL_BENCH_NOGC="1" if mode == "nogc" else "0")
    out = subprocess.run([sys.executable, "bench.py", "--steps", "10", "--warmup", "3", "--no-cpu-baseline"], capture_output=True, text=True, env=env).stdout
    d = json.loads(out.strip().splitlines()[-1]); print(mode, d["value"], d["ms_per_step"], d["diag"]["gpu_ms_per_step"])
PY
