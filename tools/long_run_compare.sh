#!/bin/bash
# 2 000 training steps of `main.py --data synthetic` (b = 32, 64x64x32 crops, 10 epochs x 200 steps) in bf16 (--amp) and in float32, same seed:
# the per-epoch meters of both runs side by side ("a 2 000-step --amp vs fp32 loss log under profiles/").
#   gpurun -- 'tools/long_run_compare.sh r05'   ->  gpurun_out/<tag>_long_run_{bf16,fp32}.log, gpurun_out/<tag>_long_run_compare.txt
TAG=${1:-long}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out /tmp/lr_out
COMMON="--data synthetic --b 32 --epochs 9 --steps_per_epoch 200 --seed 42 --gpus 0 --workers 0 --output /tmp/lr_out"
python main.py $COMMON --amp > gpurun_out/${TAG}_long_run_bf16.log 2>&1
python main.py $COMMON > gpurun_out/${TAG}_long_run_fp32.log 2>&1
# the yardstick: a second float32 run whose ONLY difference is the summation order of the 26 cosine means (one launch per mean instead of one for all):
# how far two float32 trajectories drift apart from a last-bit perturbation
PCRL_FUSED_COS=0 python main.py $COMMON > gpurun_out/${TAG}_long_run_fp32b.log 2>&1
python - "$TAG" <<'PY' > gpurun_out/${TAG}_long_run_compare.txt
import re, sys
tag = sys.argv[1]
def rows(path):
    out = {}
    for ln in open(path):
        m = re.match(r"Train: \[(\d+)\]\[(\d+)/(\d+)\].*?cos_loss ([-\d.]+) \(([-\d.]+)\).*?mg loss ([-\d.]+) \(([-\d.]+)\).*?local loss ([-\d.]+) \(([-\d.]+)\)", ln)
        if m:
            e, i, n = int(m.group(1)), int(m.group(2)), int(m.group(3))
            out[(e, i)] = tuple(float(m.group(k)) for k in (5, 7, 9))     # running averages of the epoch: global cosine, restoration (mg), local cosine
    return out
a, b, c = rows(f"gpurun_out/{tag}_long_run_bf16.log"), rows(f"gpurun_out/{tag}_long_run_fp32.log"), rows(f"gpurun_out/{tag}_long_run_fp32b.log")
print("# main.py --data synthetic --b 32 --epochs 9 --steps_per_epoch 200 --seed 42: bf16 (--amp) vs float32, per-epoch running averages of the log line")
print("# (cos_loss = global cosine term, mg = restoration MSE, local = local cosine term); one row per 50 steps")
print("# fp32' = the same float32 run with the cosine means summed by 26 launches instead of one (PCRL_FUSED_COS=0): a last-bit perturbation -- the yardstick")
print("%6s %5s | %9s %9s %9s | %9s %9s %9s | %9s %9s %9s | fp32' - fp32: %9s %9s %9s" % ("epoch", "step", "cos bf16", "cos fp32", "diff", "mg bf16", "mg fp32", "diff", "loc bf16", "loc fp32", "diff", "cos", "mg", "local"))
worst = [0.0, 0.0, 0.0]
worst2 = [0.0, 0.0, 0.0]
for k in sorted(a):
    if k in b and k[1] % 50 == 0:
        x, y = a[k], b[k]
        z = c.get(k, y)
        print("%6d %5d | %+9.4f %+9.4f %+9.1e | %9.5f %9.5f %+9.1e | %+9.4f %+9.4f %+9.1e |               %+9.1e %+9.1e %+9.1e" % (k[0], k[1], x[0], y[0], x[0] - y[0], x[1], y[1], x[1] - y[1], x[2], y[2], x[2] - y[2], z[0] - y[0], z[1] - y[1], z[2] - y[2]))
        if k[1] == max(i for (e, i) in a if e == k[0]):
            for j in range(3):
                worst[j] = max(worst[j], abs(x[j] - y[j]))
                worst2[j] = max(worst2[j], abs(z[j] - y[j]))
print("# largest |bf16 - fp32| of the END-OF-EPOCH averages over the 10 epochs: cos %.2e  mg %.2e  local %.2e" % tuple(worst))
print("# largest |fp32' - fp32| (two float32 runs, last-bit perturbation)            : cos %.2e  mg %.2e  local %.2e" % tuple(worst2))
PY
tail -5 gpurun_out/${TAG}_long_run_compare.txt
