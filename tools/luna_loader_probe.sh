python - <<'PY'
import numpy as np, os
root="/tmp/luna"
for fold in range(10):
    d=f"{root}/subset{fold}"; os.makedirs(d, exist_ok=True)
    for s in range(4):
        for k in range(2):
            np.save(f"{d}/s{fold}x{s}_global_{k}.npy", np.random.rand(2,64,64,32).astype(np.float32))
            np.save(f"{d}/s{fold}x{s}_local_{k}.npy", np.random.rand(6,16,16,16).astype(np.float32))
PY
python main.py --data /tmp/luna --n luna --d 3 --b 16 --epochs 1 --gpus 0 --amp --workers 4 --ratio 1.0 --output /tmp/ck4 2>&1 | grep "total train\|Train:\|epoch\|Error\|error" | tail -6
python - <<'PY'
import sys, time, types, torch
sys.path.insert(0, ".")
from pcrlv2_amd.data import luna_pretask_loaders
a = types.SimpleNamespace(data="/tmp/luna", ratio=1.0, b=32, workers=8, seed=0)
ld = luna_pretask_loaders(a)["train"]
it = iter(ld); next(it); torch.cuda.synchronize(); t = time.time(); n = 0
for b in it: n += b[0].shape[0]
torch.cuda.synchronize(); dt = time.time() - t
print(f"loader: {n} crops in {dt:.2f} s = {n/dt:.0f} crops/s (np.load in 8 workers + GPU augmentation, b=32)")
PY
