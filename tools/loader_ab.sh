cd $GRAFT_REPO_ROOT
python - <<PY
import numpy as np, os
root = "/tmp/luna_r4"; rng = np.random.default_rng(0)
for fold in range(10):
    d = f"{root}/subset{fold}"; os.makedirs(d, exist_ok=True)
    for s in range(137 if fold < 7 else 8):
        np.save(f"{d}/s{fold}x{s}_global_0.npy", rng.random((2, 64, 64, 32), dtype=np.float32))
        np.save(f"{d}/s{fold}x{s}_local_0.npy", rng.random((6, 16, 16, 16), dtype=np.float32))
PY
for mode in slots none; do
  echo "== PCRL_LOADER_PIN=$mode"
  PCRL_LOADER_TIMING=1 PCRL_LOADER_PIN=$mode python main.py --data /tmp/luna_r4 --n luna --d 3 --b 32 --epochs 2 --gpus 0 --amp --workers 8 --ratio 1.0 --output /tmp/ck_r4 2>&1 | grep -E "^\[loader\]|total time|\[30/30\]" | cut -c1-200
done
