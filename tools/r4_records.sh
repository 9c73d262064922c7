#!/bin/bash
# Round-4 side records kept under profiles/ (run on the GPU box through gpurun): tools/r4_records.sh <tag>
#   (1) the LUNA loader end to end: `python main.py --data <tree of .npy crops in the LUNA pre-task layout> --b 32 --amp --workers 8`
#       -- crops/s per epoch and the BT / DT meters (train_3d.py:102-103) next to the resident-batch run (`--data synthetic`, same steps per epoch);
#   (2) `python main.py --data synthetic` over epochs 0..12 with the per-epoch empty_cache ON (default: pools kept, ops.empty_cache),
#       with the raw torch.cuda.empty_cache() and without the call;
#   (3) two ranks on ONE GPU over gloo through bench.py's multi-rank path: the distributed.ab block (VERDICT r3 #2b).
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
NCROPS=${NCROPS:-960}
python - <<PY
import numpy as np, os
root = "/tmp/luna_r4"
rng = np.random.default_rng(0)
per_fold = $NCROPS // 7
for fold in range(10):
    d = f"{root}/subset{fold}"; os.makedirs(d, exist_ok=True)
    n = per_fold if fold < 7 else 8
    for s in range(n):
        np.save(f"{d}/s{fold}x{s}_global_0.npy", rng.random((2, 64, 64, 32), dtype=np.float32))
        np.save(f"{d}/s{fold}x{s}_local_0.npy", rng.random((6, 16, 16, 16), dtype=np.float32))
print("generated", per_fold * 7, "training crops under", root)
PY
STEPS=$(( (NCROPS / 7 * 7 + 31) / 32 ))
python main.py --data /tmp/luna_r4 --n luna --d 3 --b 32 --epochs 3 --gpus 0 --amp --workers 8 --ratio 1.0 --output /tmp/ck_r4 > $O/main_loader.log 2>&1
python main.py --data synthetic --d 3 --b 32 --epochs 3 --steps_per_epoch $STEPS --gpus 0 --amp --output /tmp/ck_r4s > $O/main_resident.log 2>&1
PCRL_AUG_STREAM=0 python main.py --data /tmp/luna_r4 --n luna --d 3 --b 32 --epochs 3 --gpus 0 --amp --workers 8 --ratio 1.0 --output /tmp/ck_r4 > $O/main_loader_noaugstream.log 2>&1
python - <<PY > $O/loader_end_to_end.txt
import re
def rows(path, crops_per_epoch):
    txt = open(path).read()
    out = []
    for e, t in re.findall(r"epoch (\d+), total time ([0-9.]+)", txt):
        lines = re.findall(r"Train: \[%s\]\[(\d+)/(\d+)\]\s+BT ([0-9.]+) \(([0-9.]+)\)\s+DT ([0-9.]+) \(([0-9.]+)\)" % e, txt)
        bt, dt = (lines[-1][3], lines[-1][5]) if lines else ("?", "?")
        out.append("e%s %.0f crops/s (BT avg %s s, DT avg %s s)" % (e, crops_per_epoch / float(t), bt, dt))
    return out
n = $NCROPS // 7 * 7
print("# python main.py --data <dir> --n luna --d 3 --b 32 --amp --workers 8 on a generated tree of %d training crops in the LUNA pre-task layout" % n)
print("# (np.load in 8 DataLoader workers -> pinned memory -> GPU augmentation kernels one batch ahead on their own stream), %d steps per epoch, bf16;" % $STEPS)
print("# epoch 0 includes start-up (library load, allocator provisioning, worker start)")
print("loader (default: augmentation of batch k+1 on its own stream under step k):")
print("  " + "\n  ".join(rows("$O/main_loader.log", n)))
print("loader, PCRL_AUG_STREAM=0 (augmentation on the training stream when the batch is asked for):")
print("  " + "\n  ".join(rows("$O/main_loader_noaugstream.log", n)))
print("resident batch (--data synthetic, same steps per epoch):")
print("  " + "\n  ".join(rows("$O/main_resident.log", $STEPS * 32)))
PY
cat $O/loader_end_to_end.txt
for mode in 1 raw 0; do
  PCRL_PROVISION_VERBOSE=1 PCRL_EMPTY_CACHE_PER_EPOCH=$mode python main.py --data synthetic --d 3 --b 32 --epochs 12 --steps_per_epoch 30 --gpus 0 --amp --output /tmp/out_ec$mode > $O/main_ec_$mode.log 2>&1
done
python - <<PY > $O/main_synthetic_epochs.txt
import re
names = {"1": "PCRL_EMPTY_CACHE_PER_EPOCH=1 (default: the reference's per-epoch empty_cache, steady-state pools kept -- ops.empty_cache)",
         "raw": "PCRL_EMPTY_CACHE_PER_EPOCH=raw (torch.cuda.empty_cache() itself: pools released and re-provisioned every epoch)",
         "0": "PCRL_EMPTY_CACHE_PER_EPOCH=0 (no call: round 3's default)"}
print("# python main.py --data synthetic --d 3 --b 32 --epochs 12 --steps_per_epoch 30 --amp: crops/s per epoch (epoch 0 includes start-up; divergence guard live from epoch 11)")
for mode in ("1", "raw", "0"):
    txt = open("$O/main_ec_%s.log" % mode).read()
    rows = re.findall(r"epoch (\d+), total time ([0-9.]+)", txt)
    prov = re.findall(r"\[provision\][^\n]*", txt)
    print(names[mode] + ":")
    print("  " + "  ".join("e%s %.0f" % (e, 32 * 30 / float(t)) for e, t in rows))
    print("  provisioning events: %d%s" % (len(prov), ("; slowest: " + max(prov, key=lambda l: float(re.search(r"in ([0-9.]+) s", l).group(1)))) if prov else ""))
PY
cat $O/main_synthetic_epochs.txt
PCRL_DIST_BACKEND=gloo python bench.py --gpus 2 --steps 6 --warmup 2 --no-cpu-baseline --no-alone --no-secondary > $O/bench_gloo2.json 2> $O/bench_gloo2.err
python -c "
import json; d=json.loads(open('$O/bench_gloo2.json').read().strip().splitlines()[-1]); print(json.dumps(d.get('distributed', {}).get('ab'), indent=1)); print(d['value'], d['ms_per_step'])"
