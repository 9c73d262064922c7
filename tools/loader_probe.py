#!/usr/bin/env python3
"""Where does the LUNA loader's time go?  (main.py --data <dir> ran at 841 crops/s with DT 18 ms per step against 1 021 resident.)
Host-only measurements on a generated tree: np.load per sample in-process, then batches per second of the raw DataLoader (no GPU work at all)
for several worker counts / collate forms / pin_memory."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from pcrlv2_amd import data as D

root = sys.argv[1] if len(sys.argv) > 1 else "/tmp/luna_r4"
if not os.path.isdir(root):
    rng = np.random.default_rng(0)
    for fold in range(7):
        d = f"{root}/subset{fold}"; os.makedirs(d, exist_ok=True)
        for s in range(137):
            np.save(f"{d}/s{fold}x{s}_global_0.npy", rng.random((2, 64, 64, 32), dtype=np.float32))
            np.save(f"{d}/s{fold}x{s}_local_0.npy", rng.random((6, 16, 16, 16), dtype=np.float32))
files, _ = D.luna_file_lists(root, 1.0)
print(len(files), "crops; host cpus:", os.cpu_count(), "affinity:", len(os.sched_getaffinity(0)))
ds = D.LunaCropPairs(files)
t = time.perf_counter()
for i in range(200):
    ds[i]
print("in-process __getitem__: %.2f ms per crop" % ((time.perf_counter() - t) / 200 * 1e3))
for workers, pin, pf in ((8, True, 2), (8, False, 2), (16, True, 2), (16, True, 4), (32, True, 4)):
    ld = torch.utils.data.DataLoader(ds, batch_size=32, shuffle=True, num_workers=workers, pin_memory=pin, persistent_workers=True, prefetch_factor=pf)
    for ep in range(2):
        t = time.perf_counter(); n = 0
        for b in ld:
            n += 1
        dt = time.perf_counter() - t
    print("DataLoader workers=%2d pin_memory=%d prefetch=%d: %.1f ms per batch of 32 (%.0f crops/s), second epoch" % (workers, pin, pf, dt / n * 1e3, 32 * n / dt))
    del ld
