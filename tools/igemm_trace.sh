# per-launch durations of the gather kernels inside a one-stream 3D step:  gpurun -- 'bash tools/igemm_trace.sh'
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export PCRL_WGRAD_STREAM=0 PCRL_BRANCH_STREAM=0 PCRL_VIEW_STREAMS=0
for CFG in "default::" "vmajor0:PCRL_IGEMM_VMAJOR=0:" "occ1::$R/build/var/libpcrl_occ1.so"; do
  NAME=${CFG%%:*}; REST=${CFG#*:}; E=${REST%%:*}; LIBP=${REST#*:}
  rm -rf $R/gpurun_out/igt
  env $E PCRL_LIB=$LIBP rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/igt -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-alone --no-secondary > $R/gpurun_out/igt.log 2>&1
  python - $NAME $(find $R/gpurun_out/igt -name "*kernel_trace.csv") <<'PY'
import csv, sys, collections
name, f = sys.argv[1:3]
rows = [r for r in csv.DictReader(open(f)) if "igemm_kernel" in r["Kernel_Name"] or "wgrad_upc8" in r["Kernel_Name"] or "splitk_finish" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
agg = collections.OrderedDict()
for r in rows[-len(rows) // 5:]:        # last step
    k = (r["Kernel_Name"].split("igemm_kernel")[-1][:28] if "igemm_kernel" in r["Kernel_Name"] else r["Kernel_Name"][-40:-10], r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"])
    a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
print("==", name, "total us", round(sum(v[1] for v in agg.values()), 1))
for k, (n, us) in agg.items():
    print("   %-30s grid %8s x %3s x %3s  n=%2d  %7.1f us each" % (k[0], k[1], k[2], k[3], n, us / n))
PY
done
rm -rf $R/gpurun_out/igt
