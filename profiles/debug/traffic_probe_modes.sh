export TMPDIR=/tmp
for m in same events sync alt shard fresh; do
  for c in WRITE_SIZE FETCH_SIZE; do
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/tpm/$m_$c$m -- python3 profiles/debug/traffic_probe_modes.py $m > /dev/null 2>&1
    python3 - "$m" "$c" <<'PY'
import csv,glob,sys
m,c=sys.argv[1:3]
v=[]
for f in glob.glob(f"gpurun_out/tpm/{c}{m}/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "render_frame_kernel<0, 0, 8, false, true>" in r["Kernel_Name"] and r["Counter_Name"]==c: v.append(float(r["Counter_Value"])*1024)
print(m, c, [round(x/1e6,2) for x in v])
PY
  done
done
