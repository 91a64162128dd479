set -e
mkdir -p gpurun_out/shapes
for sh in "500000 128 0 6" "500000 128 0 8"; do
  tag=$(echo $sh | tr ' ' '_')
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/shapes/$tag -- python3 tools/one_shape.py $sh 20 > gpurun_out/shapes/$tag.log 2>&1
  f=$(ls -t gpurun_out/shapes/$tag/*/*kernel_stats.csv | head -1)
  echo "== $sh"
  python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:9]:
    if 'dmf' in r['Name']: print(f"  {r['Name'][:60]:60s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:9.1f}")
PY
done
