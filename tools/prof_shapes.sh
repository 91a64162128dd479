set -e
mkdir -p gpurun_out/shapes
for sh in "500000 128 0 2" "500000 128 0 4" "500000 128 0 6" "500000 128 0 8" "500000 128 0 10" "500000 128 0 12" "500000 128 12 6"; do
  tag=$(echo $sh | tr ' ' '_')
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/shapes/$tag -- python3 tools/one_shape.py $sh 20 > gpurun_out/shapes/$tag.log 2>&1
  f=$(find gpurun_out/shapes/$tag -name '*kernel_stats.csv' | head -1)
  echo "== $sh"; head -2 gpurun_out/shapes/$tag.log | tail -1 || true
  python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:8]:
    print(f"  {r['Name'][:70]:70s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:9.1f} pct={r['Percentage']}")
PY
done
