# Round 3: everything profiles/ quotes, in one go on the GPU box (writes under gpurun_out/final6/).
#   bash tools/final_pass_r03.sh          (the -m gpu suite is run separately: python -m pytest tests -q -m gpu)
set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/final6
mkdir -p $O
python bench.py > $O/bench_line.json 2> $O/bench.err
python bench.py --workload config2_1e5x64_6+2 --no-cpu-baseline > $O/config2_bench_line.json 2> $O/config2.err
DMF_BENCH_DEPTH=120 python bench.py --no-cpu-baseline --restarts 8 > $O/deep_coverage_bench_line.json 2> $O/deep.err
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats -o r03 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --restarts 8 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/stats.log 2>&1)
echo stats done
bash tools/pmc_shape.sh final6/pmc_headline 1000000 256 12 4 6 > /dev/null
python3 tools/collect_counters.py headline_1e6x256_12+4 $O/pmc_headline $O/counters.json > /dev/null
python3 tools/pmc_summary.py $O/pmc_headline > $O/headline_pmc_summary.txt
echo pmc done
timeout -k 10 100 tools/rowpass2_probe 20 2 > $O/rowpass2_probe.txt 2>&1
(echo "round-2 kernel (commit 494a33b), unstamped, 8 launches back to back:"; timeout -k 10 100 tools/rowpass2_probe_r02 20 2 8; echo "this round's kernel, same box:"; timeout -k 10 100 tools/rowpass2_probe_ns 20 2 8; echo "round 2 again:"; timeout -k 10 100 tools/rowpass2_probe_r02 20 2 8; echo "this round again:"; timeout -k 10 100 tools/rowpass2_probe_ns 20 2 8) > $O/rowpass2_ab.txt 2>&1
tools/pmc_probe.sh final6/pmc_probe_r02 tools/rowpass2_probe_r02 20 2 3 > $O/rowpass2_lds_counters.txt 2>&1
tools/pmc_probe.sh final6/pmc_probe_now tools/rowpass2_probe_ns 20 2 3 >> $O/rowpass2_lds_counters.txt 2>&1
timeout -k 10 200 python tools/restart_overheads.py 2>&1 | grep -v amdgpu.ids > $O/restart_overheads.txt
timeout -k 10 400 python tools/ic_sweep_bench.py 50 2>&1 | grep -v amdgpu.ids > $O/ic_sweep.txt
timeout -k 10 300 python tools/wide_nu_sweep.py 2>&1 | grep -v amdgpu.ids > $O/wide_row_groups.txt
(timeout -k 10 600 python tools/bootstrap_headline_bench.py 40 20; timeout -k 10 300 python tools/replicate_overheads.py | tail -2) 2>&1 | grep -v amdgpu.ids > $O/bootstrap_headline.txt
du -sh $O
