set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/final/gpu_tests.txt 2>&1; tail -2 gpurun_out/final/gpu_tests.txt
python bench.py > gpurun_out/final/bench_line.json 2> gpurun_out/final/bench.err
python bench.py --workload config2_1e5x64_6+2 --no-cpu-baseline > gpurun_out/final/config2_bench_line.json 2> gpurun_out/final/config2.err
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/final/stats -o r02 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --restarts 8 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/final/stats.log 2>&1)
echo stats done
bash tools/pmc_passes.sh final/pmc
timeout -k 10 100 tools/rowpass2_probe 20 2 > gpurun_out/final/rowpass2_probe.txt 2>&1
timeout -k 10 100 tools/gram_i8_probe > gpurun_out/final/gram_i8_probe.txt 2>&1
timeout -k 10 100 tools/gram_i8_probe_ns >> gpurun_out/final/gram_i8_probe.txt 2>&1
(timeout -k 10 600 python tools/bootstrap_headline_bench.py 40 20; timeout -k 10 600 python tools/bootstrap_headline_bench.py 120 20; timeout -k 10 300 python tools/replicate_overheads.py | tail -2) 2>&1 | grep -v amdgpu.ids > gpurun_out/final/bootstrap_headline.txt
timeout -k 10 200 python tools/restart_overheads.py 2>&1 | grep -v amdgpu.ids > gpurun_out/final/restart_overheads.txt
du -sh gpurun_out/final
DMF_BENCH_DEPTH=120 python bench.py --no-cpu-baseline --restarts 8 > gpurun_out/final/deep_coverage_bench_line.json 2> gpurun_out/final/deep.err
timeout -k 10 400 python tools/ic_sweep_bench.py 50 2>&1 | grep -v amdgpu.ids > gpurun_out/final/ic_sweep.txt
timeout -k 10 300 python tools/wide_nu_sweep.py 2>&1 | grep -v amdgpu.ids > gpurun_out/final/wide_row_groups.txt
bash tools/pmc_shape.sh final/pmc_0_8 500000 128 0 8 6 > /dev/null && python3 tools/pmc_summary.py gpurun_out/final/pmc_0_8 > gpurun_out/final/wide_pmc_0_8.txt
bash tools/pmc_shape.sh final/pmc_0_12 500000 128 0 12 6 > /dev/null && python3 tools/pmc_summary.py gpurun_out/final/pmc_0_12 > gpurun_out/final/wide_pmc_0_12.txt
(timeout -k 10 200 python tools/odd_s_bench.py; timeout -k 10 200 python tools/big_s_bench.py; timeout -k 10 200 python tools/big_nc_bench.py; timeout -k 10 200 python tools/big_nu_bench.py) 2>&1 | grep -v amdgpu.ids > gpurun_out/final/shape_coverage.txt
