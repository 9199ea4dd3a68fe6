set -x
mkdir -p gpurun_out/r4x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
[ -n "$SKIP_TESTS" ] || (timeout 900 python -m pytest tests -m gpu -q --tb=line 2>&1 | grep -v "^$" | tail -12) > gpurun_out/r4x/t_all.log
timeout 400 rocprofv3 --kernel-trace --stats -d gpurun_out/r4x/prof -o large -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras --no-single-frame > gpurun_out/r4x/prof_bench.log 2>&1
python tools/rocprof_summary.py $(ls gpurun_out/r4x/prof/*results.db gpurun_out/r4x/prof/*/*results.db 2>/dev/null | head -1) > gpurun_out/r4x/kernel_stats.csv 2>gpurun_out/r4x/summary.err
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/r4x/prof1 -o b1 -- python bench.py --batch 1 --steps 40 --warmup 5 --no-cpu-baseline --no-extras --no-single-frame > gpurun_out/r4x/prof_bench_b1.log 2>&1
python tools/rocprof_summary.py $(ls gpurun_out/r4x/prof1/*results.db gpurun_out/r4x/prof1/*/*results.db 2>/dev/null | head -1) > gpurun_out/r4x/kernel_stats_b1.csv 2>>gpurun_out/r4x/summary.err
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/r4x/pmc_fetch -o p -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-single-frame > gpurun_out/r4x/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/r4x/pmc_write -o p -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-single-frame > gpurun_out/r4x/pmc_write.log 2>&1
python tools/pmc_traffic.py gpurun_out/r4x/pmc_fetch gpurun_out/r4x/pmc_write > gpurun_out/r4x/pmc_traffic.json 2>gpurun_out/r4x/pmc.err
UNI_PROF_DUMP=gpurun_out/r4x/pd_b16.txt timeout 200 python bench.py --no-cpu-baseline --no-extras --no-single-frame > gpurun_out/r4x/bench_b16.json 2>/dev/null
UNI_PROF_DUMP=gpurun_out/r4x/pd_b1.txt timeout 200 python bench.py --batch 1 --steps 40 --no-cpu-baseline --no-extras --no-single-frame > gpurun_out/r4x/bench_b1.json 2>/dev/null
python tools/prof_shapes.py gpurun_out/r4x/pd_b16.txt 3 > gpurun_out/r4x/shapes_b16.txt
python tools/prof_shapes.py gpurun_out/r4x/pd_b1.txt 3 > gpurun_out/r4x/shapes_b1.txt
rm -rf gpurun_out/r4x/prof gpurun_out/r4x/prof1 gpurun_out/r4x/pmc_fetch gpurun_out/r4x/pmc_write gpurun_out/r4x/pd_b16.txt gpurun_out/r4x/pd_b1.txt
cat gpurun_out/r4x/t_all.log; head -12 gpurun_out/r4x/kernel_stats.csv; cat gpurun_out/r4x/pmc_traffic.json | head -30
