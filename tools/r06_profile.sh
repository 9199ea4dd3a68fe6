# Round-6 evidence run on the GPU box (one gpurun call; everything lands in gpurun_out/r6x, the summaries are copied to profiles/r06_*):
#   the full -m gpu suite, kernel traces at 16 frames and at one frame per step, the two PMC passes (HBM-side traffic per kernel class), per-shape
#   event dumps, the full bench line incl. the PyTorch-ROCm port leg (`--torch-rocm-port`), the two-rank `--task mix` plumbing run on ONE GPU.
#   (The probes of this round were run on their own: profiles/r06_persist_probe.txt, r06_winograd_probe.txt, r06_concurrent_streams_*, r06_dwconv_b1_rowsplit.txt.)
set -x
mkdir -p gpurun_out/r6x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6x
[ -n "$SKIP_TESTS" ] || (timeout 1500 python -m pytest tests -m gpu -q --tb=line 2>&1 | grep -v "^$" | tail -12) > $O/t_all.log
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof -o large -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras --no-single-frame > $O/prof_bench.log 2>&1
python tools/rocprof_summary.py $(ls $O/prof/*results.db $O/prof/*/*results.db 2>/dev/null | head -1) > $O/kernel_stats.csv 2>$O/summary.err
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof1 -o b1 -- python bench.py --batch 1 --steps 40 --warmup 5 --no-cpu-baseline --no-extras --no-single-frame > $O/prof_bench_b1.log 2>&1
python tools/rocprof_summary.py $(ls $O/prof1/*results.db $O/prof1/*/*results.db 2>/dev/null | head -1) > $O/kernel_stats_b1.csv 2>>$O/summary.err
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o p -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-single-frame > $O/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o p -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-single-frame > $O/pmc_write.log 2>&1
python tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write > $O/pmc_traffic.json 2>$O/pmc.err
UNI_PROF_DUMP=$O/pd_b16.txt timeout 200 python bench.py --no-cpu-baseline --no-extras --no-single-frame > $O/bench_b16.json 2>/dev/null
UNI_PROF_DUMP=$O/pd_b1.txt timeout 200 python bench.py --batch 1 --steps 40 --no-cpu-baseline --no-extras --no-single-frame > $O/bench_b1.json 2>/dev/null
python tools/prof_shapes.py $O/pd_b16.txt 3 > $O/shapes_b16.txt
python tools/prof_shapes.py $O/pd_b1.txt 3 > $O/shapes_b1.txt
timeout 900 python bench.py --torch-rocm-port > $O/bench_full.json 2> $O/bench_full.err
UNI_BENCH_SHARE_GPU=1 timeout 600 python bench.py --gpus 2 --task mix --steps 4 --warmup 1 --batch 4 --no-cpu-baseline > $O/bench_mix2_shared_gpu.json 2> $O/bench_mix2.err
rm -rf $O/prof $O/prof1 $O/pmc_fetch $O/pmc_write $O/pd_b16.txt $O/pd_b1.txt
cat $O/t_all.log; head -12 $O/kernel_stats.csv; head -30 $O/pmc_traffic.json; tail -c 1200 $O/bench_mix2_shared_gpu.json; tail -c 300 $O/bench_mix2.err; cut -c1-400 $O/bench_full.json
