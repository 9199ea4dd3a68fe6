# Round-5 evidence run on the GPU box (one gpurun call; everything lands in gpurun_out/r5x, the summaries are copied to profiles/r05_*):
#   kernel traces at 16 frames and at one frame per step, the two PMC passes (HBM-side traffic per kernel class), per-shape event dumps,
#   the MOTS loop trace, the non-K-loop (fill + epilogue + drain) ablation table of the dominant GEMM shapes, the mask-path bench,
#   the full bench line, and the two-rank `--task mix` plumbing run on ONE GPU (UNI_BENCH_SHARE_GPU: no scaling number is claimed).
set -x
mkdir -p gpurun_out/r5x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5x
[ -n "$SKIP_TESTS" ] || (timeout 900 python -m pytest tests -m gpu -q --tb=line 2>&1 | grep -v "^$" | tail -12) > $O/t_all.log
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof -o large -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras --no-single-frame > $O/prof_bench.log 2>&1
python tools/rocprof_summary.py $(ls $O/prof/*results.db $O/prof/*/*results.db 2>/dev/null | head -1) > $O/kernel_stats.csv 2>$O/summary.err
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof1 -o b1 -- python bench.py --batch 1 --steps 40 --warmup 5 --no-cpu-baseline --no-extras --no-single-frame > $O/prof_bench_b1.log 2>&1
python tools/rocprof_summary.py $(ls $O/prof1/*results.db $O/prof1/*/*results.db 2>/dev/null | head -1) > $O/kernel_stats_b1.csv 2>>$O/summary.err
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_mots -o mots -- python tools/mots_profile.py 30 > $O/mots_prof.log 2>&1
python tools/rocprof_summary.py $(ls $O/prof_mots/*results.db $O/prof_mots/*/*results.db 2>/dev/null | head -1) > $O/mots_kernel_stats.csv 2>>$O/summary.err
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o p -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-single-frame > $O/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o p -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-single-frame > $O/pmc_write.log 2>&1
python tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write > $O/pmc_traffic.json 2>$O/pmc.err
UNI_PROF_DUMP=$O/pd_b16.txt timeout 200 python bench.py --no-cpu-baseline --no-extras --no-single-frame > $O/bench_b16.json 2>/dev/null
UNI_PROF_DUMP=$O/pd_b1.txt timeout 200 python bench.py --batch 1 --steps 40 --no-cpu-baseline --no-extras --no-single-frame > $O/bench_b1.json 2>/dev/null
python tools/prof_shapes.py $O/pd_b16.txt 3 > $O/shapes_b16.txt
python tools/prof_shapes.py $O/pd_b1.txt 3 > $O/shapes_b1.txt
# non-K-loop time per dominant shape: full kernel / no global stores / no epilogue (K loop only) / no epilogue + no DMA, 16 frames and 1 frame
(echo "# python tools/gemm_h2_bench.py 0 188 4188 16188 17188  (GEMM_SCALE=16: the bench step; columns: launcher heuristic | ping-pong kernel full | no global stores | no epilogue = K loop + fill / drain | no epilogue + no DMA)"; GEMM_SCALE=16 timeout 300 python tools/gemm_h2_bench.py 0 188 4188 16188 17188 2>&1 | grep -v amdgpu.ids) > $O/gemm_ablation_b16.txt
(echo "# the same at one frame per launch (GEMM_SCALE=1; cfg 0 = the launcher's choice: deep-pipeline tiles / 256 x 192)"; GEMM_SCALE=1 timeout 300 python tools/gemm_h2_bench.py 0 188 4188 16188 17188 2>&1 | grep -v amdgpu.ids) > $O/gemm_ablation_b1.txt
timeout 100 python tools/mask_bench.py 64 2>&1 | grep -v amdgpu.ids > $O/mask_bench.txt
timeout 600 python bench.py > $O/bench_full.json 2> $O/bench_full.err
UNI_BENCH_SHARE_GPU=1 timeout 600 python bench.py --gpus 2 --task mix --steps 4 --warmup 1 --batch 4 --no-cpu-baseline > $O/bench_mix2_shared_gpu.json 2> $O/bench_mix2.err
rm -rf $O/prof $O/prof1 $O/prof_mots $O/pmc_fetch $O/pmc_write $O/pd_b16.txt $O/pd_b1.txt
cat $O/t_all.log; head -12 $O/kernel_stats.csv; head -30 $O/pmc_traffic.json; cat $O/gemm_ablation_b16.txt $O/gemm_ablation_b1.txt $O/mask_bench.txt; tail -c 1500 $O/bench_mix2_shared_gpu.json; tail -c 300 $O/bench_mix2.err
