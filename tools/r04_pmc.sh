set -x
mkdir -p gpurun_out/r4x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/r4x/pmc_fetch -o p -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-single-frame > gpurun_out/r4x/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/r4x/pmc_write -o p -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-single-frame > gpurun_out/r4x/pmc_write.log 2>&1
python tools/pmc_traffic.py gpurun_out/r4x/pmc_fetch gpurun_out/r4x/pmc_write > gpurun_out/r4x/pmc_traffic.json 2>gpurun_out/r4x/pmc.err
rm -rf gpurun_out/r4x/pmc_fetch gpurun_out/r4x/pmc_write
