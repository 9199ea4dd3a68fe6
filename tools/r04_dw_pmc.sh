set -x
mkdir -p gpurun_out/r4x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU" "SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" "SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout 200 rocprofv3 --kernel-trace --pmc $set -d gpurun_out/r4x/dwpmc_$tag -o p -- python tools/dwln_bench.py > gpurun_out/r4x/dwpmc_$tag.log 2>&1
  python tools/pmc_kernel.py gpurun_out/r4x/dwpmc_$tag "lnb_kernel<768" >> gpurun_out/r4x/dw_pmc.txt 2>&1
  rm -rf gpurun_out/r4x/dwpmc_$tag
done
