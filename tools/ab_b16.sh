run() { env "$@" timeout 200 python bench.py --no-extras --no-cpu-baseline --no-single-frame --steps 12 2>/dev/null | python -c "
import json,sys;b=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('%-28s %.2f fps  gemm %.1f us/launch' % ('$*', b['value'], b['roofline']['avg_launch_us']))"; }
run X=base
run UNI_NO_H2D_192=1
run UNI_MLP_LAYOUT=0
run X=base
run UNI_DW_W12=0
run UNI_DW_PACK=0
run UNI_NO_FORK=1
run X=base
run UNI_NO_FORK_MB=1
run UNI_GN_PER=2
run UNI_GN_PER=8
run X=base
run UNI_BENCH_CORR_PER_FRAME=1
run UNI_NO_MLP_FUSED=1
run X=base
