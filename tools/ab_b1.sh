for e in "X=1" "UNI_DW_PX=4" "UNI_DW_PX=16" "UNI_NO_FORK=1" "UNI_NO_SPLITK=1"; do
  r=$(env $e python bench.py --no-extras --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | python -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); s=d['single_frame']; print(s['ms'], s['ms_min'], s['hbm']['dwconv7_ln']['ms_per_frame'], s['roofline']['gemm_ms_per_frame'], d['value'])")
  echo "$e -> single-frame ms mean/min, dwconv ms, gemm ms, B16 fps: $r"
done
