cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/tb
ADVOC_H3_REM_WS=1 ADVOC_H3_REM_WGS_PER_CU=2 ADVOC_H3_REM_SPLIT_DIV=4 timeout 900 python -m pytest tests/test_hip_conv.py tests/test_hip_fullsize.py -m gpu -q -x -k "patch or fullsize or rem" 2>&1 | tail -3
run() {
  env "$@" timeout 600 python bench.py --no-cpu-baseline --train-only --prof-steps 0 > gpurun_out/tb/b.json 2> gpurun_out/tb/bench.err
  python -c "
import json
d=json.load(open('gpurun_out/tb/b.json'))
print('$*', d['ms_per_step'])
"
}
run X=0
run ADVOC_H3_REM_WS=1 ADVOC_H3_REM_WGS_PER_CU=1 ADVOC_H3_REM_SPLIT_DIV=8
run ADVOC_H3_REM_WS=1 ADVOC_H3_REM_WGS_PER_CU=2 ADVOC_H3_REM_SPLIT_DIV=8
run ADVOC_H3_REM_WS=1 ADVOC_H3_REM_WGS_PER_CU=3 ADVOC_H3_REM_SPLIT_DIV=4
run ADVOC_H3_REM_WS=1 ADVOC_H3_REM_WGS_PER_CU=4 ADVOC_H3_REM_SPLIT_DIV=4
run X=0
