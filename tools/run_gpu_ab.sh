cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/tb
run() {
  env "$@" timeout 600 python bench.py --no-cpu-baseline --train-only --prof-steps 0 > gpurun_out/tb/b.json 2> gpurun_out/tb/bench.err
  python -c "
import json
d=json.load(open('gpurun_out/tb/b.json'))
print('$*', d['ms_per_step'])
"
}
run X=0
run ADVOC_H3_DEEP_WGS_PER_CU=3
run ADVOC_H3_DEEP_WGS_PER_CU=1
run ADVOC_H3_REM_WGS_PER_CU=1
run ADVOC_H3_REM_WGS_PER_CU=3 ADVOC_H3_REM_SPLIT_DIV=4
run ADVOC_H3_PATCH_REM=0
run ADVOC_H3_MIN_TILES=1
run X=0
