cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
ROWS=30 bash tools/micro/env_ab2.sh "" "ADVOC_H3_DEEP_STAGES=3" "ADVOC_H3_DEEP_STAGES=4" "ADVOC_H3_REM_WGS_PER_CU=3" "ADVOC_H3_REM_WGS_PER_CU=1" "ADVOC_H3_REM_SPLIT_DIV=4" "ADVOC_H3_DEEP_WGS_PER_CU=3" "ADVOC_H3_DEEP_SPLIT_DIV=4" "" > gpurun_out/r6p_deep_env.txt 2>&1
grep "ms_per_step\|gather_gemm" gpurun_out/r6p_deep_env.txt
