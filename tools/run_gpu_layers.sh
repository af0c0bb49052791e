cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/layers
timeout 600 python tools/layer_times.py regular 64 > gpurun_out/layers/layer_times.txt 2> gpurun_out/layers/err.log; echo rc=$?; head -5 gpurun_out/layers/layer_times.txt
bash tools/run_gpu_lds.sh
