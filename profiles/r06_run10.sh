cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6h
timeout 600 python -m pytest tests/test_gpu_exceptions.py -q -m gpu -x > gpurun_out/r6h/test_exc.log 2>&1
tail -12 gpurun_out/r6h/test_exc.log
timeout 300 python bench.py --gpus 8 --backend gloo --share-device --e2e-hash --bases 1000000000 > gpurun_out/r6h/bench_n8_share_1g.json 2> gpurun_out/r6h/bench_n8_1g.err
echo "rc n8 1g: $?"; grep "^{" gpurun_out/r6h/bench_n8_share_1g.json | cut -c1-600
nvidia-smi 2>/dev/null; rocm-smi --showmeminfo vram 2>/dev/null | head -8
