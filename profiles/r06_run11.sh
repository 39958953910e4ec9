cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6h
SUFR_BENCH_STACKS_AFTER=200 timeout 420 python bench.py --gpus 8 --backend gloo --share-device --e2e-hash > gpurun_out/r6h/bench_n8_share_human.json 2> gpurun_out/r6h/bench_n8.err
echo "rc n8: $?"; grep "^{" gpurun_out/r6h/bench_n8_share_human.json | cut -c1-300
grep "bench rank 0\|bench rank 7" gpurun_out/r6h/bench_n8.err | tail; grep -A8 "most recent call first" gpurun_out/r6h/bench_n8.err | head -30
