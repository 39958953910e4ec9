cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6g
timeout 1700 python -m pytest tests/test_bench_dist.py -q -m gpu -x > gpurun_out/r6g/test_bench_dist.log 2>&1
tail -5 gpurun_out/r6g/test_bench_dist.log
# C5 rehearsal at full size: eight ranks on the one GPU, host-side exchange, eight writers into one file
python bench.py --gpus 8 --backend gloo --share-device --e2e-hash > gpurun_out/r6g/bench_n8_share_human.json 2> gpurun_out/r6g/bench_n8.err
tail -c 3000 gpurun_out/r6g/bench_n8_share_human.json; tail -5 gpurun_out/r6g/bench_n8.err
python bench.py --gpus 1 --no-cpu-baseline --no-search --e2e-hash > gpurun_out/r6g/bench_n1_hash_human.json 2> gpurun_out/r6g/bench_n1.err
python - <<'PY'
import json
for f in ("gpurun_out/r6g/bench_n8_share_human.json", "gpurun_out/r6g/bench_n1_hash_human.json"):
    ls = [l for l in open(f) if l.startswith("{")]
    j = json.loads(ls[-1])
    print(f, j["n_gpus"], j["ms_per_step"], j.get("e2e_create", {}).get("sufr_sha256"), j.get("e2e_create", {}).get("seconds"), j.get("host_abi", {}).get("seconds"))
PY
