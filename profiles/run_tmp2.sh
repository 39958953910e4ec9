cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6t
timeout 700 python profiles/soak_round6.py 8 6066 > gpurun_out/r6t/soak_round6.txt 2>&1; tail -2 gpurun_out/r6t/soak_round6.txt
for p in none 0 1; do
  if [ $p = none ]; then python profiles/input_classes.py human_dna 2>&1 | grep human_dna; else SUFR_HELPER_PRIO=$p python profiles/input_classes.py human_dna 2>&1 | grep human_dna; fi
done | tee gpurun_out/r6t/prio.txt
timeout 900 python profiles/c4_oracle_hash.py 3100000000 iupac > gpurun_out/r6t/c4_hash_iupac.txt 2>&1; tail -3 gpurun_out/r6t/c4_hash_iupac.txt
timeout 1200 python profiles/c4_oracle_hash.py 3100000000 dna > gpurun_out/r6t/c4_hash_dna.txt 2>&1; tail -3 gpurun_out/r6t/c4_hash_dna.txt
