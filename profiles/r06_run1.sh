cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6a
python profiles/input_classes.py human human_iupac human_hash > gpurun_out/r6a/iupac.txt 2>&1
SUFR_HIP_DEBUG=1 python profiles/input_classes.py human_dna > gpurun_out/r6a/dna_debug.txt 2>&1
tail -3 gpurun_out/r6a/iupac.txt
