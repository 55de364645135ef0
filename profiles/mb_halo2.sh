# thick stride-1 kernels; OTHER=<lib.so> adds an A/B against another build, NO_DUO=1 a run of conv_mfma_halo2_k on the C == 32 layers
cd $GRAFT_REPO_ROOT
for cfg in "64 32 128" "32 32 128" "128 64 64" "64 64 64" "256 128 32"; do
  set -- $cfg
  for what in fwd dgrad; do
    echo "== $1 -> $2 at $3^3 $what"; python profiles/microbench_conv.py --cin $1 --cout $2 --size $3 --per-sample --what $what | tail -1
    if [ "$1" = 32 ]; then echo "   conv_mfma_halo2_k (COMA_NO_DUO=1):"; COMA_NO_DUO=1 python profiles/microbench_conv.py --cin $1 --cout $2 --size $3 --per-sample --what $what | tail -1; fi
    if [ -n "$OTHER" ]; then echo "   other lib:"; COMA_UNET_LIB=$OTHER python profiles/microbench_conv.py --cin $1 --cout $2 --size $3 --per-sample --what $what | tail -1; fi
  done
done
