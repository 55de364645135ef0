# Diagnostic builds of conv_mfma_duo_k (outputs are wrong; never shipped): what a phase costs without the matrix work, without
# the staging group's work, without the tile epilogue.  Run on the GPU box: bash profiles/ablate_duo.sh
cd $GRAFT_REPO_ROOT/coma_unet_amd/csrc
OBJ="api.o conv_direct.o conv_point1.o norm.o gate.o elementwise.o weights.o metrics.o comm.o"
for v in NO_MFMA NO_STAGE NO_EPI; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -DCOMA_DUO_$v -c conv_mfma.hip -o /tmp/cm_$v.o 2>/dev/null &
done
wait
cd $GRAFT_REPO_ROOT
for v in NO_MFMA NO_STAGE NO_EPI; do
  (cd coma_unet_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libcoma_$v.so $OBJ /tmp/cm_$v.o -ldl)
done
for c in "32 32 128" "64 32 128"; do
  set -- $c
  echo "== $1 -> $2 at $3^3 forward: shipped duo kernel / halo2 / ablations"
  python profiles/microbench_conv.py --cin $1 --cout $2 --size $3 --per-sample --what fwd 2>&1 | tail -1
  COMA_NO_DUO=1 python profiles/microbench_conv.py --cin $1 --cout $2 --size $3 --per-sample --what fwd 2>&1 | tail -1
  for v in NO_MFMA NO_STAGE NO_EPI; do
    echo "   $v:"; COMA_UNET_LIB=/tmp/libcoma_$v.so python profiles/microbench_conv.py --cin $1 --cout $2 --size $3 --per-sample --what fwd 2>&1 | tail -1
  done
done
