cd $GRAFT_REPO_ROOT
echo "== tconv 64->32, 64^3 -> 128^3 fwd (new kernel)"; python profiles/microbench_conv.py --cin 64 --cout 32 --stride 2 --transposed --size 64 --per-sample --what fwd
echo "== same, gather kernel"; COMA_NO_TCONV=1 python profiles/microbench_conv.py --cin 64 --cout 32 --stride 2 --transposed --size 64 --per-sample --what fwd
echo "== dgrad of 32->64 s2 at 128^3 (new kernel)"; python profiles/microbench_conv.py --cin 32 --cout 64 --stride 2 --size 128 --per-sample --what dgrad
echo "== same, gather"; COMA_NO_TCONV=1 python profiles/microbench_conv.py --cin 32 --cout 64 --stride 2 --size 128 --per-sample --what dgrad
echo "== tconv 128->64, 32^3 -> 64^3 fwd (new)"; python profiles/microbench_conv.py --cin 128 --cout 64 --stride 2 --transposed --size 32 --per-sample --what fwd
echo "== same, gather"; COMA_NO_TCONV=1 python profiles/microbench_conv.py --cin 128 --cout 64 --stride 2 --transposed --size 32 --per-sample --what fwd
