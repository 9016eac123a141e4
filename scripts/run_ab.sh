# same-box A/B of two builds: bash scripts/run_ab.sh <variant-name>   (ab/<variant>/libuformer_hip.so vs the in-tree library)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R; V=${1:-poly}
for r in 1 2 3; do for v in base $V; do if [ $v != base ]; then export UFORMER_HIP_LIB=$R/ab/$v/libuformer_hip.so; else unset UFORMER_HIP_LIB; fi
python bench.py --no-cpu-baseline --no-f32-mode 2>/dev/null | python scripts/print_bench.py "$v run $r"; done; done | tee $O/ab_$V.txt
