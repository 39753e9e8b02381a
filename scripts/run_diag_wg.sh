# per-workgroup phase clocks of diagnostic builds: scripts/run_diag_wg.sh lib1.so lib2.so ...  -> gpurun_out/r4_wg_<lib>.log
R=$GRAFT_REPO_ROOT
for l in "$@"; do
PISO_HIP_LIB=$R/scripts/_bin/$l PISO_CG_PERSIST_TIMING=2 python $R/scripts/bench_cg.py 2048 > $R/gpurun_out/r4_wg_$l.log 2>&1
done
