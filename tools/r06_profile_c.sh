# round 6 evidence, part C: PMC counters — the quantile configurations' dense forward family (tools/pmc_quantile.sh, incl. the LDS / VALU group) and the counter table of
# the three fc1 forward shapes (8 192 / 16 384 / 32 768 rows) under six and nine products (tools/pmc_gemm.sh)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
R=r06 bash tools/pmc_quantile.sh > gpurun_out/r06/pmc_quantile.log 2>&1; echo "pmc_quantile rc=$?"
for rows in 8192 32768; do
  for p in 6 9; do
    A0_X9_PRODUCTS=$p bash tools/pmc_gemm.sh fwd $rows 1 > gpurun_out/r06/pmc_gemm_fwd_${rows}_x$p.txt 2>&1; echo "rows=$rows products=$p"; tail -2 gpurun_out/r06/pmc_gemm_fwd_${rows}_x$p.txt | cut -c1-600
  done
done
