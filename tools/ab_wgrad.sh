# diagnostics: conv weight gradients + whole update, per-observation conv2/conv3 kernel (default) vs the implicit-GEMM path
for v in fused gemm; do
  if [ $v = gemm ]; then export A0_NO_CONV23_WGRAD_FUSED=1; else unset A0_NO_CONV23_WGRAD_FUSED; fi
  echo "== $v"; python tools/ubench_convwgrad.py; python tools/ubench_update.py
done
