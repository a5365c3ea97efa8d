for v in nt v1 v2 v3 v4 v5 nt; do
  PM_MI355X_LIB=$PWD/pytorch-models_amd/csrc/build/libpm_$v.so python tools/sk_check.py > gpurun_out/var_$v.txt 2>&1 || exit 1
  echo "== $v"; grep "kernel=" gpurun_out/var_$v.txt | cut -c13-75
done
