for lib in main l63; do for k in 084 085 087 088 089; do
  if [ $lib = main ]; then unset FS_LIB; else export FS_LIB=$PWD/flow-sim_amd/csrc/variants/lib_$lib.so; fi
  echo "$lib $k: $(timeout -k 10 100 python -m pytest tests/test_gpu_instantiations.py -m gpu -q -x -k "${k}-f64-irr" 2>&1 | tail -1 | cut -c1-60)"
done; done
