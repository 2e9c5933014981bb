mkdir -p gpurun_out/r06d
python -m pytest tests -m gpu -q -x -k "down512 or rgb or k1 or videohasher or hasher or stream" > gpurun_out/r06d/pytest.log 2>&1; tail -4 gpurun_out/r06d/pytest.log
for rep in 1 2 3; do
for v in default r05f1 nostate nod nofetch; do
  if [ $v = default ]; then unset HVD_LIB_PATH; else export HVD_LIB_PATH=$PWD/build_tmp/libhvd_$v.so; fi
  python scripts/gpu_down512w_abl.py 2>&1 | tail -1
done; done > gpurun_out/r06d/f1_variants.txt 2>&1
cat gpurun_out/r06d/f1_variants.txt
unset HVD_LIB_PATH
CH=1 python scripts/gpu_down512w_abl.py 2>&1 | tail -1
CH=1 HVD_LIB_PATH=$PWD/build_tmp/libhvd_r05f1.so python scripts/gpu_down512w_abl.py 2>&1 | tail -1
