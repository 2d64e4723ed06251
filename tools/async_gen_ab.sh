# A/B of the asynchronous generator (t2d_generator_async): GPU parity test, then the BASELINE configurations end to end
# with the generator in order on the caller's stream (T2D_ASYNC_GEN=0) and forked onto the library's stream (=1).
R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R; O=$R/gpurun_out/r02; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "async_generator or fused_random or generated" 2>&1 | tail -5
for a in 0 1; do
  echo "== T2D_ASYNC_GEN=$a"; T2D_ASYNC_GEN=$a timeout 600 python tools/config_sweep.py 2>/dev/null | tee $O/config_sweep_async$a.txt
done
