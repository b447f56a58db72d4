cd $GRAFT_REPO_ROOT; export PYTHONPATH=$GRAFT_REPO_ROOT; mkdir -p gpurun_out
{ echo "== c5 share"; python scripts/tiled_timing.py 2>&1 | grep -v amdgpu.ids
  echo "== skew"; python scripts/tiled_timing.py 0 0 skew 2>&1 | grep -v amdgpu.ids
  echo "== c5 full"; python scripts/tiled_timing.py 10000000 512 2>&1 | grep -v amdgpu.ids; } > gpurun_out/r06b_tiled_timing.log 2>&1
cat gpurun_out/r06b_tiled_timing.log
