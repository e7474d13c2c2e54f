mkdir -p gpurun_out/r4
for rep in 1 2; do
for m in 0 1 2 3; do
  echo "== LAV_PILLAR_EARLY=$m"
  LAV_PILLAR_EARLY=$m timeout 120 python tools/pillar_probe.py 2>&1 | grep -v "^torch copy\|419 MB"
done
done > gpurun_out/r4/pillar_early.log 2>&1
LAV_PILLAR_TRACE=1 timeout 120 python tools/pillar_probe.py "lidar-like 3x10923" > gpurun_out/r4/pillar_early_trace.log 2>&1
timeout 600 python -m pytest tests/test_gpu_pillar.py -x -q > gpurun_out/r4/suite_pillar_early.log 2>&1
LAV_PILLAR_EARLY=3 timeout 600 python -m pytest tests/test_gpu_pillar.py -x -q >> gpurun_out/r4/suite_pillar_early.log 2>&1
tail -3 gpurun_out/r4/suite_pillar_early.log
cat gpurun_out/r4/pillar_early.log
