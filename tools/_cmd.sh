mkdir -p gpurun_out/r3
timeout 300 python -m pytest tests/test_gpu_train.py -x -q -k "bn_act or heads_train or matches_reference_trainer" > gpurun_out/r3/bn_tests.log 2>&1
timeout 200 python tools/conv_fault_probe.py > gpurun_out/r3/conv_fault.log 2>&1
timeout 200 python bench.py --mode train_full --steps 5 --warmup 3 --log-every 1 > gpurun_out/r3/train_hip.log 2>&1
LAV_TRAIN_BN=torch timeout 200 python bench.py --mode train_full --steps 5 --warmup 3 --log-every 1 > gpurun_out/r3/train_torchbn.log 2>&1
