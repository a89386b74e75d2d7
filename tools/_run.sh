O=gpurun_out/r06/t6; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_round6.py tests/test_torch_bridge.py tests/test_gpu_round2.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -25 $O/pytest.log
