O=gpurun_out/r06/zsave; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -5 $O/pytest.log | head -3
CHGNET_WIN_MIN_ATOMS_PER_WAVE=0 timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py tests/test_v020.py -m gpu -x -q > $O/pytest_pa.log 2>&1; echo "pytest per-atom forced exit $?"; tail -5 $O/pytest_pa.log | head -3
timeout 600 python bench.py --no-configs --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['roofline'].get('traffic'))"
CHGNET_ZSAVE=0 timeout 600 python bench.py --no-configs --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])"
