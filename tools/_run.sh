O=gpurun_out/r06/idx; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -5 $O/pytest.log | head -4
CHGNET_TEAM_MIN_ANGLES=0 timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_team0.log 2>&1; echo "pytest team0 exit $?"; tail -5 $O/pytest_team0.log | head -3
python tools/gpu_md_step_probe.py 100 2,2,2 0.15 > $O/step_on.log 2>&1; head -3 $O/step_on.log
python tools/gpu_md_step_probe.py 100 4,2,2 0.15 > $O/step_on_512.log 2>&1; head -3 $O/step_on_512.log
