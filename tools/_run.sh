O=gpurun_out/r06/merge3; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -5 $O/pytest.log | head -4
python tools/gpu_md_step_probe.py 100 2,2,2 > $O/step_on.log 2>&1; head -3 $O/step_on.log
python tools/gpu_md_step_probe.py 100 4,2,2 > $O/step_on_512.log 2>&1; head -3 $O/step_on_512.log
for i in 1 2; do
timeout 300 python tools/gpu_md_anatomy.py 400 > $O/md_anatomy.log 2>&1; head -12 $O/md_anatomy.log | tail -10
CHGNET_TINY_FUSE=0 CHGNET_TEAM_MIN_ANGLES=-1 timeout 300 python tools/gpu_md_anatomy.py 400 > $O/md_anatomy_off.log 2>&1; sed -n 3,4p $O/md_anatomy_off.log
done
timeout 300 python tools/gpu_md_anatomy.py 300 4,2,2 > $O/md_anatomy_512.log 2>&1; sed -n 3,4p $O/md_anatomy_512.log
