#!/bin/bash
# shader clock / power while an MD-size prediction (256 atoms, resident, replayed) runs in a loop: does the chip clock up for ~60 launches of 5-50 us?
python tools/gpu_md_replay_probe.py 14000 2,2,2 &
sleep 5
for i in 1 2 3 4; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|socclk|Power" | head -8
  echo "--"; sleep 1.5
done
wait
rocm-smi --showclkfrq 2>/dev/null | grep -E "sclk|\*|[0-9]+Mhz" | head -20
rocm-smi --showperflevel 2>/dev/null | head -8
