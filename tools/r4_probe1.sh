#!/bin/bash
# round 4, first contact: clock probe, what the SMI tools give an ordinary user, the driver's exact bench command
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4p1; mkdir -p $O
python3 - > $O/clock.txt 2>&1 <<'PY'
import torch, time
import nvjpeg_imagecompressor_amd as mij
torch.cuda.set_device(0)
for it in (256, 1024, 1024, 4096):
    print(it, mij.clock_probe_device(it, torch.cuda.current_stream().cuda_stream))
PY
(rocm-smi --showclocks --showpower --showmaxpower --showperflevel 2>&1 | head -60) > $O/rocm_smi.txt
(amd-smi metric -g 0 --clock --power 2>&1 | head -80) > $O/amd_smi.txt
(amd-smi static -g 0 --limit 2>&1 | head -60) >> $O/amd_smi.txt
ls /sys/class/drm/card*/device/pp_dpm_sclk > $O/sysfs.txt 2>&1; cat /sys/class/drm/card*/device/pp_dpm_sclk >> $O/sysfs.txt 2>&1
cat /sys/class/drm/card*/device/pp_dpm_mclk >> $O/sysfs.txt 2>&1
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err
python3 bench.py --gpus 1 --steps 20 --warmup 5 --two-streams --no-cpu-baseline --no-psnr > $O/bench_two_streams_20.json 2> $O/bench_two_streams_20.err
python3 bench.py --gpus 1 --steps 200 --warmup 5 --no-cpu-baseline --no-psnr > $O/bench_200.json 2> $O/bench_200.err
python3 bench.py --gpus 1 --steps 200 --warmup 5 --two-streams --no-cpu-baseline --no-psnr > $O/bench_two_streams_200.json 2> $O/bench_two_streams_200.err
tail -c 600 $O/clock.txt
