set -x
mkdir -p gpurun_out/r3a
nproc; python -c "import os; print(len(os.sched_getaffinity(0)))"; cat /sys/fs/cgroup/cpu.max 2>/dev/null
(time python -m pytest tests/test_gpu_pf.py -x -q 2>&1 | tail -15) > gpurun_out/r3a/pytest_pf.log 2>&1
python tools/pf_explore.py sphere > gpurun_out/r3a/pf_sphere.log 2>&1
python tools/pf_explore.py room > gpurun_out/r3a/pf_room.log 2>&1
tail -5 gpurun_out/r3a/pytest_pf.log; cat gpurun_out/r3a/pf_sphere.log gpurun_out/r3a/pf_room.log
