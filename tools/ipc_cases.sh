TL=$(python -c "import torch,os,glob; print(glob.glob(os.path.join(os.path.dirname(torch.__file__),'lib','libamdhip64.so*'))[0])")
for mb in 2047 2048 2049; do echo "== torch lib, $mb MiB"; HIPLIB=$TL POLL_S=20 timeout -k 5 40 python tools/ipc_probe.py 8 $mb 2>&1 | grep MiB | tail -1; done
echo "== torch lib, 3360 MiB, flags 0"; HIPLIB=$TL IPCFLAGS=0 POLL_S=20 timeout -k 5 40 python tools/ipc_probe.py 8 3360 2>&1 | grep MiB | tail -1
echo "== engine probe T=300 with the system HIP/HSA runtime preloaded"
LD_PRELOAD=/opt/rocm/lib/libamdhip64.so:/opt/rocm/lib/libhsa-runtime64.so JOIN_S=60 timeout -k 5 90 python tools/ipc_engine_probe.py 1048576 300 2>&1 | grep "ipc_open\|still running\|rror" | head -16
