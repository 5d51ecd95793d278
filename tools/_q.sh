cd /root/repo
POISON_ALL=1 timeout 250 python tools/poison_empty.py > gpurun_out/poison.txt 2>&1
