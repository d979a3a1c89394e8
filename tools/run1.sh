set -o pipefail
timeout -k 10 250 python -m torch.distributed.run --nnodes=1 --nproc-per-node=4 --master-addr 127.0.0.1 --master-port 29611 tests/_dist_gpu_worker.py > gpurun_out/dist4.log 2>&1
echo rc=$?
grep -n "Error\|error\|Traceback\|File \"/root\|assert" gpurun_out/dist4.log | head -30
