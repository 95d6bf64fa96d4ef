# Two tensor-parallel ranks as two processes on one card (rehearsal of bench.py --tp).
run() { SD_BENCH_BACKEND=gloo SD_BENCH_DEVICE=0 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $1 bench.py --gpus 2 --tp 2 --tp-collective ipc --steps 20 --warmup 4 "${@:2}" 2>&1 | grep -E "^\{|RuntimeError|first exchange"; }
echo "== replicated draft"; run 29511
echo "== sharded draft"; run 29512 --tp-draft sharded
