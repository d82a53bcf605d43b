set -e
mkdir -p gpurun_out
: > gpurun_out/ab.log
timeout -k 10 900 python -m pytest tests/test_hip_models.py tests/test_hip_ddp.py -m gpu -x -q >> gpurun_out/ab.log 2>&1
for v in 0 1 0 1; do
echo "== FORK=$v" >> gpurun_out/ab.log
MSG_FUSE_INPUT_FORK=$v python bench.py --no-cpu-baseline --steps 32 --warmup 5 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['peak_mem_GiB'])" >> gpurun_out/ab.log
done
