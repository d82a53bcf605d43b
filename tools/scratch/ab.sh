set -e
mkdir -p gpurun_out
: > gpurun_out/ab.log
cp multi_stylegan_amd/libmsg_hip.so /tmp/new.so
timeout -k 10 600 python -m pytest tests/test_hip_conv.py tests/test_hip_models.py -m gpu -x -q >> gpurun_out/ab.log 2>&1
for v in old new; do
if [ $v = old ]; then cp tools/scratch/libmsg_old.so multi_stylegan_amd/libmsg_hip.so; else cp /tmp/new.so multi_stylegan_amd/libmsg_hip.so; fi
echo "== $v" >> gpurun_out/ab.log
python tools/microbench.py conv --only "shared" --reps 20 2>&1 | grep fprop >> gpurun_out/ab.log
done
for v in old new old new; do
if [ $v = old ]; then cp tools/scratch/libmsg_old.so multi_stylegan_amd/libmsg_hip.so; else cp /tmp/new.so multi_stylegan_amd/libmsg_hip.so; fi
echo "== $v" >> gpurun_out/ab.log
python bench.py --no-cpu-baseline --steps 32 --warmup 5 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['achieved'])" >> gpurun_out/ab.log
done
cp /tmp/new.so multi_stylegan_amd/libmsg_hip.so
