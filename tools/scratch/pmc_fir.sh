#!/bin/bash
# PMC passes over the blur kernel (run on the GPU box): bash tools/scratch/pmc_fir.sh
set -o pipefail
out=gpurun_out/pmc_fir; rm -rf $out; mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
for set in "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INST_CYCLES_VMEM" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/$tag -- python tools/microbench.py fir --only "blur 512ch 256" --reps 3 > $out/$tag.log 2>&1
done
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/pmc_fir/*/*/*counter_collection.csv')):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if 'upfirdn2d_vec' in r['Kernel_Name'] and 'unsigned short' in r['Kernel_Name']:
            a = acc[r['Counter_Name']]; a[0] += float(r['Counter_Value']); a[1] += 1
    for k, (v, n) in acc.items():
        print(f"{k:32s} {v / n:14.4g} per launch ({n} launches)")
PY
find $out -name "*.csv" -size +1M -delete
