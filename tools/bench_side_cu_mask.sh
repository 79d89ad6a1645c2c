# The second stream (next batch's sampling + geometry, frozen text encoder) confined to a subset of the CUs: what the main stream
# gets back.  usage (GPU box): bash tools/bench_side_cu_mask.sh > gpurun_out/side_cu_mask.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for m in "" 128 64 32 16 low128 low64 low32; do
  for rep in 1 2; do
    EDA_SIDE_CU_MASK=$m python bench.py --in-step-steps 0 --cpu-scenes 0 --kernel-steps 0 2>/tmp/err.txt | python -c "
import json,sys
l=[x for x in sys.stdin.read().splitlines() if x.startswith('{')]
if l:
    d=json.loads(l[-1]); print('mask=%-8s %8.2f scenes/s %7.3f ms' % ('$m' or 'none', d['value'], d['ms_per_step']))
else:
    print('mask=$m FAILED'); print(open('/tmp/err.txt').read()[-1500:])
"
  done
done
