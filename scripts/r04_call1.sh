#!/bin/bash
# round 4, first GPU call: the whole GPU suite with the round's new paths (complex ORAS / GEVP / Helmholtz, overlapped halo, device Gram
# all-reduce), then configs[4]'s share on the real Helmholtz problem, then the shared-GPU double of the N > 1 path (exchange hidden)
mkdir -p gpurun_out/r04
export OMP_NUM_THREADS=8
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/r04/call1_gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -25 gpurun_out/r04/call1_gpu_tests.log | cut -c1-400
( time timeout 600 python bench.py --problem helmholtz --grid 64 --mu 8 --geneo-nu 12 ) > gpurun_out/r04/call1_helmholtz.log 2> gpurun_out/r04/call1_helmholtz.err
grep '^{"metric"' gpurun_out/r04/call1_helmholtz.log | tail -1 > gpurun_out/r04/call1_helmholtz.json
python - <<'PY'
import json
try:
    d = json.load(open('gpurun_out/r04/call1_helmholtz.json'))
    print('helmholtz', d['value'], d['ms_per_step'], d['roofline']['frac'], d['phases_ms'], d['two_level'])
except Exception as e:
    print('helmholtz line missing', e)
PY
tail -5 gpurun_out/r04/call1_helmholtz.err
BENCH_SHARE_GPU=1 timeout 600 python bench.py --gpus 4 --problem helmholtz --grid 32 --mu 8 --geneo-nu 6 --no-cpu-baseline > gpurun_out/r04/call1_share4_helmholtz.log 2> gpurun_out/r04/call1_share4_helmholtz.err
grep '^{"metric"' gpurun_out/r04/call1_share4_helmholtz.log | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print('share4', d['value'], d['ms_per_step'], d.get('exchange_ms'), d['config'].get('peer_gpus_histogram'), d['two_level'].get('gmres'))
except Exception as e: print('share4 line missing', e)"
tail -5 gpurun_out/r04/call1_share4_helmholtz.err
