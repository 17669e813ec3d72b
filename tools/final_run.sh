#!/bin/bash
# one gpurun call: the whole GPU suite, smoke, the headline bench and the other scenes (outputs under gpurun_out/)
cd ${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-final}
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_tests.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/${TAG}_tests.log
timeout -k 10 300 python __graft_entry__.py smoke > gpurun_out/${TAG}_smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/${TAG}_smoke.log
timeout -k 10 300 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench rc=$?"
python - <<PY
import json
d=json.load(open('gpurun_out/${TAG}_bench.json'))
print({k:d[k] for k in ('value','ms_per_step','moving_camera_ms_per_frame','frame_matches_oracle')}, 'pipelined', d['pipelined']['ms_per_frame'], 'valu frac', d['roofline']['frac'], 'hbm frac', d['roofline'].get('hbm_frac'), 'cpu', d['cpu_baseline']['value'])
PY
timeout -k 10 300 python bench.py --scene heightfield5m --no-cpu-baseline > gpurun_out/${TAG}_bench_5m.json 2> gpurun_out/${TAG}_bench_5m.err
timeout -k 10 300 python bench.py --scene soup --no-cpu-baseline > gpurun_out/${TAG}_bench_soup.json 2> gpurun_out/${TAG}_bench_soup.err
python - <<PY
import json
for s in ('5m','soup'):
    d=json.load(open('gpurun_out/${TAG}_bench_%s.json'%s)); print(s, d['ms_per_step'], d['value'], d['roofline'].get('hbm_frac'))
PY
timeout -k 10 300 python bench.py --force-dist --steps 50 --warmup 5 --check-dist-frame > gpurun_out/${TAG}_bench_dist1.json 2> gpurun_out/${TAG}_bench_dist1.err; echo "dist rc=$?"
