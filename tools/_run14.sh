cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "wide and C48" 2>&1 | tail -5
python - <<'PY'
import json
d=json.load(open('gpurun_out/parity.json'))
for r in d['records']:
    if 'C48' in r['test']: print(r['test'], f"l2={r['rel_l2']:.2e} max={r['max_rel']:.2e}")
PY
python tools/profile_block.py --level 1 --iters 30; SWF_WIN48=0 python tools/profile_block.py --level 1 --iters 30
python tools/profile_block.py --level 1 --iters 30 --decoder 1; SWF_WIN48=0 python tools/profile_block.py --level 1 --iters 30 --decoder 1
