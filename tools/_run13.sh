cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "win7 or w7 or odd or tiny7 or (basic_block and (w8 or hid4))" 2>&1 | tail -5
python - <<'PY'
import json
d=json.load(open('gpurun_out/parity.json'))
for r in d['records']:
    if 'fast' in r['test'] and ('win7' in r['test'] or 'w7' in r['test'] or 'C24_h8x3_w7' in r['test']): print(r['test'], f"l2={r['rel_l2']:.2e} max={r['max_rel']:.2e}")
PY
