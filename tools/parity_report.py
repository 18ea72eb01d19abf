"""gpurun_out/parity_r02.jsonl (appended by tests/test_gpu_longt.py on the GPU box) -> profiles/parity_r02.json:
measured error, fp32 floor and bound per config, so that DESIGN.md section 2 is reproducible from the tree.
  python tools/parity_report.py [round_tag]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else 'r02'
recs = [json.loads(l) for l in open(os.path.join(ROOT, 'gpurun_out', 'parity_%s.jsonl' % tag)) if l.strip()]
dst = os.path.join(ROOT, 'profiles', 'parity_%s.json' % tag)
by = json.load(open(dst))['configs'] if os.path.exists(dst) else {}     # a gpurun call brings back only its own records
for r in recs:
    by.setdefault('cfg%d' % r['cfg'], {}).update(r)     # the last run of each config wins, field by field
out = dict(source='tests/test_gpu_longt.py on one MI355X, float64 numpy oracle (oracle/asr_oracle.py)',
           error_measure='max|a-b| / max|b| per tensor (gradients: + 1e-3 of the largest gradient tensor in the denominator)',
           precision=sorted({r.get('precision', 'f32x3') for r in by.values() if r['cfg'] < 100}),
           configs=by)
with open(dst, 'w') as f:
    json.dump(out, f, indent=1, sort_keys=True)
print(json.dumps(out, indent=1, sort_keys=True)[:600])
