"""gpurun_out/<tag>_xchg_floor.txt (tools/latency_evidence.sh) -> profiles/<tag>_latency_floor.json: the two figures
bench.py's roofline.latency_floor_us is built from.
  exchange_us      one chip-wide all-gather under the sc1 counter protocol of scan_common.h, 2 groups x 128 workgroups
                   (the persistent scans' geometry), payload 32 KB per group, nothing else on the chip: xchg3 NG=128
  mfma_issue_ns    issue interval of a dependent v_mfma_f32_16x16x4_f32 at one wave per SIMD (mfma_rate, grid 256)
usage: python tools/latency_floor.py [tag]"""
import json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else 'r03'
src = os.path.join(ROOT, 'gpurun_out', tag + '_xchg_floor.txt')
txt = open(src).read()
sec = {}
cur = None
for line in txt.splitlines():
    if line.startswith('## '):
        cur = sec.setdefault(line[3:].strip(), [])
    elif cur is not None:
        cur.append(line)
ex = [float(m.group(1)) for l in sec['xchg3_128_128'] if (m := re.search(r'([\d.]+) us/iter', l))]
mf = [float(m.group(1)) for l in sec['mfma_rate'] if 'chains 1' in l and 'grid 256' in l.replace('grid  ', 'grid ')
      and (m := re.search(r'([\d.]+) ns per MFMA', l))]
if not mf:
    mf = [float(m.group(1)) for l in sec['mfma_rate'] if 'chains 1' in l and (m := re.search(r'([\d.]+) ns per MFMA', l))]
out = dict(source='profiles/%s_xchg_floor.txt (tools/latency_evidence.sh on one MI355X)' % tag,
           exchange_us=min(ex), exchange_us_runs=ex, mfma_issue_ns=min(mf), mfma_issue_ns_runs=mf,
           per_step={'hyper_scan_fwd_kernel': dict(dependent_exchanges=2, mfma_per_wave=40 + 32 + 24),
                     'hyper_scan_bwd_kernel': dict(dependent_exchanges=2, mfma_per_wave=24 + 32 + 32)},
           note='latency floor of a step = dependent_exchanges * exchange_us + mfma_per_wave * mfma_issue_ns / 1000 '
                '(cfg3 shape: N=512, H=128, one wave per SIMD); exchanges measured in isolation, so a floor, not a model')
dst = os.path.join(ROOT, 'profiles', tag + '_latency_floor.json')
json.dump(out, open(dst, 'w'), indent=1)
print(json.dumps(out, indent=1))
