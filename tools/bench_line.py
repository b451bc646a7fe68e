import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%s GF %.0f ms %.4f orth %.2e res %.2e' % (sys.argv[1] if len(sys.argv) > 1 else '', d['value'], d['ms_per_step'], d['orth_fro'], d['residual']),
      {k: round(v * 1e3, 1) for k, v in d['roofline']['kernel_ms_per_step'].items()}, d['roofline'].get('r_factor_engine'))
