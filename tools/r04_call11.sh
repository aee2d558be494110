#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04c11; mkdir -p $O
timeout 900 python -m pytest tests/test_float_gpu.py tests/test_decode_sum_gpu.py tests/test_last_register_gpu.py tests/test_decode_gpu.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
r=json.load(open('gpurun_out/r04c11/bench.json'))
e=r['extras']
print('headline', r['value'], r['roofline']['frac'])
for k in ('decode_sweep_by_bit_width','decode_sweep_by_bit_width_2pct_exceptions'): print(k, e[k]['summary'])
for k in ('encode_alp_mixed','encode_alp_rd','encode_alp_mixed_exc0','encode_alp_mixed_exc10'): print(k, {q:e[k][q] for q in ('ms','roofline_frac_algorithmic','vector_encode_ms','search_in_front_ms','decode_roofline_frac_algorithmic')})
print({q:(v['ms'] if isinstance(v,dict) else v) for q,v in e['decode_sum_fused'].items() if q!='note'}, e['decode_sum_fused']['roofline_frac_algorithmic'])
for k,v in e['float_path'].items(): print(k, {q:v[q] for q in ('encode_ms','encode_roofline_frac_algorithmic','decode_ms','decode_roofline_frac_algorithmic','decode_sum_fused_ms','decode_sum_roofline_frac_algorithmic','decode_sum_by_kernel')})
PY
