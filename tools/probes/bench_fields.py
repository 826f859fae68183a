import json,sys
for l in sys.stdin:
    if l.startswith('{"metric"'):
        d=json.loads(l); r=d['roofline']
        print(d['value'], d['ms_per_step'], 'gemm_ms', r['gemm_ms_per_step'], 'f64_gemm_ms', r['f64_gemm_ms_per_step'], 'TF', r['achieved'], 'final_loss', d.get('final_loss'),
              'potrf64', d.get('potrf_ms_f64'), 'potrf32', d.get('potrf_ms_f32'), 'map', d.get('gibbs_map_step_ms_f64'))
