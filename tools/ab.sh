#!/bin/bash
# tools/ab.sh OUT LIB...: the default C3 bench line (no CPU leg, no secondary pass) once per library build, for A/B runs on one box.
# AB_ARGS: extra bench arguments; AB_REPS: runs per library (default 2)
out=$1; shift
mkdir -p gpurun_out
for lib in "$@"; do
  for rep in $(seq 1 ${AB_REPS:-2}); do
    PQV_LIB_PATH=$PWD/pq-vector_amd/libpqv_$lib.so python bench.py --no-cpu --no-secondary --no-configs --recall 0 --parity-queries 0 ${AB_ARGS} 2>/dev/null |
      python -c "
import sys, json
d = json.loads(sys.stdin.readline())
r = d.get('roofline', {})
s = d.get('single_query', {})
print('$lib ${AB_TAG}', $rep, 'qps %.0f  ms %.4f  serial %.4f  kernel_ms %.4f  single_p50 %.1f  surv/q %.0f' % (d['value'], d['ms_per_step'], d.get('ms_per_step_serial', 0), r.get('kernel_ms', 0), s.get('p50_us', 0), d['counters']['screen_survivors'] / max(1, d['counters']['queries'])))
" >> gpurun_out/$out
  done
done
