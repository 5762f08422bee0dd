cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; : > gpurun_out/lanes_other.log
for cfg in "--kind iid --ensemble 1 --denoise 4" "--kind iid --ensemble 3 --denoise 4" "--kind normals --ensemble 4" "--scheduler lcm --denoise 4 --ensemble 1"; do
 for n in 1 2 3; do
  for rep in 1 2; do
  timeout 300 python bench.py $cfg --in-flight $n --steps 12 --warmup 2 --no-cpu-baseline --no-profile 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$cfg | in_flight=$n |', j['ms_per_step'], 'ms per map', 'one_at_a_time', j.get('latency_ms_per_map'))
" >> gpurun_out/lanes_other.log
  done
 done
done
cat gpurun_out/lanes_other.log
