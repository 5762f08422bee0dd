#!/bin/bash
# Kernel trace of one map -> the idle gaps between consecutive kernels on the stream (start[i+1] - end[i]).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; mkdir -p gpurun_out; rm -rf gpurun_out/prof
(cd /tmp && export TMPDIR=/tmp && MARIGOLD_TUNING=1 MARIGOLD_GN_COOP=0 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof -o tr -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile > $R/gpurun_out/rocprof_bench.log 2>&1)
echo "rocprof rc=$?"
python - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/prof/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# the last ~2800 kernels = the timed map (bench ran warmup + 1 step); take kernels after the calibration
ks = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows]
n = len(ks)
tail = ks[-2700:]
gaps = [tail[i + 1][0] - tail[i][1] for i in range(len(tail) - 1)]
dur = [e - s for s, e, _ in tail]
import statistics
g = [x for x in gaps if x < 50000]
print('kernels', n, 'window', len(tail), 'sum dur ms', sum(dur) / 1e6, 'sum gaps ms', sum(g) / 1e6, 'median gap ns', statistics.median(g), 'mean', sum(g) / len(g), 'wall ms', (tail[-1][1] - tail[0][0]) / 1e6)
h = collections.Counter(min(x // 500, 20) for x in g)
print('gap histogram (0.5 us bins):', sorted(h.items()))
byk = collections.defaultdict(list)
for i in range(len(tail) - 1):
    if gaps[i] < 50000:
        byk[tail[i + 1][2][:60]].append(gaps[i])
for k, v in sorted(byk.items(), key=lambda kv: -sum(kv[1]))[:25]:
    print(f'{k:60s} n={len(v):4d} mean gap before {sum(v)/len(v)/1000:6.2f} us')
PY
rm -rf gpurun_out/prof
