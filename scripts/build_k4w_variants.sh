#!/bin/bash
# builds libmarigold_hip_k4w<i>.so for alternative gen_k4w.py schedules (same-box A/B via MARIGOLD_HIP_LIB); restores the default
cd "$(dirname "$0")/../marigold_amd/csrc" || exit 1
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result -Wno-unused-value -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form=0"
i=0
cp igemm2_k4w.inc /tmp/igemm2_k4w.inc.keep
while read -r sched; do
  [ -z "$sched" ] && continue
  i=$((i + 1))
  python3 gen_k4w.py $sched > igemm2_k4w.inc || exit 1
  mkdir -p build_v$i
  /opt/rocm/bin/hipcc $FLAGS -c igemm2_big.hip -o build_v$i/igemm2_big.o || exit 1
  objs=$(ls build/*.o | grep -v igemm2_big.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs build_v$i/igemm2_big.o -o ../libmarigold_hip_k4w$i.so || exit 1
  echo "k4w$i: $sched"
done
cp /tmp/igemm2_k4w.inc.keep igemm2_k4w.inc
touch -r /tmp/igemm2_k4w.inc.keep igemm2_k4w.inc
