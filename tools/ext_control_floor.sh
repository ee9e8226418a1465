#!/bin/bash
# What is left of k_extend_wave when every ksw_extend2 call returns at once: a -DBWAGPU_FAKE_DP build of the library (regions are nonsense, the stage's
# time is mem_chain2aln's control), run through tools/ext_pack_probe.py.  Build here (CPU box, hipcc cross-compiles), run through gpurun:
#   tools/ext_control_floor.sh build && gpurun -- 'bash tools/ext_control_floor.sh run'        (round 6: 5.8 ms of the stage's 34.7, profiles/r06_ext_pack.md)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
if [ "$1" = build ]; then
  mkdir -p $ROOT/tools/_scratch
  (cd $ROOT/bwa_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -DBWAGPU_FAKE_DP -c bwagpu.hip -o /tmp/bwagpu_fake.o &&
   /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/bwagpu_fake.o bwagpu_index.o -o $ROOT/tools/_scratch/libbwagpu_fake.so)
  echo "built tools/_scratch/libbwagpu_fake.so (git-ignored; travels with the gpurun snapshot)"
else
  cd $ROOT && LIB=tools/_scratch/libbwagpu_fake.so python tools/ext_pack_probe.py 0 0
fi
