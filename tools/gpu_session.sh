# Session r6-4: what is left of k_extend_wave when every ksw_extend2 call returns at once (a -DBWAGPU_FAKE_DP build: regions are nonsense, the time is mem_chain2aln's control)
mkdir -p gpurun_out/s4
export TMPDIR=/tmp
(LIB=tools/_scratch/libbwagpu_fake.so timeout 600 python tools/ext_pack_probe.py 0 0 > gpurun_out/s4/fake.log 2>&1; echo "rc $?" >> gpurun_out/s4/fake.log)
grep -a "ext_pack\|stats run\|rc " gpurun_out/s4/fake.log
