cd $GRAFT_REPO_ROOT
MBP=3100 PAIRS=3000000 STREAMS=2 python tools/e2e_probe.py 2>&1 | grep -v "amdgpu.ids"
