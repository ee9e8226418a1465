cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_cli.py tests/test_sam_via_reference.py -m gpu -x -q 2>&1 | tail -2
MBP=3100 PAIRS=3000000 STREAMS=2,3 python tools/e2e_probe.py 2>&1 | grep -v "amdgpu.ids"
